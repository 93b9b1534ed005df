"""Oracle: functional CPU restatement of the reference UNet (test infrastructure only).

Operates on a plain ``state_dict`` (reference key layout) so
that it shares no module code with either the reference or the product.

Follows reference models/unet/blocks.py:8-134 and models/unet/unet.py:10-104.
"""
import math

import torch
import torch.nn.functional as F

HEADS = 4          # blocks.py:119  LinearAttention(dim, heads=4, dim_head=32)
DIM_HEAD = 32
GROUPS = 8         # blocks.py:75   Block(..., groups=8)
GN_EPS = 1e-5      # torch.nn.GroupNorm default
LN_EPS = 1e-5      # blocks.py:51


def mish(x):
    """x * tanh(softplus(x)), softplus threshold 20 (torch F.mish semantics; blocks.py:80,93)."""
    sp = torch.where(x > 20.0, x, torch.log1p(torch.exp(torch.clamp(x, max=20.0))))
    return x * torch.tanh(sp)


def sinusoidal_freqs(dim):
    """blocks.py:24-26: exp(arange(half) * -(ln 1e4 / (half-1))) in fp32."""
    half = dim // 2
    step = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half) * -step)


def sinusoidal_embedding(t, dim):
    """blocks.py:22-29: cat(sin(t*f), cos(t*f)); t may be int64 (promotes to fp32)."""
    arg = t[:, None] * sinusoidal_freqs(dim)[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_mlp(sd, pre, t, dim):
    """unet.py:30-35: SinusoidalPosEmb -> Linear(dim,4dim) -> Mish -> Linear(4dim,dim)."""
    e = sinusoidal_embedding(t, dim)
    e = F.linear(e, sd[pre + "time_mlp.1.weight"], sd[pre + "time_mlp.1.bias"])
    e = mish(e)
    return F.linear(e, sd[pre + "time_mlp.3.weight"], sd[pre + "time_mlp.3.bias"])


def block(sd, pre, x):
    """blocks.py:74-84: Conv3x3(pad 1) -> GroupNorm(8) -> Mish."""
    h = F.conv2d(x, sd[pre + "block.0.weight"], sd[pre + "block.0.bias"], padding=1)
    h = F.group_norm(h, GROUPS, sd[pre + "block.1.weight"], sd[pre + "block.1.bias"], GN_EPS)
    return mish(h)


def resnet_block(sd, pre, x, temb):
    """blocks.py:105-115 in eval mode (dropout is identity).

    h = Block1(x); h += Linear(Mish(temb))[:, :, None, None]; h = Block2(h); return h + res_conv(x)
    """
    h = block(sd, pre + "block1.", x)
    shift = F.linear(mish(temb), sd[pre + "mlp.1.weight"], sd[pre + "mlp.1.bias"])
    h = h + shift[:, :, None, None]
    h = block(sd, pre + "block2.", h)
    if (pre + "res_conv.weight") in sd:
        res = F.conv2d(x, sd[pre + "res_conv.weight"], sd[pre + "res_conv.bias"])
    else:
        res = x
    return h + res


def chan_layernorm(x, g, b):
    """blocks.py:57-60: per-pixel mean / biased var over C; eps is added to the STD."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) / (var.sqrt() + LN_EPS) * g + b


def linear_attention(sd, pre, x):
    """blocks.py:126-134: to_qkv (no bias) -> softmax_n(k) -> ctx = k v^T -> out = ctx^T q -> to_out."""
    bsz, _, hh, ww = x.shape
    n = hh * ww
    qkv = F.conv2d(x, sd[pre + "to_qkv.weight"])
    qkv = qkv.reshape(bsz, 3, HEADS, DIM_HEAD, n)          # channel = (qkv, head, c)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q)
    out = out.reshape(bsz, HEADS * DIM_HEAD, hh, ww)
    return F.conv2d(out, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])


def attention_block(sd, pre, x):
    """Residual(PreNorm(dim, LinearAttention(dim))): blocks.py:8-14,63-71."""
    xn = chan_layernorm(x, sd[pre + "fn.norm.g"], sd[pre + "fn.norm.b"])
    return linear_attention(sd, pre + "fn.fn.", xn) + x


def downsample(sd, pre, x):
    """blocks.py:41-47: Conv2d(C, C, 3, stride 2, pad 1)."""
    return F.conv2d(x, sd[pre + "conv.weight"], sd[pre + "conv.bias"], stride=2, padding=1)


def upsample(sd, pre, x):
    """blocks.py:32-38: ConvTranspose2d(C, C, 4, stride 2, pad 1)."""
    return F.conv_transpose2d(x, sd[pre + "conv.weight"], sd[pre + "conv.bias"], stride=2, padding=1)


def unet_levels(cfg):
    """unet.py:19-27: dims = [in, chan*m...]; in_out pairs."""
    dims = [cfg["unet_in"]] + [cfg["unet_chan"] * m for m in cfg["unet_dims"]]
    return list(zip(dims[:-1], dims[1:]))


def unet_forward(sd, cfg, x, t, pre=""):
    """unet.py:74-104 (eval mode).  x: B x C x S x S fp32, t: B (int64) -> B x C x S x S.

    Structure facts: every down level but the last has a stride-2 conv,
    every up level has a transpose conv (``is_last`` is never true in the up loop), and
    the first (full-resolution) skip is pushed but never popped.
    """
    in_out = unet_levels(cfg)
    nres = len(in_out)
    temb = time_mlp(sd, pre, t, cfg["unet_chan"])
    skips = []
    for i in range(nres):
        p = f"{pre}downs.{i}."
        x = resnet_block(sd, p + "0.", x, temb)
        x = resnet_block(sd, p + "1.", x, temb)
        x = attention_block(sd, p + "2.", x)
        skips.append(x)
        if i < nres - 1:
            x = downsample(sd, p + "3.", x)
    x = resnet_block(sd, pre + "mid_block1.", x, temb)
    x = attention_block(sd, pre + "mid_attn.", x)
    x = resnet_block(sd, pre + "mid_block2.", x, temb)
    for i in range(nres - 1):
        p = f"{pre}ups.{i}."
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(sd, p + "0.", x, temb)
        x = resnet_block(sd, p + "1.", x, temb)
        x = attention_block(sd, p + "2.", x)
        x = upsample(sd, p + "3.", x)
    x = block(sd, pre + "final_conv.0.", x)
    return F.conv2d(x, sd[pre + "final_conv.1.weight"], sd[pre + "final_conv.1.bias"])
