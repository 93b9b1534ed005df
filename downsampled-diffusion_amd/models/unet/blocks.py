"""UNet building blocks: reference-shaped parameter containers whose arithmetic runs in HIP kernels.

Module nesting and parameter names follow reference models/unet/blocks.py:8-134 exactly, so a
reference ``state_dict`` loads with ``strict=True``.  The nn.Conv2d / nn.Linear /
nn.GroupNorm members only HOLD the canonical (OIHW etc.) parameters; nothing here calls their forward.
Each block has ``forward_nhwc`` (the real implementation, NHWC tensors, kernels from ddk.ops) and a
``forward`` that takes / returns NCHW like the reference for drop-in use of a single block.
The whole-network eval path does not go through these per-block calls: ``Unet.forward`` hands the
entire forward to the native plan (csrc/unet_plan.hip).
"""
import torch
from torch import nn

from ddk import ops
from ddk.lib import DDKError

HEADS, DIM_HEAD = 4, 32  # LinearAttention defaults, the only values the reference ever uses (blocks.py:119)


class _Packed:
    """Cache of kernel-layout copies of canonical parameters, refreshed when the parameter changes."""

    def __init__(self):
        self._store = {}

    def get(self, key, param, fn):
        tag = (param._version, param.data_ptr(), str(param.device))
        hit = self._store.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, fn(param.detach()))
            self._store[key] = hit
        return hit[1]


def _require_device(x):
    if not x.is_cuda:
        raise DDKError("the UNet runs on HIP kernels only: move the model and inputs to a ROCm device "
                       "(there is no CPU fallback)")


def _to_nhwc(x):
    _require_device(x)
    return ops.nchw_to_nhwc(x.contiguous().float(), ops.pad32(x.shape[1]))


class Residual(nn.Module):
    """fn(x) + x (blocks.py:8-14); the add is fused into fn's last kernel when fn supports it."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward_nhwc(self, x):
        return self.fn.forward_nhwc(x, resid=x)

    def forward(self, x, *args, **kwargs):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class SinusoidalPosEmb(nn.Module):
    """Position 0 of ``time_mlp`` (blocks.py:17-29).  Parameter-free; its sin/cos are evaluated inside
    ddk_time_mlp together with the two Linears, from the frequency table of ddk.plan.sinusoidal_freqs."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        raise DDKError("SinusoidalPosEmb is fused into the time-MLP kernel; call Unet.time_embedding(t)")


class Upsample(nn.Module):
    """ConvTranspose2d(dim, dim, 4, 2, 1) (blocks.py:32-38) as 4 output phases of 2x2 taps."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose2d(dim, dim, 4, 2, 1)
        self._packed = _Packed()

    def forward_nhwc(self, x):
        w = self._packed.get("w", self.conv.weight, ops.pack_convT_weight)
        return ops.conv(ops.CONVT4X4_S2, x, w, self.conv.bias.detach())

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class Downsample(nn.Module):
    """Conv2d(dim, dim, 3, stride 2, pad 1) (blocks.py:41-47)."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim, 3, 2, 1)
        self._packed = _Packed()

    def forward_nhwc(self, x):
        w = self._packed.get("w", self.conv.weight, ops.pack_conv_weight)
        return ops.conv(ops.CONV3X3_S2, x, w, self.conv.bias.detach())

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class LayerNorm(nn.Module):
    """Channel LayerNorm with eps added to the std (blocks.py:50-60)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))

    def forward_nhwc(self, x):
        return ops.chan_layernorm(x, self.g.detach(), self.b.detach(), self.eps)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class PreNorm(nn.Module):
    """fn(LayerNorm(x)) (blocks.py:63-71)."""

    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)

    def forward_nhwc(self, x, resid=None):
        return self.fn.forward_nhwc(self.norm.forward_nhwc(x), resid=resid)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class Block(nn.Module):
    """Conv3x3(pad 1) -> GroupNorm(groups) -> Mish (blocks.py:74-84)."""

    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv2d(dim, dim_out, 3, padding=1),
            nn.GroupNorm(groups, dim_out),
            nn.Mish(),
        )
        self.groups = groups
        self._packed = _Packed()

    def forward_nhwc(self, x, x2=None, temb=None, addend=None):
        conv, norm = self.block[0], self.block[1]
        w = self._packed.get("w", conv.weight, ops.pack_conv_weight)
        return ops.conv3x3_groupnorm_mish(x, w, conv.bias.detach(), norm.weight.detach(), norm.bias.detach(), x2=x2,
                                          temb=temb, addend=addend, groups=self.groups, eps=norm.eps)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class ResnetBlock(nn.Module):
    """Block -> (+ Linear(Mish(t))) -> Dropout -> Block, plus a 1x1 (or identity) skip (blocks.py:87-115)."""

    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8, dropout=0):
        super().__init__()
        self.mlp = nn.Sequential(nn.Mish(), nn.Linear(time_emb_dim, dim_out)) if time_emb_dim is not None else None
        self.dropout = nn.Dropout(p=dropout)
        self.block1 = Block(dim, dim_out, groups)
        self.block2 = Block(dim_out, dim_out, groups)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()
        self._packed = _Packed()

    def time_shift(self, time_emb):
        """Linear(Mish(time_emb)) -> [B, dim_out] (blocks.py:92-95,108-109)."""
        lin = self.mlp[1]
        wt = self._packed.get("mlp", lin.weight, lambda w: w.t().contiguous())
        return ops.time_proj(ops.mish(time_emb.contiguous()), wt, lin.bias.detach())

    def forward_nhwc(self, x, time_emb=None, x2=None):
        if self.training and self.dropout.p > 0:
            raise DDKError("train-mode dropout is handled by the training path (trainers/), not by block forward")
        shift = self.time_shift(time_emb) if (self.mlp is not None and time_emb is not None) else None
        h = self.block1.forward_nhwc(x, x2=x2, temb=shift)
        if isinstance(self.res_conv, nn.Identity):
            res = x
        else:
            w = self._packed.get("res", self.res_conv.weight, ops.pack_conv_weight)
            res = ops.conv(ops.CONV1X1, x, w, self.res_conv.bias.detach(), x2=x2)
        return self.block2.forward_nhwc(h, addend=res)

    def forward(self, x, time_emb):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x), time_emb))


class LinearAttention(nn.Module):
    """O(n) attention: softmax over pixels on k, 32x32 context per head (blocks.py:118-134)."""

    def __init__(self, dim, heads=HEADS, dim_head=DIM_HEAD):
        super().__init__()
        if dim_head != 32:
            raise DDKError("LinearAttention: the HIP kernels are built for dim_head = 32")
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden, dim, 1)
        self._packed = _Packed()

    def forward_nhwc(self, x, resid=None):
        wq = self._packed.get("qkv", self.to_qkv.weight, ops.pack_conv_weight)
        wo = self._packed.get("out", self.to_out.weight, ops.pack_conv_weight)
        qkv = ops.conv(ops.CONV1X1, x, wq)
        att, _ = ops.linattn(qkv, self.heads)
        return ops.conv(ops.CONV1X1, att, wo, self.to_out.bias.detach(), resid=resid)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))
