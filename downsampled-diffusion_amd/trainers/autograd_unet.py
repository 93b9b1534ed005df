"""Differentiable forward of the UNet / resamplers / loss for TRAINING: the same arithmetic as the native inference plan,
composed from ddk.autograd Functions (HIP forward + HIP backward kernels) so ``objective.backward()`` of
reference trainers/trainer_ddpm.py:124-128 works unchanged on the drop-in modules.

Follows reference models/unet/unet.py:74-104 (train mode: Dropout(p) between the two Blocks of the down-path
ResnetBlocks, unet.py:46-47 vs :54-63) and models/downsampled/convblocks.py:112-159.
"""
import torch
from torch import nn

from ddk import autograd as AG
from ddk import ops
from ddk.plan import sinusoidal_freqs

HEADS = 4


class _Seeds:
    """Dropout seeds: one 62-bit seed per forward (from torch's CPU generator: reproducible under manual_seed),
    a distinct layer id per dropout site."""

    def __init__(self, training):
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0
        self.layer = 0

    def next(self):
        self.layer += 1
        return self.seed, self.layer


def _block(blk, x, x2=None, temb=None, addend=None, drop_p=0.0, seeds=None, give=None, take=None, dy_link=None, dx_link=None, give2=None):
    conv, norm = blk.block[0], blk.block[1]
    seed, layer = seeds.next() if (seeds is not None and drop_p > 0) else (0, 0)
    return AG.conv_groupnorm_mish(x, conv.weight, conv.bias, norm.weight, norm.bias, x2=x2, temb=temb, addend=addend, drop_p=drop_p,
                                  seed=seed, layer=layer, groups=blk.groups, eps=norm.eps, give=give, take=take, dy_link=dy_link,
                                  dx_link=dx_link, give2=give2)


def _resnet(rb, x, temb_slice, x2=None, seeds=None, give2=None):
    """blocks.py:105-115: h = drop(Block1(x) + shift); out = Block2(h) + res(x)"""
    p = rb.dropout.p if rb.training else 0.0
    # the gradient of the skip path reaches x (and x2) through Block1's input-gradient conv epilogue, not through an autograd add:
    # identity skip -> Block2 hands its addend gradient over; 1x1 skip -> the skip conv hands its input gradients over
    hand = AG.GradHandoff()
    # h has one consumer (Block2's conv): its gradient may travel from that conv's input-gradient launch to Block1's GroupNorm
    # backward as unreduced split-K slabs
    link = AG.SlabLink()
    h = _block(rb.block1, x, x2=x2, temb=temb_slice, drop_p=p, seeds=seeds, take=hand, dy_link=link, give2=give2)
    if isinstance(rb.res_conv, nn.Identity):
        return _block(rb.block2, h, addend=x, give=hand, dx_link=link)
    res = AG.conv(ops.CONV1X1, x, rb.res_conv.weight, rb.res_conv.bias, x2=x2, handoff=hand)
    return _block(rb.block2, h, addend=res, dx_link=link)


def _attention(res_mod, x):
    """Residual(PreNorm(LinearAttention)): blocks.py:8-14,63-71,126-134"""
    pre = res_mod.fn
    att = pre.fn
    hand = AG.GradHandoff()        # the Residual's gradient reaches x inside the LayerNorm backward kernel, not through an autograd add
    xn = AG.ChanLayerNormFn.apply(x, pre.norm.g, pre.norm.b, pre.norm.eps, hand)
    qkv = AG.conv(ops.CONV1X1, xn, att.to_qkv.weight)
    o = AG.LinAttnFn.apply(qkv, att.heads)
    return AG.conv(ops.CONV1X1, o, att.to_out.weight, att.to_out.bias, resid=x, resid_handoff=hand)


def _resnet_blocks(unet):
    """All ResnetBlocks in forward order (the order of the time-shift table's column blocks)."""
    blocks = []
    for lvl in unet.downs:
        blocks += [lvl[0], lvl[1]]
    blocks += [unet.mid_block1, unet.mid_block2]
    for lvl in unet.ups:
        blocks += [lvl[0], lvl[1]]
    return blocks


def unet_forward_autograd(unet, x, time):
    """x [B,H,W,C_in] (NHWC, unpadded), time [B] -> eps_hat [B,H,W,C_in], differentiable w.r.t. x and every parameter."""
    dev = x.device
    seeds = _Seeds(unet.training)
    rbs = _resnet_blocks(unet)
    freqs = getattr(unet, "_freqs_dev", None)
    if freqs is None or freqs.device != dev:
        freqs = sinusoidal_freqs(unet.dim).to(dev)
        unet._freqs_dev = freqs
    mlp_args = []
    for rb in rbs:
        mlp_args += [rb.mlp[1].weight, rb.mlp[1].bias]
    shifts = AG.TimeEmbedFn.apply(time.to(torch.int64).contiguous(), freqs, unet.time_mlp[1].weight, unet.time_mlp[1].bias,
                                  unet.time_mlp[3].weight, unet.time_mlp[3].bias, *mlp_args)
    shift_of = {id(rb): s for rb, s in zip(rbs, shifts)}      # one [B, C_out] column range per ResnetBlock

    def shift(rb):
        return shift_of[id(rb)]

    c_in = x.shape[-1]
    if c_in % 32:
        h = AG.NhwcToNchwFn.apply(x, c_in)                   # pad channels through the layout kernels (differentiable)
        h = AG.NchwToNhwcFn.apply(h, ops.pad32(c_in))
    else:
        h = x
    skips = []
    for lvl in unet.downs:
        rb1, rb2, attn, down = lvl
        h = _resnet(rb1, h, shift(rb1), seeds=seeds)
        h = _resnet(rb2, h, shift(rb2), seeds=seeds)
        h = _attention(attn, h)
        # the skip tensor has two consumers: the Downsample conv here and the up path's first Block; the latter's gradient is added by the
        # former's input-gradient launch (the last level has no Downsample: autograd adds there)
        carry = None if isinstance(down, nn.Identity) else AG.GradHandoff()
        skips.append((h, carry))
        if not isinstance(down, nn.Identity):
            h = AG.conv(ops.CONV3X3_S2, h, down.conv.weight, down.conv.bias, take=carry)
    h = _resnet(unet.mid_block1, h, shift(unet.mid_block1), seeds=seeds)
    h = _attention(unet.mid_attn, h)
    h = _resnet(unet.mid_block2, h, shift(unet.mid_block2), seeds=seeds)
    for lvl in unet.ups:
        rb1, rb2, attn, up = lvl
        skip, carry = skips.pop()
        h = _resnet(rb1, h, shift(rb1), x2=skip, seeds=seeds, give2=carry)
        h = _resnet(rb2, h, shift(rb2), seeds=seeds)
        h = _attention(attn, h)
        h = AG.conv(ops.CONVT4X4_S2, h, up.conv.weight, up.conv.bias)
    h = _block(unet.final_conv[0], h)
    last = unet.final_conv[1]
    return AG.SmallNConvFn.apply(h, last.weight, last.bias)


# ---------------------------------------------------------------- widths that are not multiples of 32 (blocks.py:75: any C % 8 == 0)
# Every tensor keeps a pitch of pad32(C) channels with zero padding; the conv family runs unchanged on zero-padded copies of the
# parameters (AG.pad_param: the padding rows / columns get no gradient back), GroupNorm and the channel LayerNorm on kernels that
# see the real channel count.  The tuned path's hand-offs (gradients riding on conv epilogues, split-K slabs summed by their
# consumer) are not used here: autograd adds where two gradients meet.  Correct and complete, untuned.
def _gblock(blk, x, c_in, x2=None, c_in2=None, temb=None, addend=None, drop_p=0.0, seeds=None):
    conv, norm = blk.block[0], blk.block[1]
    seed, layer = seeds.next() if (seeds is not None and drop_p > 0) else (0, 0)
    w = AG.pad_param(conv.weight, out_real=True, in_real=[c_in] if c_in2 is None else [c_in, c_in2])
    raw = AG.conv(ops.CONV3X3_S1, x, w, AG.pad_param(conv.bias, out_real=True), x2=x2)
    return AG.GNMishGenericFn.apply(raw, norm.weight, norm.bias, temb, addend, float(drop_p), int(seed), int(layer), blk.groups, norm.eps)


def _gresnet(rb, x, c_in, temb_slice, x2=None, c_in2=None, seeds=None):
    p = rb.dropout.p if rb.training else 0.0
    h = _gblock(rb.block1, x, c_in, x2=x2, c_in2=c_in2, temb=temb_slice, drop_p=p, seeds=seeds)
    c_out = rb.block1.block[0].weight.shape[0]
    if isinstance(rb.res_conv, nn.Identity):
        res = x
    else:
        w = AG.pad_param(rb.res_conv.weight, out_real=True, in_real=[c_in] if c_in2 is None else [c_in, c_in2])
        res = AG.conv(ops.CONV1X1, x, w, AG.pad_param(rb.res_conv.bias, out_real=True), x2=x2)
    return _gblock(rb.block2, h, c_out, addend=res), c_out


def _gattention(res_mod, x, c):
    pre = res_mod.fn
    att = pre.fn
    xn = AG.ChanLayerNormGenericFn.apply(x, pre.norm.g, pre.norm.b, pre.norm.eps)
    qkv = AG.conv(ops.CONV1X1, xn, AG.pad_param(att.to_qkv.weight, in_real=[c]))
    o = AG.LinAttnFn.apply(qkv, att.heads)
    return AG.conv(ops.CONV1X1, o, AG.pad_param(att.to_out.weight, out_real=True), AG.pad_param(att.to_out.bias, out_real=True), resid=x)


def unet_forward_autograd_generic(unet, x, time):
    """unet_forward_autograd for unet_chan % 8 == 0 that is not a multiple of 32 (24, 40, 72, ...): same network, padded pitch."""
    dev = x.device
    seeds = _Seeds(unet.training)
    rbs = _resnet_blocks(unet)
    freqs = getattr(unet, "_freqs_dev", None)
    if freqs is None or freqs.device != dev:
        freqs = sinusoidal_freqs(unet.dim).to(dev)
        unet._freqs_dev = freqs
    mlp_args = []
    for rb in rbs:
        mlp_args += [rb.mlp[1].weight, rb.mlp[1].bias]
    shifts = AG.TimeEmbedFn.apply(time.to(torch.int64).contiguous(), freqs, unet.time_mlp[1].weight, unet.time_mlp[1].bias,
                                  unet.time_mlp[3].weight, unet.time_mlp[3].bias, *mlp_args)
    shift_of = {id(rb): sft for rb, sft in zip(rbs, shifts)}      # [B, C_out] REAL channels per block (the GroupNorm kernel indexes them so)
    c = x.shape[-1]
    if c % 32:
        h = AG.NhwcToNchwFn.apply(x, c)
        h = AG.NchwToNhwcFn.apply(h, ops.pad32(c))
    else:
        h = x
    skips = []
    for lvl in unet.downs:
        rb1, rb2, attn, down = lvl
        h, c = _gresnet(rb1, h, c, shift_of[id(rb1)], seeds=seeds)
        h, c = _gresnet(rb2, h, c, shift_of[id(rb2)], seeds=seeds)
        h = _gattention(attn, h, c)
        skips.append((h, c))
        if not isinstance(down, nn.Identity):
            h = AG.conv(ops.CONV3X3_S2, h, AG.pad_param(down.conv.weight, out_real=True, in_real=[c]), AG.pad_param(down.conv.bias, out_real=True))
    h, c = _gresnet(unet.mid_block1, h, c, shift_of[id(unet.mid_block1)], seeds=seeds)
    h = _gattention(unet.mid_attn, h, c)
    h, c = _gresnet(unet.mid_block2, h, c, shift_of[id(unet.mid_block2)], seeds=seeds)
    for lvl in unet.ups:
        rb1, rb2, attn, up = lvl
        skip, cs = skips.pop()
        h, c = _gresnet(rb1, h, c, shift_of[id(rb1)], x2=skip, c_in2=cs, seeds=seeds)
        h, c = _gresnet(rb2, h, c, shift_of[id(rb2)], seeds=seeds)
        h = _gattention(attn, h, c)
        # ConvTranspose2d weight is (in, out, 4, 4): both channel axes padded
        h = AG.conv(ops.CONVT4X4_S2, h, AG.pad_param(up.conv.weight, out_real=True, in_real=[c]).contiguous(),
                    AG.pad_param(up.conv.bias, out_real=True))
    h = _gblock(unet.final_conv[0], h, c)
    last = unet.final_conv[1]
    return AG.SmallNConvFn.apply(h, AG.pad_param(last.weight, in_real=[c]), last.bias)


def sq_err_sum_autograd(a, b):
    """per-sample sum over C,H,W of (a-b)^2 (ddpm.py:279 + utils/utils.py:34-40)"""
    return AG.SqErrSumFn.apply(a.contiguous(), b.contiguous())


def _conv_res_block(blk, x, a=None, want_next_act=False):
    """convblocks.py:112-130 with the pre-activations kept for the backward: every conv's epilogue writes its output AND Mish of it
    (the next conv's input), every input-gradient conv multiplies by Mish' of the pre-activation -- one Mish launch per block (on the
    block input) instead of four forward and four backward ones; and none where the block before handed Mish of its output over
    (`a`, written by its c4 epilogue: want_next_act, for a block that neither pools nor upsamples).  -> (out, Mish(out) or None)"""
    if a is None:
        a = ops.mish(x.detach())
    drop_p = float(blk.drop.p) if blk.training else 0.0
    hand = AG.GradHandoff() if (blk.residual and drop_p == 0.0) else None   # the skip's gradient rides on c1's input-gradient conv (no 17-67 MB add launch)
    w1, b1, w2, b2, w3, b3, w4, b4 = blk.c1.weight, blk.c1.bias, blk.c2.weight, blk.c2.bias, blk.c3.weight, blk.c3.bias, blk.c4.weight, blk.c4.bias
    inner, outer = w1.shape[0], w1.shape[1]
    if inner % 32 or outer % 32:
        # d_chans = 48, 16, ... / d_chans / 2 = 16, 48, ...: tensors keep a pitch of pad32(channels) (zero weight rows and bias entries keep
        # the padding exactly zero through Mish, the convs and the residual add); the kernels see zero-padded copies of the parameters
        w1, b1 = AG.pad_param(w1, out_real=True, in_real=[outer]), AG.pad_param(b1, out_real=True)
        w2, b2 = AG.pad_param(w2, out_real=True, in_real=[inner]), AG.pad_param(b2, out_real=True)
        w3, b3 = AG.pad_param(w3, out_real=True, in_real=[inner]), AG.pad_param(b3, out_real=True)
        w4, b4 = AG.pad_param(w4, out_real=True, in_real=[inner]), AG.pad_param(b4, out_real=True)
    h, a = AG.preact_conv(ops.CONV1X1, x, a, w1, b1, handoff=hand)
    h, a = AG.preact_conv(ops.CONV3X3_S1, h, a, w2, b2)
    h, a = AG.preact_conv(ops.CONV3X3_S1, h, a, w3, b3)
    if drop_p > 0.0:
        # nn.Dropout2d on c4's output (convblocks.py:106,121-124): whole channels of a sample are zeroed with probability p, the others scaled
        # by 1 / (1 - p), BEFORE the residual.  train.py:36 uses d_dropout = 0, so this path is plain: c4 without its fused residual, the
        # mask as one broadcast multiply (torch elementwise kernels; autograd differentiates it), the residual as an add.
        x_hat, _ = AG.preact_conv(ops.CONV1X1, h, a, w4, b4, resid=None, want_act=False, handoff=None)
        mask = blk.channel_mask(x_hat.shape[0], x_hat.shape[-1], x_hat.device)
        x_hat = x_hat * mask[:, None, None, :]
        out, a_next = (x + x_hat if blk.residual else x_hat), None
    else:
        hand_over = want_next_act and not (blk.upsample or blk.downsample)
        out, a_next = AG.preact_conv(ops.CONV1X1, h, a, w4, b4, resid=x if blk.residual else None, want_act=hand_over, handoff=hand)
    if blk.upsample:
        out = AG.UpNearest2Fn.apply(out)
    elif blk.downsample:
        out = AG.AvgPool2Fn.apply(out)
    return out, a_next


def resnet_forward_autograd(net, x_nchw, final_tanh):
    """ConvResNet (dDDPM encoder / decoder) on an NCHW tensor -> NCHW, differentiable."""
    h = AG.NchwToNhwcFn.apply(x_nchw.contiguous().float(), ops.pad32(x_nchw.shape[1]))
    first, last = net.conv[0], net.conv[-1]
    wf, bf, wl = first.weight, first.bias, last.weight
    if net.dim % 32:
        # d_chans that is not a multiple of 32 (convblocks.py:133-159 takes any): the trunk keeps a pitch of pad32(d_chans) channels
        wf, bf = AG.pad_param(wf, out_real=True), AG.pad_param(bf, out_real=True)
        wl = AG.pad_param(wl, in_real=[net.dim])
    h = AG.conv(ops.CONV1X1, h, wf, bf)
    blocks = list(net.conv)[1:-1]
    a = None
    for k, blk in enumerate(blocks):
        h, a = _conv_res_block(blk, h, a, want_next_act=k + 1 < len(blocks))
    out = AG.SmallNConvFn.apply(h, wl, last.bias)
    if final_tanh:
        out = AG.TanhFn.apply(out)
    return AG.NhwcToNchwFn.apply(out, out.shape[-1])
