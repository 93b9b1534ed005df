"""dDDPM encoder / decoder networks (reference models/downsampled/convblocks.py:92-159), HIP-backed.

Only the 'convolutional_res' mode that train.py:34-35 selects is built (the reference's unused
SimpleDownConv / SimpleUpConv / interpolate modes are out of scope).  Parameter names
(`conv.{i}.c1..c4`, `conv.0`, `conv.{last}`) match the reference so checkpoints load strictly.

Kernel mapping of one ConvResBlock (convblocks.py:112-130), NHWC:
    c1: 1x1 on Mish(x)         -> igemm with Mish applied while staging the input, Mish in the epilogue
    c2, c3: 3x3                -> igemm, Mish in the epilogue (each activation is computed exactly once)
    c4: 1x1 (+ x residual)     -> igemm with the residual added in the epilogue
    avg_pool2d(2) / nearest x2 -> one elementwise kernel
"""
import torch.nn as nn

from ddk import ops
from ddk.lib import DDKError
from models.unet.blocks import _Packed, _to_nhwc


def get_3x3(in_dim, out_dim, stride=1, padding=1, padding_mode='zeros'):
    return nn.Conv2d(in_dim, out_dim, kernel_size=3, stride=stride, padding=padding, padding_mode=padding_mode)


def get_1x1(in_dim, out_dim):
    return nn.Conv2d(in_dim, out_dim, kernel_size=1, stride=1, padding=0)


class ConvResBlock(nn.Module):
    def __init__(self, dim, in_channels, out_channels=None, upsample=False, downsample=False, dropout=0, residual=False):
        super().__init__()
        assert not (upsample and downsample), 'Does not make sense to both down- and upsample.'
        if dropout:
            raise DDKError("ConvResBlock: d_dropout > 0 is not supported by the HIP path (train.py:36 uses 0)")
        self.upsample, self.downsample, self.residual = upsample, downsample, residual
        self.c1 = get_1x1(in_channels, dim)
        self.c2 = get_3x3(dim, dim)
        self.c3 = get_3x3(dim, dim)
        self.c4 = get_1x1(dim, out_channels)
        self.drop = nn.Dropout2d(p=dropout)
        self._packed = _Packed()

    def forward_nhwc(self, x):
        mid = self.c1.out_channels
        mp = ops.pad32(mid)
        if mp == mid:
            pk = lambda name, conv: self._packed.get(name, conv.weight, ops.pack_conv_weight)
            bs = lambda name, conv: conv.bias.detach()
        else:
            # d_chans / 2 is not a multiple of 32 (d_chans = 32, 96, ...): the block's inner tensors keep a pitch of pad32(mid) channels.
            # Output rows and bias entries beyond `mid` are zero, Mish(0) = 0, and the next conv's weights are zero on the padded
            # inputs (ddk_pack_conv_weight pads the input side itself): the padding stays exactly zero through the block.
            def pad_o(w):
                wp = ops.pack_conv_weight(w)
                if w.shape[0] == mid:
                    out = wp.new_zeros((mp,) + tuple(wp.shape[1:]))
                    out[:mid] = wp
                    return out
                return wp
            def pad_b(b):
                out = b.new_zeros(mp)
                out[:mid] = b
                return out
            pk = lambda name, conv: self._packed.get(name, conv.weight, pad_o)
            bs = lambda name, conv: self._packed.get(name + ".b", conv.bias, pad_b) if conv.out_channels == mid else conv.bias.detach()
        h = ops.conv(ops.CONV1X1, x, pk("c1", self.c1), bs("c1", self.c1), pre_mish=True, post_mish=True)
        h = ops.conv(ops.CONV3X3_S1, h, pk("c2", self.c2), bs("c2", self.c2), post_mish=True)
        h = ops.conv(ops.CONV3X3_S1, h, pk("c3", self.c3), bs("c3", self.c3), post_mish=True)
        out = ops.conv(ops.CONV1X1, h, pk("c4", self.c4), bs("c4", self.c4), resid=x if self.residual else None)
        if self.upsample:
            out = ops.upsample_nearest2(out)
        elif self.downsample:
            out = ops.avgpool2(out)
        return out

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))


class ConvResNet(nn.Module):
    """1x1 explode -> n_downsamples x [resampling block + (n_blocks-1) plain blocks] -> 1x1 condense
    (convblocks.py:133-159)."""

    def __init__(self, dim, in_channels, out_channels, n_downsamples=1, upsample=False, dropout=0, n_blocks=1):
        super().__init__()
        if dim % 32 != 0 or dim <= 0:
            raise DDKError("ConvResNet: d_chans must be a multiple of 32 for the HIP conv kernels (train.py:37 uses 64); training "
                           "additionally needs a multiple of 64")
        layers = [get_1x1(in_channels, dim)]
        for _ in range(n_downsamples):
            layers.append(ConvResBlock(int(dim / 2), dim, dim, upsample, not upsample, dropout, residual=True))
            for _ in range(int(n_blocks) - 1):
                layers.append(ConvResBlock(int(dim / 2), dim, dim, False, False, dropout, residual=True))
        layers.append(get_1x1(dim, out_channels))
        self.conv = nn.Sequential(*layers)
        self.in_channels, self.out_channels, self.dim = in_channels, out_channels, dim
        self._packed = _Packed()

    def invalidate_plan(self):
        self._packed._store.clear()
        for m in self.conv:
            pk = getattr(m, "_packed", None)
            if pk is not None:
                pk._store.clear()

    def flops(self, batch, height, width):
        """Algorithmic FLOPs (2*MAC) of one forward on a [batch, in_channels, height, width] input."""
        d = self.dim
        total, h, w = 2 * batch * height * width * self.in_channels * d, height, width
        for blk in list(self.conv)[1:-1]:
            m = d // 2
            total += 2 * batch * h * w * (d * m + 2 * 9 * m * m + m * d)
            if blk.upsample:
                h, w = 2 * h, 2 * w
            elif blk.downsample:
                h, w = h // 2, w // 2
        return total + 2 * batch * h * w * d * self.out_channels

    def forward_nhwc(self, x, final_tanh=False):
        """x [B,H,W,pad32(in_channels)] -> [B,H',W',out_channels]; optional fused-after tanh (dddpm.py:99,110)."""
        first, last = self.conv[0], self.conv[-1]
        w0 = self._packed.get("first", first.weight, ops.pack_conv_weight)
        h = ops.conv(ops.CONV1X1, x, w0, first.bias.detach())
        for blk in list(self.conv)[1:-1]:
            h = blk.forward_nhwc(h)
        out = ops.conv1x1_small_n(h, last.weight.detach().contiguous(), last.bias.detach())
        return ops.tanh(out) if final_tanh else out

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))
