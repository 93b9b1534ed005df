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
        self.upsample, self.downsample, self.residual = upsample, downsample, residual
        self.c1 = get_1x1(in_channels, dim)
        self.c2 = get_3x3(dim, dim)
        self.c3 = get_3x3(dim, dim)
        self.c4 = get_1x1(dim, out_channels)
        self.drop = nn.Dropout2d(p=dropout)
        self._packed = _Packed()
        self._mask_hook = None          # tests: callable (batch, channels, device) -> the Dropout2d mask to use (already scaled by 1 / (1 - p))

    def channel_mask(self, batch, channels_padded, device):
        """The nn.Dropout2d draw of one training forward (convblocks.py:106,121-124): [batch, channels] of {0, 1 / (1 - p)}, zero on the
        padding channels of a d_chans that is not a multiple of 32."""
        import torch
        p, c = float(self.drop.p), self.c4.out_channels
        if self._mask_hook is not None:
            m = self._mask_hook(batch, c, device).to(device=device, dtype=torch.float32)
        else:
            m = (torch.rand((batch, c), device=device) >= p).to(torch.float32) / (1.0 - p)
        if channels_padded != c:
            full = m.new_zeros((batch, channels_padded))
            full[:, :c] = m
            m = full
        return m

    def forward_nhwc(self, x):
        if self.training and self.drop.p > 0 and not x.requires_grad:
            # a train-mode forward outside autograd (reference: model.train() + torch.no_grad()): the differentiable path has the dropout
            from trainers.autograd_unet import _conv_res_block
            import torch
            with torch.no_grad():
                return _conv_res_block(self, x)[0]
        mid, outer = self.c1.out_channels, self.c1.in_channels
        mp = ops.pad32(mid)
        if mp == mid and outer % 32 == 0:
            pk = lambda name, conv: self._packed.get(name, conv.weight, ops.pack_conv_weight)
            bs = lambda name, conv: conv.bias.detach()
        else:
            # d_chans / 2 is not a multiple of 32 (d_chans = 32, 96, ...): the block's inner tensors keep a pitch of pad32(mid) channels.
            # Output rows and bias entries beyond `mid` are zero, Mish(0) = 0, and the next conv's weights are zero on the padded
            # inputs (ddk_pack_conv_weight pads the input side itself): the padding stays exactly zero through the block.
            # The same for d_chans itself (48, 16, ...): the block's input / output pitch is pad32(d_chans).
            def pad_o(w):
                wp = ops.pack_conv_weight(w)                    # pads the input side to 32 itself
                op = ops.pad32(w.shape[0])
                if op != w.shape[0]:
                    out = wp.new_zeros((op,) + tuple(wp.shape[1:]))
                    out[:w.shape[0]] = wp
                    return out
                return wp
            def pad_b(b):
                out = b.new_zeros(ops.pad32(b.shape[0]))
                out[:b.shape[0]] = b
                return out
            pk = lambda name, conv: self._packed.get(name, conv.weight, pad_o)
            bs = lambda name, conv: self._packed.get(name + ".b", conv.bias, pad_b)
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
        if dim % 2 != 0 or dim <= 0:
            raise DDKError("ConvResNet: d_chans must be a positive even number (its blocks are d_chans / 2 wide, convblocks.py:147); "
                           "multiples of 64 take the tuned kernels (train.py:37 uses 64), other widths run on zero-padded channel pitches")
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
        dp = ops.pad32(self.dim)
        if dp == self.dim:
            w0 = self._packed.get("first", first.weight, ops.pack_conv_weight)
            b0, wl = first.bias.detach(), last.weight.detach().contiguous()
        else:
            # d_chans not a multiple of 32: the trunk keeps a pitch of pad32(d_chans) channels, zero beyond d_chans
            def pad_first(w):
                wp = ops.pack_conv_weight(w)
                out = wp.new_zeros((dp,) + tuple(wp.shape[1:]))
                out[:self.dim] = wp
                return out
            def pad_bias(b):
                out = b.new_zeros(dp)
                out[:self.dim] = b
                return out
            def pad_last(w):
                out = w.new_zeros((w.shape[0], dp) + tuple(w.shape[2:]))
                out[:, :self.dim] = w
                return out.contiguous()
            w0 = self._packed.get("first", first.weight, pad_first)
            b0 = self._packed.get("first.b", first.bias, pad_bias)
            wl = self._packed.get("last", last.weight, pad_last)
        h = ops.conv(ops.CONV1X1, x, w0, b0)
        for blk in list(self.conv)[1:-1]:
            h = blk.forward_nhwc(h)
        out = ops.conv1x1_small_n(h, wl, last.bias.detach())
        return ops.tanh(out) if final_tanh else out

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(_to_nhwc(x)))
