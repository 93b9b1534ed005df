"""Epsilon-prediction UNet with the reference's constructor, forward signature and state_dict layout
(reference models/unet/unet.py:9-104), evaluated by the native HIP plan.

``Unet(config)`` reads ``unet_chan, unet_in, unet_dims, unet_dropout`` (unet.py:19-22).  The module tree
below only owns the canonical parameters (232 tensors at the default config).
``forward(x, time)`` converts NCHW -> NHWC, runs the whole network as one C call
(ddk_unet_forward: ~150 kernel launches sequenced natively) and converts back.
"""
import torch
from torch import nn

from ddk import ops
from ddk.lib import DDKError
from ddk.plan import UnetPlan
from .blocks import Block, Downsample, LinearAttention, PreNorm, Residual, ResnetBlock, SinusoidalPosEmb, Upsample


class Unet(nn.Module):
    def __init__(self, config: dict):
        super().__init__()
        dim = config['unet_chan']
        in_channels = config['unet_in']
        dim_mults = tuple(config['unet_dims'])
        dropout = config['unet_dropout']
        if dim % 8 != 0 or not 8 <= dim <= 512:
            # the reference's GroupNorm(8, C) (blocks.py:75) needs C % 8 == 0 as well; multiples of 32 run the tuned kernels, other
            # multiples of 8 the generic ones (channels padded to 32 with zeros: correct, untuned; inference and training)
            raise DDKError(f"unet_chan={dim}: must be a multiple of 8 in [8, 512] (GroupNorm(8, C), reference blocks.py:75)")
        self.dim, self.in_channels, self.dim_mults = dim, in_channels, dim_mults

        dims = [in_channels] + [dim * m for m in dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        n_res = len(in_out)

        # unet.py:30-35 -- registration order (time_mlp, downs, ups, mid, final) matches the reference
        self.time_mlp = nn.Sequential(SinusoidalPosEmb(dim), nn.Linear(dim, dim * 4), nn.Mish(), nn.Linear(dim * 4, dim))
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        for i, (c_in, c_out) in enumerate(in_out):           # unet.py:43-50
            last = i >= n_res - 1
            self.downs.append(nn.ModuleList([
                ResnetBlock(c_in, c_out, time_emb_dim=dim, dropout=dropout),
                ResnetBlock(c_out, c_out, time_emb_dim=dim, dropout=dropout),
                Residual(PreNorm(c_out, LinearAttention(c_out))),
                Downsample(c_out) if not last else nn.Identity(),
            ]))
        mid = dims[-1]                                        # unet.py:53-56
        self.mid_block1 = ResnetBlock(mid, mid, time_emb_dim=dim)
        self.mid_attn = Residual(PreNorm(mid, LinearAttention(mid)))
        self.mid_block2 = ResnetBlock(mid, mid, time_emb_dim=dim)
        for c_in, c_out in reversed(in_out[1:]):              # unet.py:59-66: every up level upsamples (F5)
            self.ups.append(nn.ModuleList([
                ResnetBlock(c_out * 2, c_in, time_emb_dim=dim),
                ResnetBlock(c_in, c_in, time_emb_dim=dim),
                Residual(PreNorm(c_in, LinearAttention(c_in))),
                Upsample(c_in),
            ]))
        self.final_conv = nn.Sequential(Block(dim, dim), nn.Conv2d(dim, in_channels, 1))   # unet.py:69-72

        self._plan = None
        self._plan_tag = None

    # ------------------------------------------------------------------ native plan plumbing
    def _weights_tag(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def plan(self) -> UnetPlan:
        """The native plan with packed weights current for the parameters as they are now."""
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise DDKError("Unet: parameters are on the CPU; the HIP path needs model.to('cuda') (no CPU fallback)")
        if self._plan is None:
            self._plan = UnetPlan(self.in_channels, self.dim, self.dim_mults)
        tag = self._weights_tag()
        if tag != self._plan_tag:
            self._plan.pack({k: v for k, v in self.state_dict().items()}, p0.device)
            self._plan_tag = tag
        return self._plan

    def invalidate_plan(self):
        """Force a re-pack of the weights at the next forward (used after in-place optimiser / EMA kernels, which
        update the parameters behind torch's version counters)."""
        self._plan_tag = None
        for m in self.modules():
            pk = getattr(m, "_packed", None)
            if pk is not None:
                pk._store.clear()

    def flops(self, batch, height, width):
        """Algorithmic FLOPs (2*MAC) of one forward."""
        if self._plan is None:
            self._plan = UnetPlan(self.in_channels, self.dim, self.dim_mults)
        return self._plan.flops(batch, height, width)

    def flops_executed(self, batch, height, width):
        """FLOPs the dispatched HIP kernels issue for one forward (Winograd convs: 16/36 of the direct multiplies)."""
        if self._plan is None:
            self._plan = UnetPlan(self.in_channels, self.dim, self.dim_mults)
        return self._plan.flops_executed(batch, height, width)

    # ------------------------------------------------------------------ forward
    def _wants_grad(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward_nhwc(self, x, time):
        """x [B,H,W,C_in] fp32 on the device, time [B] integer -> eps_hat [B,H,W,C_in]."""
        if self._wants_grad(x):
            if self.dim % 32 != 0:
                # other multiples of 8 (blocks.py:75): the same HIP kernels on a zero-padded channel pitch, generic normalisations
                from trainers.autograd_unet import unet_forward_autograd_generic
                return unet_forward_autograd_generic(self, x, time)
            from trainers.autograd_unet import unet_forward_autograd   # training path: HIP forward + backward kernels
            return unet_forward_autograd(self, x, time)
        if self.training and self.downs[0][0].dropout.p > 0:
            raise DDKError("Unet in train() mode with dropout > 0 outside autograd: call model.eval() for inference")
        return self.plan().forward_nhwc(x.contiguous(), time.to(torch.int64).contiguous())

    def forward(self, x, time):
        """unet.py:74-104: x B x C x H x W, time B -> B x C x H x W."""
        if not x.is_cuda:
            raise DDKError("Unet.forward: input is on the CPU; the HIP path needs ROCm device tensors (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise DDKError(f"Unet.forward: expected B x {self.in_channels} x H x W, got {tuple(x.shape)}")
        if self._wants_grad(x):
            from ddk import autograd as AG
            y = self.forward_nhwc(AG.NchwToNhwcFn.apply(x.contiguous().float(), x.shape[1]), time)
            return AG.NhwcToNchwFn.apply(y, y.shape[-1])
        y = self.forward_nhwc(ops.nchw_to_nhwc(x.contiguous().float()), time)
        return ops.nhwc_to_nchw(y)
