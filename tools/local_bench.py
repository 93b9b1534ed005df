#!/usr/bin/env python3
"""Kernel-only (graph replay) timing of the one-launch conv3x3+GroupNorm+Mish kernel (conv_local.hip) against the two-launch
path (Winograd conv with channel-chunk slabs, then GroupNorm summing the slabs) on the 4x4 / 8x8 shapes of a cfg4 reverse step.
GPU-box tool: python tools/local_bench.py [B=32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from ddk import ops  # noqa: E402
from conv_sweep import graph_time  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [("256->256 @4", 4, 256, 0, 256), ("512->256 @4", 4, 256, 256, 256), ("256->256 @8", 8, 256, 0, 256), ("512->256 @8", 8, 256, 256, 256)]
for name, H, c0, c1, N in SHAPES:
    cin = c0 + c1
    x0 = torch.randn(B, H, H, c0, device="cuda")
    x1 = torch.randn(B, H, H, c1, device="cuda") if c1 else None
    w = torch.randn(N, cin, 3, 3, device="cuda") * (cin * 9) ** -0.5
    wp, wu, wl = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w), ops.pack_conv_weight_local(w)
    bias = torch.randn(N, device="cuda") * 0.1
    gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    temb = torch.randn(B, N, device="cuda")
    t_two = graph_time(lambda: ops.conv3x3_groupnorm_mish(x0, wp, bias, gam, bet, x2=x1, temb=temb, w_wino=wu))
    t_one = graph_time(lambda: ops.conv3x3_gn_mish(x0, wl, bias, gam, bet, temb=temb, x2=x1))
    a = ops.conv3x3_groupnorm_mish(x0, wp, bias, gam, bet, x2=x1, temb=temb, w_wino=wu)
    b = ops.conv3x3_gn_mish(x0, wl, bias, gam, bet, temb=temb, x2=x1)
    if H == 8 and ops.L.load().ddk_conv3x3_gn_mish_wino_ok(H, H, cin, c0, N, 8):
        wwl = ops.pack_conv_weight_wino_local(w)
        t_w = graph_time(lambda: ops.conv3x3_gn_mish_wino(x0, wwl, bias, gam, bet, temb=temb, x2=x1))
        cw = ops.conv3x3_gn_mish_wino(x0, wwl, bias, gam, bet, temb=temb, x2=x1)
        print(f"{name:12s} B={B}: one launch, Winograd form {t_w:6.1f} us   max|diff| {float((a - cw).abs().max()):.2e}", flush=True)
    print(f"{name:12s} B={B}: two launches {t_two:6.1f} us   one launch {t_one:6.1f} us   max|diff| {float((a - b).abs().max()):.2e}", flush=True)
