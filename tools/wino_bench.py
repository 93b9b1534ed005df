#!/usr/bin/env python3
"""Kernel-only (graph replay) timing of the Winograd F(2x2,3x3) conv against the direct kernels on the 3x3 shapes of one cfg4
reverse step (B=32).  GPU-box tool: python tools/wino_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from ddk import ops  # noqa: E402
from conv_sweep import graph_time  # noqa: E402

SHAPES = [  # name, H, c0, c1, N, count per step
    ("3x3 128->128 @32", 32, 128, 0, 128, 4), ("3x3 128->256 @16", 16, 128, 0, 256, 1), ("3x3 256->256 @16", 16, 256, 0, 256, 3),
    ("3x3 512->128 @16", 16, 256, 256, 128, 1), ("3x3 128->128 @16", 16, 128, 0, 128, 3), ("3x3 256->256 @8", 8, 256, 0, 256, 7),
    ("3x3 512->256 @8", 8, 256, 256, 256, 1), ("3x3 256->256 @4", 4, 256, 0, 256, 11), ("3x3 512->256 @4", 4, 256, 256, 256, 1),
    ("3x3  32->128 @32", 32, 32, 0, 128, 1),
]
B = 32
tot_d = tot_w = 0.0
for name, H, c0, c1, N, count in SHAPES:
    cin = c0 + c1
    x0 = torch.randn(B, H, H, c0, device="cuda")
    x1 = torch.randn(B, H, H, c1, device="cuda") if c1 else None
    w = torch.randn(N, cin, 3, 3, device="cuda") * (cin * 9) ** -0.5
    wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w)
    bias = torch.zeros(N, device="cuda")
    gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    td = graph_time(lambda: ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1))
    tw = graph_time(lambda: ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1, w_wino=wu))
    err = float((ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1) - ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1, w_wino=wu)).abs().max())
    fl = 2.0 * B * H * H * 9 * cin * N
    sp = ops.L.load().ddk_conv_wino_splits(B, H, H, cin, N)
    tot_d += td * count
    tot_w += tw * count
    print(f"{name:18s} x{count:2d} {fl / 1e9:6.3f} GF  direct {td:6.1f} us ({fl / td / 1e6:5.1f} TF)   winograd {tw:6.1f} us "
          f"({fl / tw / 1e6:5.1f} TF algorithmic, splits {sp})   max|diff| {err:.2e}", flush=True)
print(f"sum over the step's 3x3 launches (conv + its reduce when split): direct {tot_d:.0f} us, winograd {tot_w:.0f} us")
