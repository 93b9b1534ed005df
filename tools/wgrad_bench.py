#!/usr/bin/env python3
"""Weight-gradient GEMM timing on the cfg2 training shapes (B=64).  GPU-box tool: python tools/wgrad_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops

SHAPES = [("3x3 128->128 @32", 64, 32, 32, 128, 128), ("3x3 256->256 @16", 64, 16, 16, 256, 256), ("3x3 256->256 @8", 64, 8, 8, 256, 256),
          ("3x3 256->256 @4", 64, 4, 4, 256, 256), ("3x3 512->256 @8", 64, 8, 8, 512, 256)]
for name, B, H, W, C, N in SHAPES:
    x = torch.randn(B, H, W, C, device="cuda")
    dy = torch.randn(B, H, W, N, device="cuda")
    gw = torch.zeros(N, C, 3, 3, device="cuda")
    fn = lambda: ops.conv_wgrad_(ops.CONV3X3_S1, x, dy, gw, c_real=C, cw=C, c_off=0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * B * H * W * 9 * C * N
    print(f"{name:18s} {fl / 1e9:6.2f} GF  {us:7.1f} us  {fl / us / 1e6:5.1f} TF (wgrad + slab reduce)", flush=True)
