#!/usr/bin/env python3
"""Weight-gradient timing (kernel + slab reduce) on the UNet shapes of the cfg3 training step (16x16 latents, batch 64) that run on
the per-tap kernel.  With the tuning build (DDK_LIB=.../libddk_tune.so) DDK_WGRAD_TILE / DDK_WGRAD_WAVES force the tile and the
target wave count.    python tools/wgrad_bench_cfg3.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops

SHAPES = [("1x1 128->384 @16", ops.CONV1X1, 64, 16, 128, 384), ("1x1 128->128 @16", ops.CONV1X1, 64, 16, 128, 128),
          ("1x1 256->384 @8", ops.CONV1X1, 64, 8, 256, 384), ("1x1 128->256 @8", ops.CONV1X1, 64, 8, 128, 256),
          ("3x3 256->256 @4", ops.CONV3X3_S1, 64, 4, 256, 256), ("3x3 256->256 @2", ops.CONV3X3_S1, 64, 2, 256, 256),
          ("3x3 512->256 @4", ops.CONV3X3_S1, 64, 4, 512, 256), ("1x1 256->384 @4", ops.CONV1X1, 64, 4, 256, 384)]
if os.environ.get("BIG"):      # the same kinds at larger pixel counts: where does the larger tile start to pay?
    SHAPES = [("1x1 128->384 @32", ops.CONV1X1, 64, 32, 128, 384), ("1x1 128->128 @32", ops.CONV1X1, 64, 32, 128, 128),
              ("1x1 128->384 @64", ops.CONV1X1, 64, 64, 128, 384), ("1x1 256->256 @32", ops.CONV1X1, 64, 32, 256, 256),
              ("1x1 128->384 @128", ops.CONV1X1, 8, 128, 128, 384), ("1x1 128->128 @256", ops.CONV1X1, 4, 256, 128, 128),
              ("3x3 256->256 @4 b256", ops.CONV3X3_S1, 256, 4, 256, 256)]
if os.environ.get("HALO"):     # the 3x3 shapes of the cfg3 / cfg2 steps that take the 64x64 halo kernel (DDK_NO_WGRAD_HALO=1: per-tap instead)
    SHAPES = [("3x3 256->256 @8", ops.CONV3X3_S1, 64, 8, 256, 256), ("3x3 128->128 @16", ops.CONV3X3_S1, 64, 16, 128, 128),
              ("3x3 128->256 @8", ops.CONV3X3_S1, 64, 8, 128, 256), ("3x3 512->256 @8", ops.CONV3X3_S1, 64, 8, 512, 256),
              ("3x3 256->256 @16", ops.CONV3X3_S1, 64, 16, 256, 256), ("3x3 128->128 @32", ops.CONV3X3_S1, 64, 32, 128, 128)]
tot = 0.0
for name, kind, B, H, C, N in SHAPES:
    k = 1 if kind == ops.CONV1X1 else 3
    x = torch.randn(B, H, H, C, device="cuda")
    dy = torch.randn(B, H, H, N, device="cuda")
    gw = torch.zeros(N, C, k, k, device="cuda")
    fn = lambda: ops.conv_wgrad_(kind, x, dy, gw, c_real=C, cw=C, c_off=0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    tot += us
    fl = 2.0 * B * H * H * k * k * C * N
    print(f"{name:18s} {fl / 1e9:6.2f} GF  {us:7.1f} us  {fl / us / 1e6:5.1f} TF (wgrad + slab reduce)", flush=True)
print(f"sum {tot:.1f} us   tile={os.environ.get('DDK_WGRAD_TILE')} waves={os.environ.get('DDK_WGRAD_WAVES')}")
