#!/usr/bin/env python3
"""Kernel-only timing of the training GroupNorm forward / backward on the UNet's shapes at batch 64 (graph replay of 20 launches):
python tools/gn_bwd_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops

dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B, H, C in ((64, 32, 128), (64, 16, 128), (64, 16, 256), (64, 8, 256), (64, 4, 256), (64, 2, 256), (32, 32, 128)):
    x = torch.randn(B, H, H, C, device=dev)
    dy = torch.randn(B, H, H, C, device=dev)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    te = torch.randn(B, C, device=dev)
    mb = x.numel() * 4 / 1e6
    tf = timed(lambda: ops.groupnorm_mish_train(x, ga, be, temb=te))
    tb = timed(lambda: ops.groupnorm_mish_bwd(x, ga, be, dy))
    print(f"B={B} {H}x{H} C={C}: {mb:6.1f} MB | fwd {tf:6.1f} us ({2 * mb / tf:5.2f} TB/s) | bwd {tb:6.1f} us ({3 * mb / tb:5.2f} TB/s)")
