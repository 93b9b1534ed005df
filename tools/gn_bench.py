#!/usr/bin/env python3
"""Kernel-only timing (device-graph replay) of the small fixed-cost kernels of a sampler step: GroupNorm+Mish, channel
LayerNorm, linear attention.  GPU-box tool: python tools/gn_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops

DEV = "cuda"


def graph_time(fn, n=40, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay(); side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            g.replay()
        e1.record(side); side.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


for (B, H, W, C) in [(32, 32, 32, 128), (32, 16, 16, 256), (32, 16, 16, 128), (32, 8, 8, 256), (32, 4, 4, 256)]:
    x = torch.randn(B, H, W, C, device=DEV)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    temb = torch.randn(B, C, device=DEV)
    add = torch.randn(B, H, W, C, device=DEV)
    t_gn = graph_time(lambda: ops.groupnorm_mish(x, g, b, temb=temb, addend=add))
    t_ln = graph_time(lambda: ops.chan_layernorm(x, g.view(1, C, 1, 1), b.view(1, C, 1, 1)))
    qkv = torch.randn(B, H, W, 384, device=DEV)
    t_at = graph_time(lambda: ops.linattn(qkv, 4))
    mb = B * H * W * C * 4 / 1e6
    print(f"{B}x{H}x{W}x{C}: GN+Mish(+temb+addend) {t_gn:6.1f} us ({3 * mb / t_gn:5.2f} TB/s)  LN {t_ln:6.1f} us ({2 * mb / t_ln:5.2f} TB/s)  "
          f"linattn ctx+apply {t_at:6.1f} us", flush=True)
