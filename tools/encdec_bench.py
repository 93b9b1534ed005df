#!/usr/bin/env python3
"""Kernel-only timings (graph replay) of the dDDPM encoder / decoder conv shapes of cfg3 training (d_chans 64, B = 64):
the 3x3 32->32 convs with the Mish epilogues of the training path, the 1x1 64->32 / 32->64 convs.  python tools/encdec_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops

DEV = "cuda"


def graph_us(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for H in (64, 32, 16):
    for (kind, ci, co, k) in ((ops.CONV3X3_S1, 32, 32, 3), (ops.CONV1X1, 64, 32, 1), (ops.CONV1X1, 32, 64, 1)):
        x = torch.randn(B, H, H, ci, device=DEV)
        w = torch.randn(co, ci, k, k, device=DEV) * 0.05
        wp = ops.pack_conv_weight(w)
        wd = ops.pack_conv_weight_dgrad(w, i_pad=ci)
        bias = torch.randn(co, device=DEV)
        a_out = torch.empty(B, H, H, co, device=DEV)
        hsrc = torch.randn(B, H, H, ci, device=DEV)
        dy = torch.randn(B, H, H, co, device=DEV)
        res = torch.randn(B, H, H, co, device=DEV)
        fl = 2.0 * B * H * H * k * k * ci * co
        by = 4.0 * B * H * H * (ci + co)
        t_plain = graph_us(lambda: ops.conv(kind, x, wp, bias))
        t_mo = graph_us(lambda: ops.conv(kind, x, wp, bias, mish_out=a_out))
        t_dg = graph_us(lambda: ops.conv(kind, dy, wd, None, n_out=ci, dmish_src=hsrc))
        t_rs = graph_us(lambda: ops.conv(kind, x, wp, bias, resid=res))
        print(f"{k}x{k} {ci:3d}->{co:3d} @{H:2d}x{H:2d} B={B}: {fl / 1e9:6.2f} GF, {by / 1e6:6.1f} MB in+out | plain {t_plain:7.1f} us ({fl / t_plain / 1e6:5.1f} TF, "
              f"{by / t_plain / 1e3:6.0f} GB/s) | +mish_out {t_mo:7.1f} | dgrad*mish' {t_dg:7.1f} | +resid {t_rs:7.1f}", flush=True)
