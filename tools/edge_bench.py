#!/usr/bin/env python3
"""Kernel-only timing (device-graph replay, HIP events) of the first / last kernels of a reverse step at cfg4:
conv_first (8 -> 128 @32x32, B = 32), GroupNorm-apply from partials with and without the on-the-fly 1x1 addend, the fused tail.
    python tools/edge_bench.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch  # noqa: E402

from bench import graph_kernel_seconds  # noqa: E402
from ddk import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    H = W = 32
    cin, N = 8, 128
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(N, cin, 3, 3, device=dev) * (cin * 9) ** -0.5
    b = torch.randn(N, device=dev)
    wf = ops.pack_conv_weight_first(w)
    raw, part, tiles = ops.conv_first(x, wf, b, N)
    gam, bet, temb = torch.ones(N, device=dev), torch.zeros(N, device=dev), torch.randn(B, N, device=dev)
    wr, br = torch.randn(N, cin, device=dev), torch.randn(N, device=dev)
    wo, bo = torch.randn(8, N, device=dev) * N ** -0.5, torch.randn(8, device=dev)
    xs = torch.randn(B, H, W, 8, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.long)
    tab = {k: torch.rand(1000, device=dev) for k in ("c_recip", "c_recipm1", "c1", "c2", "sigma")}
    z = torch.randn(B, H, W, 8, device=dev)

    def us(fn, n=50):
        return graph_kernel_seconds(dev, fn, n=n) * 1e6

    print(f"B = {B}")
    print(f"conv_first 8->128 + partials        {us(lambda: ops.conv_first(x, wf, b, N)):7.2f} us")
    print(f"conv_first 8->128 no partials       {us(lambda: ops.conv_first(x, wf, b, N, partials=False)):7.2f} us")
    print(f"conv_first 8->128 no bias           {us(lambda: ops.conv_first(x, wf, None, N)):7.2f} us")
    print(f"gn_apply parts (+temb)              {us(lambda: ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet, temb=temb)):7.2f} us")
    print(f"gn_apply parts + tensor addend      {us(lambda: ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet, addend=raw)):7.2f} us")
    print(f"gn_apply parts + res1x1 on the fly  {us(lambda: ops.groupnorm_mish_from_partials_res1x1(raw, part, tiles, gam, bet, x, wr, br)):7.2f} us")
    print(f"final_tail eps only                 {us(lambda: ops.final_tail(raw, part, tiles, gam, bet, wo, bo)):7.2f} us")
    print(f"final_tail + update (noise)         {us(lambda: ops.final_tail(raw, part, tiles, gam, bet, wo, bo, x=xs, t=t, tables=tab, noise=z, want_eps=False)):7.2f} us")
    print(f"final_tail + update (Philox)        {us(lambda: ops.final_tail(raw, part, tiles, gam, bet, wo, bo, x=xs, t=t, tables=tab, seed=3, want_eps=False)):7.2f} us")
    a1 = ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet)
    eps = ops.conv1x1_small_n(a1, wo, bo)
    print(f"unfused: conv1x1_n8                 {us(lambda: ops.conv1x1_small_n(a1, wo, bo)):7.2f} us")
    print(f"unfused: p_sample (Philox)          {us(lambda: ops.p_sample_update_(xs, eps, t, seed=3, **tab)):7.2f} us")


if __name__ == "__main__":
    main()
