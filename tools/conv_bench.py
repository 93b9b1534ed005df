#!/usr/bin/env python3
"""Times the implicit-GEMM conv on the cfg4 layer shapes (B=32) for each tile / split-K choice.
GPU-box tool for kernel tuning: python tools/conv_bench.py [--quick]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch  # noqa: E402
from ddk import ops  # noqa: E402

TILES = {0: "128x128", 1: "128x64", 2: "64x64", 3: "128x32", 4: "64x32"}
SHAPES = [  # name, kind, B, H, W, c0, c1, N
    ("3x3 128->128 @32", ops.CONV3X3_S1, 32, 32, 32, 128, 0, 128),
    ("3x3 256->256 @16", ops.CONV3X3_S1, 32, 16, 16, 256, 0, 256),
    ("3x3 512->128 @16", ops.CONV3X3_S1, 32, 16, 16, 256, 256, 128),
    ("3x3 128->128 @16", ops.CONV3X3_S1, 32, 16, 16, 128, 0, 128),
    ("3x3 128->256 @16", ops.CONV3X3_S1, 32, 16, 16, 128, 0, 256),
    ("3x3 256->256 @8", ops.CONV3X3_S1, 32, 8, 8, 256, 0, 256),
    ("3x3 512->256 @8", ops.CONV3X3_S1, 32, 8, 8, 256, 256, 256),
    ("3x3 256->256 @4", ops.CONV3X3_S1, 32, 4, 4, 256, 0, 256),
    ("3x3 512->256 @4", ops.CONV3X3_S1, 32, 4, 4, 256, 256, 256),
    ("1x1 128->384 @32", ops.CONV1X1, 32, 32, 32, 128, 0, 384),
    ("1x1 128->128 @32", ops.CONV1X1, 32, 32, 32, 128, 0, 128),
    ("1x1 256->384 @8", ops.CONV1X1, 32, 8, 8, 256, 0, 384),
    ("1x1 256->384 @16", ops.CONV1X1, 32, 16, 16, 256, 0, 384),
    ("1x1 256->384 @4", ops.CONV1X1, 32, 4, 4, 256, 0, 384),
    ("s2 128->128 @32", ops.CONV3X3_S2, 32, 32, 32, 128, 0, 128),
    ("s2 256->256 @16", ops.CONV3X3_S2, 32, 16, 16, 256, 0, 256),
    ("s2 256->256 @8", ops.CONV3X3_S2, 32, 8, 8, 256, 0, 256),
    ("T 128->128 @16", ops.CONVT4X4_S2, 32, 16, 16, 128, 0, 128),
    ("T 256->256 @8", ops.CONVT4X4_S2, 32, 8, 8, 256, 0, 256),
    ("T 256->256 @4", ops.CONVT4X4_S2, 32, 4, 4, 256, 0, 256),
]


def flops(kind, B, H, W, cin, N):
    if kind == ops.CONV3X3_S1:
        return 2.0 * B * H * W * 9 * cin * N
    if kind == ops.CONV3X3_S2:
        return 2.0 * B * (H // 2) * (W // 2) * 9 * cin * N
    if kind == ops.CONV1X1:
        return 2.0 * B * H * W * cin * N
    return 2.0 * B * H * W * 4 * 4 * cin * N


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    quick = "--quick" in sys.argv
    dev = "cuda"
    for name, kind, B, H, W, c0, c1, N in SHAPES:
        cin = c0 + c1
        x0 = torch.randn(B, H, W, c0, device=dev)
        x1 = torch.randn(B, H, W, c1, device=dev) if c1 else None
        if kind == ops.CONVT4X4_S2:
            wp = ops.pack_convT_weight(torch.randn(cin, N, 4, 4, device=dev) * 0.02)
        else:
            k = 1 if kind == ops.CONV1X1 else 3
            wp = ops.pack_conv_weight(torch.randn(N, cin, k, k, device=dev) * 0.02)
        bias = torch.zeros(N, device=dev)
        fl = flops(kind, B, H, W, cin, N)
        os.environ.pop("DDK_FORCE_TILE", None)
        us = timeit(lambda: ops.conv(kind, x0, wp, bias, x2=x1))
        line = f"{name:18s} {fl / 1e9:6.3f} GF auto {us:6.1f}us {fl / us / 1e6:5.1f}TF |"
        if not quick:
            for t in (0, 1, 2):
                for s in (1, 2, 4, 8, 16):
                    os.environ["DDK_FORCE_TILE"] = f"{t},{s}"
                    try:
                        u = timeit(lambda: ops.conv(kind, x0, wp, bias, x2=x1), n=15)
                        line += f" {TILES[t]}/{s}:{fl / u / 1e6:5.1f}"
                    except Exception:  # noqa: BLE001
                        line += f" {TILES[t]}/{s}: err"
            os.environ.pop("DDK_FORCE_TILE", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
