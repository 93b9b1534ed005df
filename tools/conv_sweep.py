#!/usr/bin/env python3
"""Kernel-only (device-graph replay) sweep of tile / split-K / ring depth for every conv shape of one cfg4 reverse step (B=32).

3x3 stride-1 shapes are timed as the production pair conv + GroupNorm(+Mish) -- the GroupNorm reads the split-K slabs, so the
cost of more splits shows up where it is paid; the other kinds as conv (+ its reduce kernel when split).
Needs the tuning build:   make -C downsampled-diffusion_amd/csrc tune
    DDK_LIB=downsampled-diffusion_amd/csrc/libddk_tune.so python tools/conv_sweep.py [--quick] [--only SUBSTR]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
import torch  # noqa: E402
from ddk import ops  # noqa: E402

TILES = {0: "128x128", 1: "128x64", 2: "64x64"}
STAGES = {0: (2,), 1: (2, 4), 2: (2, 4, 6)}
S1, S2, P1, TC = ops.CONV3X3_S1, ops.CONV3X3_S2, ops.CONV1X1, ops.CONVT4X4_S2
SHAPES = [  # name, kind, H, c0, c1, N, count per step
    ("3x3  32->128 @32", S1, 32, 32, 0, 128, 1),
    ("3x3 128->128 @32", S1, 32, 128, 0, 128, 4),
    ("3x3 128->256 @16", S1, 16, 128, 0, 256, 1),
    ("3x3 256->256 @16", S1, 16, 256, 0, 256, 3),
    ("3x3 512->128 @16", S1, 16, 256, 256, 128, 1),
    ("3x3 128->128 @16", S1, 16, 128, 0, 128, 3),
    ("3x3 256->256 @8", S1, 8, 256, 0, 256, 7),
    ("3x3 512->256 @8", S1, 8, 256, 256, 256, 1),
    ("3x3 256->256 @4", S1, 4, 256, 0, 256, 11),
    ("3x3 512->256 @4", S1, 4, 256, 256, 256, 1),
    ("s2 128->128 @32", S2, 32, 128, 0, 128, 1),
    ("s2 256->256 @16", S2, 16, 256, 0, 256, 1),
    ("s2 256->256 @8", S2, 8, 256, 0, 256, 1),
    ("T 256->256 @4", TC, 4, 256, 0, 256, 1),
    ("T 256->256 @8", TC, 8, 256, 0, 256, 1),
    ("T 128->128 @16", TC, 16, 128, 0, 128, 1),
    ("1x1 128->384 @32", P1, 32, 128, 0, 384, 1),
    ("1x1 128->128 @32", P1, 32, 128, 0, 128, 1),
    ("1x1  32->128 @32", P1, 32, 32, 0, 128, 1),
    ("1x1 256->384 @16", P1, 16, 256, 0, 384, 1),
    ("1x1 128->384 @16", P1, 16, 128, 0, 384, 1),
    ("1x1 128->256 @16", P1, 16, 128, 0, 256, 2),
    ("1x1 128->128 @16", P1, 16, 128, 0, 128, 1),
    ("1x1 512->128 @16", P1, 16, 256, 256, 128, 1),
    ("1x1 256->384 @8", P1, 8, 256, 0, 384, 2),
    ("1x1 128->256 @8", P1, 8, 128, 0, 256, 2),
    ("1x1 512->256 @8", P1, 8, 256, 256, 256, 1),
    ("1x1 256->384 @4", P1, 4, 256, 0, 384, 3),
    ("1x1 128->256 @4", P1, 4, 128, 0, 256, 3),
    ("1x1 512->256 @4", P1, 4, 256, 256, 256, 1),
]


def graph_time(fn, n=20, reps=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            g.replay()
        e1.record(side)
        side.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def flops(kind, B, H, cin, N):
    if kind == S1:
        return 2.0 * B * H * H * 9 * cin * N
    if kind == S2:
        return 2.0 * B * (H // 2) * (H // 2) * 9 * cin * N
    if kind == P1:
        return 2.0 * B * H * H * cin * N
    return 2.0 * B * H * H * 16 * cin * N


def main():
    quick = "--quick" in sys.argv
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    dev, B = "cuda", 32
    total_auto = 0.0
    for name, kind, H, c0, c1, N, count in SHAPES:
        if only and only not in name:
            continue
        cin = c0 + c1
        x0 = torch.randn(B, H, H, c0, device=dev)
        x1 = torch.randn(B, H, H, c1, device=dev) if c1 else None
        if kind == TC:
            wp = ops.pack_convT_weight(torch.randn(cin, N, 4, 4, device=dev) * 0.02)
        else:
            k = 1 if kind == P1 else 3
            wp = ops.pack_conv_weight(torch.randn(N, cin, k, k, device=dev) * 0.02)
        bias = torch.zeros(N, device=dev)
        gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        if kind == S1:
            fn = lambda: ops.conv3x3_groupnorm_mish(x0, wp, bias, gamma, beta, x2=x1)   # noqa: E731
        else:
            fn = lambda: ops.conv(kind, x0, wp, bias, x2=x1)   # noqa: E731
        fl = flops(kind, B, H, cin, N)
        os.environ.pop("DDK_FORCE_TILE", None)
        us = graph_time(fn)
        total_auto += us * count
        line = f"{name:18s} x{count:2d} {fl / 1e9:6.3f} GF auto {us:6.1f}us {fl / us / 1e6:5.1f}TF |"
        if not quick:
            res = []
            if kind == S1:      # the halo kernel (eligible shapes only; splits = channel-chunk splits), then the im2col tiles without it
                os.environ["DDK_HALO_MIN_CHUNKS"] = "1"
                for s in (1, 2, 4, 8):
                    os.environ["DDK_FORCE_TILE"] = f"0,{s},0"
                    try:
                        res.append((graph_time(fn, n=10, reps=2), f"halo?/{s}"))
                    except Exception:  # noqa: BLE001
                        pass
                os.environ.pop("DDK_HALO_MIN_CHUNKS", None)
                os.environ["DDK_NO_HALO"] = "1"
            for t in (0, 1, 2):
                for s in (1, 2, 4, 8, 16):
                    for st in STAGES[t]:
                        os.environ["DDK_FORCE_TILE"] = f"{t},{s},{st}"
                        try:
                            res.append((graph_time(fn, n=10, reps=2), f"{TILES[t]}/{s}/s{st}"))
                        except Exception:  # noqa: BLE001
                            pass
            os.environ.pop("DDK_FORCE_TILE", None)
            os.environ.pop("DDK_NO_HALO", None)
            res.sort()
            line += "  best: " + "  ".join(f"{nm} {u:5.1f}" for u, nm in res[:6])
        print(line, flush=True)
    print(f"sum over the step's conv launches (auto): {total_auto:.1f} us", flush=True)


if __name__ == "__main__":
    main()
