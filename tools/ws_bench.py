#!/usr/bin/env python3
"""Kernel-only timing (device-graph replay, HIP events) of the weights-stationary 1x1 conv (conv1x1_ws.hip) at the cfg4 shapes it
takes in a reverse step, and of the im2col kernel on the same shapes when the library is the tuning build run with
DDK_NO_CONV1X1_WS=1 (DDK_LIB=.../libddk_tune.so).
    python tools/ws_bench.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch  # noqa: E402

from bench import graph_kernel_seconds  # noqa: E402
from ddk import ops  # noqa: E402

CASES = [  # name, H, N, bias, resid, ln
    ("to_qkv 32x32 128->384 (LayerNorm folded)", 32, 384, False, False, True),
    ("to_out 32x32 128->128 + bias + residual", 32, 128, True, True, False),
    ("res_conv 16x16 128->256 + bias", 16, 256, True, False, False),
    ("to_out 16x16 128->256 + bias + residual", 16, 256, True, True, False),
    ("to_qkv 16x16 128->384 (LayerNorm folded)", 16, 384, False, False, True),
    ("to_out 16x16 128->128 + bias + residual", 16, 128, True, True, False),
]


PROBES = [  # what a tile costs: the same slice count with and without the LayerNorm pass / bias / residual, even and ragged tile counts
    ("probe 32x32 128->384 plain", 32, 384, False, False, False),
    ("probe 32x32 128->256 plain (4 tiles each)", 32, 256, False, False, False),
    ("probe 32x32 128->256 LayerNorm (4 tiles each)", 32, 256, False, False, True),
    ("probe 32x32 128->128 plain (2 tiles each)", 32, 128, False, False, False),
    ("probe 32x32 128->512 plain (8 tiles each)", 32, 512, False, False, False),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    if len(sys.argv) > 2 and sys.argv[2] == "probe":
        CASES.extend(PROBES)
    dev = torch.device("cuda", 0)
    K = 128
    print(f"B = {B}; MFMA-bound time = 2 M N K / 157.3 TFLOP/s")
    for name, H, N, ub, ur, ul in CASES:
        x = torch.randn(B, H, H, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.randn(N, device=dev) if ub else None
        r = torch.randn(B, H, H, N, device=dev) if ur else None
        ln = (torch.randn(N, device=dev), torch.randn(N, device=dev)) if ul else None
        t_ws = graph_kernel_seconds(dev, lambda: ops.conv1x1_ws(x, w, b, r, ln), n=50) * 1e6
        line = f"{name:44s} ws {t_ws:7.2f} us"
        if not ul:
            wp = w.reshape(N, 1, K).contiguous()
            t_g = graph_kernel_seconds(dev, lambda: ops.conv(ops.CONV1X1, x, wp, b, resid=r), n=50) * 1e6
            line += f" | generic entry {t_g:7.2f} us"
        line += f" | MFMA-bound {2.0 * B * H * H * N * K / 157.3e12 * 1e6:6.2f} us"
        print(line)


if __name__ == "__main__":
    main()
