#!/usr/bin/env python3
"""Launch one conv shape N times (for rocprofv3 counter runs): python tools/conv_one.py [tile,splits] [shape idx]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops
if len(sys.argv) > 1 and sys.argv[1] != "auto":
    os.environ["DDK_FORCE_TILE"] = sys.argv[1]
B, H, W, C, N = 32, 32, 32, 128, 128
if len(sys.argv) > 2 and sys.argv[2] == "16":
    B, H, W, C, N = 32, 16, 16, 256, 256
x = torch.randn(B, H, W, C, device="cuda")
wp = ops.pack_conv_weight(torch.randn(N, C, 3, 3, device="cuda") * 0.03)
b = torch.zeros(N, device="cuda")
for _ in range(20):
    y = ops.conv(ops.CONV3X3_S1, x, wp, b)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
