#!/usr/bin/env python3
"""Launch one conv shape N times (for rocprofv3 counter runs): python tools/conv_one.py [tile,splits|auto] [16] [wino]"""
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
w = torch.randn(N, C, 3, 3, device="cuda") * 0.03
wp = ops.pack_conv_weight(w)
wu = ops.pack_conv_weight_wino(w) if "wino" in sys.argv else None      # "wino": the Winograd kernel (the one the sampler runs)
b = torch.zeros(N, device="cuda")
for _ in range(20):
    y = ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
