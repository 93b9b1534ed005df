#!/usr/bin/env python3
"""Timeline stamps of conv3x3_gn_wlocal_kernel (tuning build): where a launch on the 8x8 maps spends its cycles.
    make -C downsampled-diffusion_amd/csrc tune && python tools/wl_clock.py [C_in]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ddk import lib, ops  # noqa: E402

B, H, C, N = 32, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 256
x = torch.randn(B, H, H, C, device="cuda")
w = torch.randn(N, C, 3, 3, device="cuda") * (C * 9) ** -0.5
wl = ops.pack_conv_weight_wino_local(w)
b = torch.zeros(N, device="cuda")
gam, bet, temb = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), torch.randn(B, N, device="cuda")
for _ in range(200):
    ops.conv3x3_gn_mish_wino(x, wl, b, gam, bet, temb=temb)
torch.cuda.synchronize()
lib.load()
fn = ctypes.CDLL(lib.LIB_PATH).ddk_debug_read_wl_stamps
buf = (ctypes.c_ulonglong * (8 * 512))()
fn(buf)
for _ in range(5):
    ops.conv3x3_gn_mish_wino(x, wl, b, gam, bet, temb=temb)
fn(buf)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(512, 8)
a = a[a[:, 7] > 0]
nch = a[0, 6]
med = lambda v: float(np.median(v))
print(f"8x8 {C}->{N}, B={B}: {len(a)} workgroups, {nch:.0f} chunks.  shader cycles (median over workgroups):")
print(f"  entry -> image staged {med(a[:, 1] - a[:, 0]):7.0f} | k loop {med(a[:, 2] - a[:, 1]):7.0f} ({med(a[:, 2] - a[:, 1]) / nch:.0f} per chunk; MFMA-bound 2048) "
      f"| epilogue (output transform + GroupNorm tail + stores) {med(a[:, 3] - a[:, 2]):7.0f} | total {med(a[:, 3] - a[:, 0]):7.0f}")
print(f"  matrix wave 0 parked at the chunk barriers {med(a[:, 4]) / nch:.0f} cycles per chunk; transform wave busy {med(a[:, 5]) / nch:.0f} cycles per chunk")
