#!/usr/bin/env python3
"""Timeline stamps of conv3x3_wino_kernel (tuning build, DDK_WINO_STAMPS=1): prologue / loop / epilogue per workgroup and the share
of the loop the matrix wave spends parked at the stage barriers.   python tools/wino_clock.py [H C N]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
os.environ["DDK_WINO_STAMPS"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ddk import lib, ops  # noqa: E402

B, H, C, N = 32, 32, 128, 128
if len(sys.argv) > 3:
    H, C, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B, H, H, C, device="cuda")
w = torch.randn(N, C, 3, 3, device="cuda") * (C * 9) ** -0.5
wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w)
bias = torch.zeros(N, device="cuda")
for _ in range(200):
    ops.conv(ops.CONV3X3_S1, x, wp, bias, w_wino=wu)
torch.cuda.synchronize()
L = lib.load()
fn = ctypes.CDLL(lib.LIB_PATH).ddk_debug_read_wino_stamps
buf = (ctypes.c_ulonglong * (8 * 1024))()
fn(buf)
for _ in range(10):
    ops.conv(ops.CONV3X3_S1, x, wp, bias, w_wino=wu)
fn(buf)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(1024, 8)
a = a[a[:, 7] > 0]
t00 = a[:, 0].min()
us = lambda v: (v - t00) / 100.0
ent, l0, l1, end = us(a[:, 0]), us(a[:, 1]), us(a[:, 2]), us(a[:, 6])
ns = a[:, 5]
print(f"{H}x{H} {C}->{N}: {len(a)} WGs, {ns[0]:.0f} stages; entry median {np.median(ent):.1f} (max {ent.max():.1f}) us | prologue {np.median(l0 - ent):.1f} | "
      f"loop {np.median(l1 - l0):.1f} (max {np.max(l1 - l0):.1f}) | epilogue {np.median(end - l1):.1f} | last WG done at {end.max():.1f} us")
print(f"   matrix wave 0: cycles per stage {np.median(a[:, 4] / np.maximum(ns - 1, 1)):.0f} (MFMA-bound 2048), of which parked at the barrier "
      f"{np.median(a[:, 3] / ns):.0f}")

if os.environ.get("DDK_WINO_DEBUG") == "17":
    tl = (ctypes.c_ulonglong * (12 * 8 * 2))()
    ctypes.CDLL(lib.LIB_PATH).ddk_debug_read_wino_timeline(tl)
    t = np.frombuffer(tl, dtype=np.uint64).astype(np.float64).reshape(12, 8, 2)
    base = t[:, 0, 1].min()
    print("   timeline of workgroup 0, stages 8..15: cycles after the first release; per wave: arrive at the stage barrier / leave it")
    for w in range(12):
        role = "matrix" if w < 8 else "loader"
        print(f"   wave {w:2d} ({role}): " + "  ".join(f"{t[w, s, 0] - base:6.0f}/{t[w, s, 1] - base:6.0f}" for s in range(8)))
