#!/usr/bin/env python3
"""Timeline stamps of conv3x3_gn_local_kernel (tuning build): where a launch on the 4x4 maps spends its cycles.
    make -C downsampled-diffusion_amd/csrc tune && python tools/local_clock.py [C_in]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
import numpy as np, torch
from ddk import lib, ops
B, H, C, N = 32, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 256
x = torch.randn(B, H, H, C, device="cuda")
w = torch.randn(N, C, 3, 3, device="cuda") * (C * 9) ** -0.5
wl = ops.pack_conv_weight_local(w)
b = torch.zeros(N, device="cuda")
gam, bet, temb = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), torch.randn(B, N, device="cuda")
for _ in range(300):
    ops.conv3x3_gn_mish(x, wl, b, gam, bet, temb=temb)
torch.cuda.synchronize()
lib.load()
fn = ctypes.CDLL(lib.LIB_PATH).ddk_debug_read_wl_stamps
buf = (ctypes.c_ulonglong * (8 * 512))()
fn(buf)
for _ in range(5):
    ops.conv3x3_gn_mish(x, wl, b, gam, bet, temb=temb)
fn(buf)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(512, 8)
a = a[a[:, 7] == 2]
med = lambda v: float(np.median(v))
print(f"4x4 {C}->{N}, B={B}: {len(a)} workgroups.  shader cycles (median over workgroups):")
print(f"  entry -> image + first weights staged {med(a[:, 1] - a[:, 0]):7.0f} | k loop {med(a[:, 2] - a[:, 1]):7.0f} | partial tiles through LDS {med(a[:, 4] - a[:, 2]):7.0f} "
      f"| GroupNorm tail + stores {med(a[:, 3] - a[:, 4]):7.0f} | total {med(a[:, 3] - a[:, 0]):7.0f}")
t0 = a[:, 5].min()
ent, ext = (a[:, 5] - t0) / 100, (a[:, 6] - t0) / 100
print(f"  timeline (us from the first entry): entries median {med(ent):.2f}, last {ent.max():.2f} | exits first {ext.min():.2f}, median {med(ext):.2f}, last {ext.max():.2f}; "
      f"clock {med((a[:, 3] - a[:, 0]) / (a[:, 6] - a[:, 5])) * 0.1:.2f} GHz")
