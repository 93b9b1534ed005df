#!/usr/bin/env python3
"""In-kernel stamps of conv3x3_c32_kernel (tuning build, DDK_DEBUG=32): cycles of wave 0 in the MFMA loops, at the barrier, in the
epilogues.   make -C downsampled-diffusion_amd/csrc tune && python tools/c32_clock.py [H] [B] [variant: plain|mo|dg]"""
import os, sys, ctypes, time
os.environ["DDK_DEBUG"] = os.environ.get("DDK_DEBUG", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
import numpy as np, torch
from ddk import ops, lib
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
var = sys.argv[3] if len(sys.argv) > 3 else "plain"
x = torch.randn(B, H, H, 32, device="cuda"); w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05; b = torch.randn(32, device="cuda")
wp, wd = ops.pack_conv_weight(w), ops.pack_conv_weight_dgrad(w, i_pad=32)
ao, hs = torch.empty_like(x), torch.randn_like(x)
fn = {"plain": lambda: ops.conv(ops.CONV3X3_S1, x, wp, b), "mo": lambda: ops.conv(ops.CONV3X3_S1, x, wp, b, mish_out=ao),
      "dg": lambda: ops.conv(ops.CONV3X3_S1, x, wd, None, n_out=32, dmish_src=hs)}[var]
t_end = time.time() + 1.0
while time.time() < t_end:
    for _ in range(50): fn()
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (6 * 4096))()
lib.load().ddk_debug_read_stamps(buf)
for _ in range(20): fn()
lib.load().ddk_debug_read_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)[:2048 * 8].reshape(2048, 8)
a = a[a[:, 3] > 0]
med = lambda v: float(np.median(v))
nt = a[:, 2]
dur = a[:, 6] - a[:, 1]
print(f"c32 {var} {H}x{H} B={B}: {len(a)} workgroups, {med(nt):.0f} tiles each; per tile (cycles, median over workgroups): MFMA loop {med(a[:, 0] / nt):.0f} "
      f"(= {med(a[:, 0] / nt) / 144:.1f} per MFMA) | loop end -> behind barrier {med(a[:, 4] / nt):.0f} | epilogue {med(a[:, 5] / nt):.0f} | "
      f"whole workgroup {med(a[:, 7]):.0f} cycles = {med(dur) / 100:.1f} us -> clock {med(a[:, 7] / dur) * 0.1:.2f} GHz")
t0 = a[:, 1].min()
ent, ext = (a[:, 1] - t0) / 100, (a[:, 6] - t0) / 100
print(f"   timeline (us from the first workgroup's entry): entries median {med(ent):.1f}, 90 % {np.percentile(ent, 90):.1f}, last {ent.max():.1f} | "
      f"exits first {ext.min():.1f}, median {med(ext):.1f}, last {ext.max():.1f}")
h, edges = np.histogram(ent, bins=[0, 0.5, 1, 2, 3, 4, 6, 8, 10, 12, 15, 20, 30, 50])
print("   entry histogram (us):", " ".join(f"<{e:g}:{c}" for e, c in zip(edges[1:], h)))
