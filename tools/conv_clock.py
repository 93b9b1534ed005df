#!/usr/bin/env python3
"""In-kernel clock of the conv k-loop (DDK_DEBUG=32 stamps): cycles per k-chunk and the shader clock actually held."""
import os, sys, ctypes
os.environ["DDK_DEBUG"] = os.environ.get("DDK_DEBUG", "32")
tile = sys.argv[1] if len(sys.argv) > 1 else "auto"
if tile != "auto":
    os.environ["DDK_FORCE_TILE"] = tile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
import numpy as np, torch
from ddk import ops, lib
# usage: conv_clock.py [tile,splits,stages|auto] [H C N [s1|s2|1x1|T]]   (batch 32; default 3x3 stride-1)
B, H, W, C, N = 32, 32, 32, 128, 128
if len(sys.argv) > 4:
    H = W = int(sys.argv[2]); C = int(sys.argv[3]); N = int(sys.argv[4])
kind_name = sys.argv[5] if len(sys.argv) > 5 else "s1"
KIND = {"s1": ops.CONV3X3_S1, "s2": ops.CONV3X3_S2, "1x1": ops.CONV1X1, "T": ops.CONVT4X4_S2}[kind_name]
x = torch.randn(B, H, W, C, device="cuda"); b = torch.zeros(N, device="cuda")
if kind_name == "T":
    wp = ops.pack_convT_weight(torch.randn(C, N, 4, 4, device="cuda") * 0.03)
else:
    k = 1 if kind_name == "1x1" else 3
    wp = ops.pack_conv_weight(torch.randn(N, C, k, k, device="cuda") * 0.03)
# sustained load first (DVFS settles), then read the stamps of the last launch
t_end = __import__("time").time() + 1.0
while __import__("time").time() < t_end:
    for _ in range(50): ops.conv(KIND, x, wp, b, n_out=N)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (6 * 4096))()
lib.load().ddk_debug_read_stamps(buf)          # clear
for _ in range(20): ops.conv(KIND, x, wp, b, n_out=N)
lib.load().ddk_debug_read_stamps(buf)
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)
a = raw[:2048 * 8].reshape(2048, 8)
a = a[a[:, 3] > 0]
if len(a) == 0:
    a = np.ones((1, 8))
cyc, ticks, nit = a[:, 0], a[:, 1], a[:, 2]
clk = cyc / ticks * 100e6
print(f"{kind_name} {H}x{W} {C}->{N} tile {tile}: {len(a)} WGs, k-chunks {nit[0]:.0f}, loop cycles median {np.median(cyc):.0f} (max {cyc.max():.0f}) -> {np.median(cyc / nit):.0f} cycles per chunk; "
      f"shader clock median {np.median(clk) / 1e9:.3f} GHz (min {clk.min() / 1e9:.3f}, max {clk.max() / 1e9:.3f})")
t00 = a[:, 4].min()
us = lambda v: (v - t00) / 100.0
ent, l0, l1, end = us(a[:, 4]), us(a[:, 5]), us(a[:, 6]), us(a[:, 7])
print(f"  timeline (us from first WG entry): entry median {np.median(ent):.1f} max {ent.max():.1f} | loop start median {np.median(l0):.1f} max {l0.max():.1f} | "
      f"loop end median {np.median(l1):.1f} max {l1.max():.1f} | stores drained median {np.median(end):.1f} max {end.max():.1f}")
print(f"  per WG: prologue median {np.median(l0 - ent):.1f} us, loop median {np.median(l1 - l0):.1f} (max {np.max(l1 - l0):.1f}) us, epilogue median {np.median(end - l1):.1f} (max {np.max(end - l1):.1f}) us")

seg = raw[2048 * 8:2048 * 8 + 512 * 16].reshape(2048, 4)
seg = seg[seg[:, 3] > 0]
if len(seg):
    per = seg[:, :3] / seg[:, 3:4]
    print(f"  per k-chunk (cycles, median over {len(seg)} waves): wait+barrier {np.median(per[:, 0]):.0f} | DMA issue {np.median(per[:, 1]):.0f} | reads+MFMA {np.median(per[:, 2]):.0f}")
