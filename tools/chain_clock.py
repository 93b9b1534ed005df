#!/usr/bin/env python3
"""Where the level chain (csrc/level_chain.hip) spends its time, op by op: in-kernel s_memrealtime stamps of three workgroups
(blocks 0, 100, 255) from the TUNING build.   make -C downsampled-diffusion_amd/csrc tune && DDK_LIB=.../libddk_tune.so python tools/chain_clock.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk_tune.so"))
import torch
from bench import cfg4
from ddk import lib as L, ops
from models import DownsampleDDPM, Unet
from utils import synthetic as syn

dev = torch.device("cuda", 0)
cfg = cfg4()
model = DownsampleDDPM(cfg, Unet(cfg), "cuda", 3)
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
model = model.to(dev).eval()
plan = model.latent_model.plan()
tables = model._tables()
x = ops.randn((32, 32, 32, 8), dev, seed=1, step=1000, stream_id=0)
with torch.no_grad():
    plan.sample_nhwc(x, tables, 999, 900, seed=1, stream_id=0, use_graph=True)
torch.cuda.synchronize()
lib = L.load()
NOPS = 24
buf = (C.c_ulonglong * (9 * NOPS * 8))()
lib.ddk_debug_read_lc_stamps.argtypes = [C.c_void_p]
assert lib.ddk_debug_read_lc_stamps(buf) == 0
names = ["down", "d0.c1", "d0.c2", "d1.c1", "d1.c2", "d.attn", "d.out", "m1.c1", "m1.c2", "m.attn", "m.out", "m2.c1", "m2.c2", "u0.res", "u0.c1",
         "u0.c2", "u1.c1", "u1.c2", "u.attn", "u.out", "upT"]
names8d = ["d0.c1", "d0.c2", "d1.c1", "d1.c2", "attn", "out"]
names8u = ["u0.res", "u0.c1", "u0.c2", "u1.c1", "u1.c2", "attn", "out"]
which = sys.argv[1] if len(sys.argv) > 1 else "4"
base, names = {"4": (0, names), "8d": (3, names8d), "8u": (6, names8u)}[which]
for w0, blk in enumerate((0, 100, 255)):
    w = base + w0
    print(f"chain {which} workgroup {blk}: us per phase   wait   stage   loop   tail+signal | op total | since kernel entry")
    t00 = buf[(w * NOPS) * 8]
    tot = [0.0] * 4
    for k, nm in enumerate(names):
        s = [buf[(w * NOPS + k) * 8 + i] for i in range(5)]
        if s[4] == 0:
            continue
        d = [(s[i + 1] - s[i]) / 100.0 for i in range(4)]
        for i in range(4):
            tot[i] += d[i]
        print(f"  {k:2d} {nm:7s}              {d[0]:6.2f} {d[1]:6.2f} {d[2]:6.2f} {d[3]:6.2f}        | {(s[4] - s[0]) / 100.0:6.2f}  | {(s[4] - t00) / 100.0:7.2f}")
    print(f"  sums                    {tot[0]:6.2f} {tot[1]:6.2f} {tot[2]:6.2f} {tot[3]:6.2f}")
