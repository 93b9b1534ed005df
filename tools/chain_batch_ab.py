#!/usr/bin/env python3
"""A/B of the level chain (DDK_OPT_LEVEL_CHAIN) and the one-launch first Block (DDK_OPT_FIRST_GROUPNORM) on the cfg4 reverse step at several
batch sizes: the chain walks images in rounds of 32, so beyond some batch the launches it replaces are the cheaper form.
    python tools/chain_batch_ab.py [batches, default 16,32,48,64,96,128,192]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from bench import cfg4
from ddk import ops
from models import DownsampleDDPM, Unet
from utils import synthetic as syn

batches = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 32, 48, 64, 96, 128, 192]
dev = torch.device("cuda", 0)
cfg = cfg4()
model = DownsampleDDPM(cfg, Unet(cfg), "cuda", 3)
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
model = model.to(dev).eval()
plan = model.latent_model.plan()
tables = model._tables()
with torch.no_grad():
    for B in batches:
        x = ops.randn((B, 32, 32, 8), dev, seed=1, step=1000, stream_id=0)
        res = {}
        for name, opts in (("chain on", {7: 1, 8: 1}), ("chain off", {7: 0, 8: 1}), ("first-gn off", {7: 1, 8: 0})):
            for k, v in opts.items():
                plan.set_option(k, v)
            plan.sample_nhwc(x, tables, 999, 960, seed=1, stream_id=0, use_graph=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 96
            plan.sample_nhwc(x, tables, 999, 1000 - n, seed=1, stream_id=0, use_graph=True)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e3
        print(f"B={B:4d}: " + "  ".join(f"{k} {v:.4f} ms" for k, v in res.items()), flush=True)
        plan.set_option(7, 1); plan.set_option(8, 1)
