#!/usr/bin/env python3
"""A/B of a plan option on the cfg4 reverse step (batch 32, graph replay): python tools/fold_ab.py [option id] -- ms per step with the
option off / on, alternating, 3 rounds of 200 steps each.  Default option: DDK_OPT_FOLD_DOWNSAMPLE_REDUCE (5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from bench import cfg4
from ddk import ops
from models import DownsampleDDPM, Unet
from utils import synthetic as syn

opt = int(sys.argv[1]) if len(sys.argv) > 1 else 5
vals = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]       # option values to alternate between
dev = torch.device("cuda", 0)
cfg = cfg4()
model = DownsampleDDPM(cfg, Unet(cfg), "cuda", 3)
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
model = model.to(dev).eval()
plan = model.latent_model.plan()
tables = model._tables()
x = ops.randn((32, 32, 32, 8), dev, seed=1, step=1000, stream_id=0)


def run(k):
    plan.sample_nhwc(x, tables, 999, 1000 - k, seed=1, stream_id=0, use_graph=True)


with torch.no_grad():
    for rnd in range(3):
        for val in vals:
            plan.set_option(opt, val)
            run(40)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(200)
            torch.cuda.synchronize()
            print(f"round {rnd} option {opt} = {val}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step", flush=True)
