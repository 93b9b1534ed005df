#!/usr/bin/env python3
"""A few optimiser steps of one training config for rocprofv3 --kernel-trace --stats: python tools/train_profile.py [cfg3|cfg2|cfg5] [graph|merged]
(graph: the step's two micro-batches captured as two passes; merged: as ONE pass over their concatenation -- the trainer's default since round 5)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from train_bench import cfg, DEV
from models import DDPM, DownsampleDDPMAutoencoder, Unet
from trainers.optim import FusedAdam
from utils import synthetic as syn
from ddk import ops

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg3":
    c = cfg(128, 8, 64, down=2); model = DownsampleDDPMAutoencoder(c, Unet(c), DEV, 3); xshape = (64, 3, 64, 64)
elif which == "cfg4":
    c = cfg(128, 8, 256, down=3); model = DownsampleDDPMAutoencoder(c, Unet(c), DEV, 3); xshape = (32, 3, 256, 256)
elif which == "cfg2":
    c = cfg(128, 3, 32); model = DDPM(c, Unet(c), DEV, 3); xshape = (64, 3, 32, 32)
else:
    c = cfg(128, 3, 256); model = DDPM(c, Unet(c), DEV, 3); xshape = (8, 3, 256, 256)
model = model.to(DEV).train()
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
opt = FusedAdam(model, lr=2e-4)
x = torch.rand(xshape, device=DEV) * 2 - 1
if len(sys.argv) > 2 and sys.argv[2] in ("graph", "merged"):      # the captured step the trainers replay (tools/train_breakdown.py reads its trace)
    from trainers.graph_step import GraphedAccumulation
    xs = [x, x] if sys.argv[2] == "graph" else [torch.cat([x, x])]
    g = GraphedAccumulation(model, len(xs)).capture(xs)
    opt.zero_grad()
    for step in range(6):
        out = g.replay(xs)
        opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    print("done", float(out[-1][0]))
    sys.exit(0)
for step in range(4):
    for _ in range(2):
        with ops.deferred_wgrad():          # as the trainers do: the slab reduces of a backward pass in a few launches
            out = model(x)
            obj = out[0] if isinstance(out, tuple) else out
            (obj / 2).backward()
    opt.step(); opt.zero_grad()
    for m in model.modules():
        if hasattr(m, "invalidate_plan"):
            m.invalidate_plan()
torch.cuda.synchronize()
print("done", float(obj))
