#!/usr/bin/env python3
"""Training-step timing: cfg3 (CelebA 64x64 dDDPM x2, batch 64), cfg2-like plain DDPM 32x32, cfg5 (full-resolution DDPM
256x256, batch 8), eager vs device-graph replay; `cfg4`: the headline configuration as a training step.  An optimiser step = gradient_accumulate_every = 2 micro-batches of a FULL
batch_size each (reference trainers/trainer_ddpm.py:118-128).  GPU-box tool: python tools/train_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from models import DDPM, DownsampleDDPMAutoencoder, Unet
from trainers.optim import FusedAdam
from utils import synthetic as syn

DEV = "cuda"
if os.environ.get("TRAIN_BENCH_NO_DEFER"):          # A/B: one reduce launch per weight gradient instead of the deferred few
    import contextlib
    from ddk import ops as _ops
    _ops.deferred_wgrad = contextlib.nullcontext


if os.environ.get("TRAIN_BENCH_SIDE_BATCH"):        # A/B: weight-gradient launches per fork of the side stream (0: caller's stream)
    from ddk import ops as _ops
    _ops.WGRAD_SIDE_BATCH = int(os.environ["TRAIN_BENCH_SIDE_BATCH"])
    _ops.WGRAD_SIDE_STREAM = _ops.WGRAD_SIDE_BATCH > 0


def cfg(chan, cin, size, down=0):
    c = dict(unet_chan=chan, unet_in=cin, unet_dims=(1, 2, 2, 2), unet_dropout=0.1, image_size=size, T=1000, loss_type="simple",
             beta_schedule="linear", loss_flat="sum", ema_decay=0.995)
    if down:
        c.update(d_mode="convolutional_res", u_mode="convolutional_res", d_dropout=0, d_chans=64, d_n_blocks=3, u_n_blocks=3, unet_in=8,
                 ae_loss=True, t_rec_max=100, force_latent=True, n_downsamples=down)
    return c


def time_train(name, model, xshape, steps=5):
    """eager: Python + torch.autograd sequence the ~900 launches; graph: the 2 x (forward, backward) replayed as one device graph"""
    from trainers.graph_step import GraphedAccumulation
    model = model.to(DEV).train()
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    opt = FusedAdam(model, lr=2e-4)
    xs = [(torch.rand(xshape, device=DEV) * 2 - 1) for _ in range(2)]
    ga = GraphedAccumulation(model, 2)

    def finish():
        opt.step(); opt.zero_grad()
        for m in model.modules():
            if hasattr(m, "invalidate_plan"):
                m.invalidate_plan()

    def eager():
        ga.static_x = xs
        rows = ga._run()
        finish()
        return rows

    def graphed():
        rows = ga.replay(xs)
        finish()
        return rows

    # the form the trainer runs by default (merge_micro_batches): the step's two micro-batches as one pass over their concatenation
    xm = [torch.cat(xs)]
    gm = GraphedAccumulation(model, 1)

    def merged():
        rows = gm.replay(xm)
        finish()
        return rows

    res = {}
    for mode, fn in (("eager", eager), ("graph", graphed), ("merged", merged)):
        if mode == "graph":
            ga.capture(xs)
            opt.zero_grad()
        if mode == "merged":
            gm.capture(xm)
            opt.zero_grad()
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            rows = fn()
        torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / steps
    print(f"{name}: eager {res['eager'] * 1e3:7.1f} ms, device graph {res['graph'] * 1e3:7.1f} ms, one merged pass (graph) "
          f"{res['merged'] * 1e3:7.1f} ms per optimiser step (2 micro-batches of {xshape[0]}), objective {float(rows[0, 0]):.3f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cfg4":
        # the headline sampling configuration as a TRAINING step: CelebA-HQ 256x256 dDDPM -downsample 3 (32x32 latents of 8), -bs 32
        c = cfg(128, 8, 256, down=3)
        time_train("cfg4 dDDPM-x3 256x256 bs32", DownsampleDDPMAutoencoder(c, Unet(c), DEV, 3), (32, 3, 256, 256), steps=3)
        print(f"peak HBM allocated: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
        sys.exit(0)
    c = cfg(128, 8, 64, down=2)
    time_train("cfg3 dDDPM-x2 64x64 bs64", DownsampleDDPMAutoencoder(c, Unet(c), DEV, 3), (64, 3, 64, 64))
    c = cfg(128, 3, 32)
    time_train("cfg2 DDPM 32x32 bs64    ", DDPM(c, Unet(c), DEV, 3), (64, 3, 32, 32))
    # cfg5: full-resolution DDPM, -bs 8 per GPU: 2 micro-batches of 8 (round 3 timed 2 x 4 under this label: half the work)
    c = cfg(128, 3, 256)
    time_train("cfg5 DDPM 256x256 bs8     ", DDPM(c, Unet(c), DEV, 3), (8, 3, 256, 256), steps=3)
    print(f"peak HBM allocated: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
