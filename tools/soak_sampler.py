#!/usr/bin/env python3
"""Soak: full T = 1000 reverse chains at cfg4 (batch 32) back to back; checks the outputs stay finite, two chains from the same seed are
bit-identical, and the cluster-GroupNorm exchange never gave up (ddk_debug_cluster_timeouts() == 0).   python tools/soak_sampler.py [chains]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402

from ddk import ops  # noqa: E402
from models import DownsampleDDPM, Unet  # noqa: E402
from sample_bench import DEV, cfg  # noqa: E402
from utils import synthetic as syn  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c = cfg(8, 256, down=3)
model = DownsampleDDPM(c, Unet(c), DEV, 3).to(DEV).eval()
model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
C, S, _ = model.sample_shape
T = model.timesteps
plan, tables = model.latent_model.plan(), model._tables()
outs = []
t0 = time.perf_counter()
with torch.no_grad():
    for k in range(chains):
        x = ops.randn((32, S, S, C), DEV, seed=7, step=T, stream_id=0)
        plan.sample_nhwc(x, tables, T - 1, 0, seed=11 + (k % 2))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(x).all()), f"chain {k}: non-finite latents"
        outs.append(x.clone())
        print(f"chain {k}: |x| max {float(x.abs().max()):.3f}, cluster give-ups {ops.cluster_timeouts()}, {time.perf_counter() - t0:.1f} s", flush=True)
assert ops.cluster_timeouts() == 0
for k in range(2, chains):
    assert torch.equal(outs[k], outs[k - 2]), f"chain {k} differs from chain {k - 2} (same seed)"
print("soak ok")
