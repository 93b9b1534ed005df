#!/usr/bin/env python3
"""Experiment: does running the reverse chain as TWO concurrent half-batch chains on two HIP streams of one process
(two sampler graphs in flight, kernels of one chain filling the launch gaps / small-grid phases of the other) beat one
full-batch chain?  cfg4 (dDDPM-x3 latents 8x32x32).  GPU-box tool.

usage: dual_chain.py [B_total=32] [steps=100]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import lib as L, ops
from ddk.plan import UnetPlan
from models import DownsampleDDPM, Unet
from utils import synthetic as syn
from tools.sample_bench import cfg

DEV = "cuda"
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def chain(plan, tables, x, ws, nbytes, t_start, t_end, stream, stream_id):
    b, h, w, c = x.shape
    a = L.SamplerArgs(plan.handle, L.ptr(plan.packed), L.ptr(x), None, L.ptr(tables["c_recip"]), L.ptr(tables["c_recipm1"]),
                      L.ptr(tables["c1"]), L.ptr(tables["c2"]), L.ptr(tables["sigma"]), b, h, w, t_start, t_end, 1, stream_id, 1,
                      L.ptr(ws), nbytes)
    L.check(plan._lib.ddk_sampler_run(C.byref(a), stream.cuda_stream), "sampler_run")


def main():
    c = cfg(8, 256, down=3)
    model = DownsampleDDPM(c, Unet(c), DEV, 3).to(DEV).eval()
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    unet = model.latent_model
    tables = model._tables()
    T = model.timesteps
    sd = {k: v for k, v in unet.state_dict().items()}
    # (chains, batch per chain, in-launch GroupNorm): concurrent chains cannot host a whole in-launch-GroupNorm cluster each, so
    # they run with the option off; the single chains are timed both ways so that the comparison is like for like
    cases = [(1, BT, 1), (1, BT, 0), (2, BT // 2, 0), (4, BT // 4, 0)]
    if BT >= 64:
        cases += [(1, BT // 2, 1), (1, BT // 2, 0)]
    for nchains, b, cluster in cases:
        plans, xs, wss, streams = [], [], [], []
        for i in range(nchains):
            p = UnetPlan(unet.in_channels, unet.dim, unet.dim_mults)
            p.pack(sd, DEV)
            p.set_option(p.OPT_CLUSTER_GROUPNORM, cluster)
            nbytes = p._lib.ddk_sampler_workspace_bytes(p.handle, b, 32, 32, T - 1)
            plans.append(p)
            xs.append(ops.randn((b, 32, 32, 8), DEV, seed=1, step=T, stream_id=i))
            wss.append((torch.empty(nbytes // 4 + 4, device=DEV), nbytes))
            streams.append(torch.cuda.Stream())
        torch.cuda.synchronize()

        def go(n):
            for i in range(nchains):
                chain(plans[i], tables, xs[i], wss[i][0], wss[i][1], T - 1, T - n, streams[i], i)
        go(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(STEPS)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        print(f"{nchains} chain(s) x batch {b} (in-launch GroupNorm {'on' if cluster else 'off'}): {dt * 1e3:7.3f} ms per reverse step of all "
              f"chains, {nchains * b / (T * dt):6.2f} images/s (latent chain only, no decode); cluster give-ups {ops.cluster_timeouts()}", flush=True)
        del plans, xs, wss


main()
