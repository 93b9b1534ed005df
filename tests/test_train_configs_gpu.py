"""Training steps of BASELINE.json's configurations at their REAL sizes (reference trainers/trainer_ddpm.py:113-158: two
accumulation micro-batches of a full `batch_size` each, obj / 2 backward, clip, Adam).

The CPU oracle cannot run 2 x 64 images of cfg3 or 2 x 8 images of cfg5 in a test's time, so the full batch is tied to it through
what the objective offers: it is a batch MEAN of per-sample terms, so (a) a small slice run alone on the GPU matches torch-CPU
autograd through oracle/, and (b) the full batch's objective and gradients equal the mean over its slices (each run alone on the
GPU) -- batch independence and linearity -- and (c) are bit-identical run to run (fixed-order reductions everywhere)."""
import os

import numpy as np
import pytest
import torch

from helpers import dddpm_cfg, det_load, rel_err
from oracle import diffusion_ref as D
from oracle import resampler_ref as R
from oracle import train_ref as TR
from oracle import unet_ref as U
from utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run_ae(model, x, t, eps, scale):
    """One micro-batch of the dDDPM-AE objective with injected t / eps (dddpm.py:155-177); adds d(obj * scale) into .grad."""
    model.t_sample = lambda n, t=t: t
    orig = torch.randn_like
    torch.randn_like = lambda z, eps=eps: eps
    try:
        obj, extra = model(x)
    finally:
        torch.randn_like = orig
    (obj * scale).backward()
    return float(obj), float(extra["latent"]), float(extra["recon"])


def _zero(model):
    for p in model.parameters():
        p.grad = None


def test_cfg3_optimiser_step_at_its_real_size():
    """cfg3 (CelebA 64x64 dDDPM -downsample 2, unet_chan 128, 2 micro-batches of 64 images): the x2 encoder / decoder backward at
    64x64 x 64 channels x B = 64 (narrow halo weight-gradient kernel, 256-split slabs, Mish-in-epilogue chains), the UNet on 16x16
    latents at B = 64, then clip + Adam on the 22.7 M-parameter bucket."""
    from models import DownsampleDDPMAutoencoder, Unet
    from trainers.optim import FusedAdam
    cfg = dddpm_cfg(128, 64, 2)          # unet_dropout 0, d_dropout 0: no Philox masks in the comparison
    cfg["ema_decay"] = 0.995
    model = det_load(DownsampleDDPMAutoencoder(cfg, Unet(cfg), DEV, 3))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    params = dict(model.named_parameters())
    probe = ["downsample.conv.1.c2.weight", "downsample.conv.4.c3.weight", "upsample.conv.2.c2.weight", "upsample.conv.6.c4.weight",
             "latent_model.downs.0.0.block1.block.0.weight", "latent_model.ups.1.1.block2.block.0.weight",
             "latent_model.mid_attn.fn.fn.to_qkv.weight", "upsample.conv.7.bias"]
    for k in probe:
        assert k in params, k
    B = 64
    xs = [syn.synthetic_input((B, 3, 64, 64), f"cfg3.train.x{mb}") for mb in range(2)]
    # timesteps on both sides of t_rec_max = 100 (the reconstruction term is masked above it, dddpm.py:114-120)
    ts = [((torch.arange(B) * 37 + 11 * mb) % 1000).long() for mb in range(2)]
    for t in ts:
        t[:8] = torch.tensor([0, 5, 40, 99, 100, 101, 500, 999])
    es = [syn.synthetic_normal((B, 8, 16, 16), f"cfg3.train.eps{mb}") for mb in range(2)]
    xd, td, ed = [x.to(DEV) for x in xs], [t.to(DEV) for t in ts], [e.to(DEV) for e in es]

    # ---- the optimiser step's accumulation at full size, twice: bit-stable
    def accumulate():
        _zero(model)
        vals = [_run_ae(model, xd[mb], td[mb], ed[mb], 0.5) for mb in range(2)]
        return vals, {k: params[k].grad.clone() for k in probe}
    vals, g_full = accumulate()
    vals2, g_full2 = accumulate()
    assert vals == vals2
    for k in probe:
        assert torch.equal(g_full[k], g_full2[k]), k
    assert all(np.isfinite(v).all() for v in vals)

    # ---- one slice (samples 0:2 of micro-batch 0: t = 0 and 5, both below t_rec_max) against torch-CPU autograd through oracle/
    buf = D.schedule_buffers("linear", 1000)
    leaves = {k: v.clone().requires_grad_(k in probe) for k, v in sd.items()}
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    obj_ref, extra_ref = R.dddpm_ae_losses(leaves, buf, cfg, xs[0][:2], ts[0][:2], es[0][:2])
    obj_ref.backward()
    _zero(model)
    o, lat, rec = _run_ae(model, xd[0][:2], td[0][:2], ed[0][:2], 1.0)
    assert abs(o / float(obj_ref) - 1) < 1e-4 and abs(lat / float(extra_ref["latent"]) - 1) < 1e-4
    assert abs(rec / float(extra_ref["recon"]) - 1) < 1e-4
    for k in probe:
        assert rel_err(params[k].grad.cpu(), leaves[k].grad) < 1e-3, k          # the bar of the G6 gradient test

    # ---- the full batch == the mean over its 32 slices of 2, each run alone (batch independence of every per-sample term,
    #      linearity of the fixed-order reductions): objective, latent / recon terms and the probe gradients of micro-batch 0
    _zero(model)
    full0 = _run_ae(model, xd[0], td[0], ed[0], 1.0)
    g0 = {k: params[k].grad.clone() for k in probe}
    _zero(model)
    acc = np.zeros(3)
    for s in range(0, B, 2):
        acc += np.array(_run_ae(model, xd[0][s:s + 2], td[0][s:s + 2], ed[0][s:s + 2], 2.0 / B))
    assert np.allclose(acc / (B // 2), full0, rtol=2e-5), (acc / (B // 2), full0)
    for k in probe:
        assert rel_err(params[k].grad.cpu(), g0[k].cpu()) < 2e-4, k

    # ---- clip + Adam at full size: the fused kernels on the flat bucket against the oracle's arithmetic on the same gradients
    opt = FusedAdam(model, lr=2e-4)
    params = dict(model.named_parameters())          # FusedAdam re-homes the parameters into its flat buffer
    opt.zero_grad()
    for mb in range(2):
        _run_ae(model, xd[mb], td[mb], ed[mb], 0.5)
    flat_g = opt.fp.grad.detach().clone()
    total = torch.sqrt((flat_g.double() ** 2).sum())
    before = {k: params[k].detach().clone() for k in probe}
    grads = {k: params[k].grad.detach().clone() for k in probe}
    norm = opt.step()
    assert abs(float(norm[0]) / float(total) - 1) < 1e-5
    coef = min(1.0, 1.0 / (float(total) + 1e-6))
    assert coef < 1.0          # loss_flat = 'sum': the clip is always active
    for k in probe:
        want, _, _ = TR.adam_step(before[k].cpu(), grads[k].cpu() * coef, torch.zeros_like(before[k].cpu()),
                                  torch.zeros_like(before[k].cpu()), 1, 2e-4)
        assert float((params[k].detach().cpu() - want).abs().max()) < 5e-7, k


def test_cfg5_training_micro_batch_of_8_at_full_resolution():
    """cfg5 (CelebA-HQ 256x256 DDPM, unet_chan 128, -bs 8: micro-batches of EIGHT 256x256 images, trainer_ddpm.py:118-128).  The
    B = 1 slice is pinned to the oracle by test_unet_grads_cfg5_full_resolution_vs_oracle; here the micro-batch of 8: every
    sample's output equals its B = 1 run, the gradients equal the sum of the eight B = 1 gradients, and both are bit-stable."""
    from models import Unet
    cfg = dict(unet_chan=128, unet_in=3, unet_dims=(1, 2, 2, 2), unet_dropout=0.0)
    model = Unet(cfg)
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), 55))
    model = model.to(DEV).train()
    params = dict(model.named_parameters())
    probe = ["downs.0.0.block1.block.0.weight", "downs.0.2.fn.fn.to_qkv.weight", "downs.1.0.res_conv.weight",
             "mid_block1.block2.block.1.weight", "ups.2.0.block1.block.0.weight", "final_conv.1.weight", "time_mlp.1.weight"]
    B = 8
    x = syn.synthetic_input((B, 3, 256, 256), "cfg5.train.x8").to(DEV)
    t = torch.tensor([437, 0, 999, 12, 650, 88, 301, 777], device=DEV)
    wgt = syn.synthetic_normal((B, 3, 256, 256), "cfg5.train.w8").to(DEV)

    def run(sl):
        out = model(x[sl], t[sl])
        (out * wgt[sl]).sum().backward()
        return out.detach()

    _zero(model)
    out8 = run(slice(0, B))
    g8 = {k: params[k].grad.clone() for k in probe}
    _zero(model)
    out8b = run(slice(0, B))
    assert torch.equal(out8, out8b)
    for k in probe:
        assert torch.equal(params[k].grad, g8[k]), k
    _zero(model)
    for b in range(B):
        ob = run(slice(b, b + 1))
        assert rel_err(ob.cpu(), out8[b:b + 1].cpu()) < 2e-5, b
    for k in probe:
        assert rel_err(params[k].grad.cpu(), g8[k].cpu()) < 3e-4, k
    assert torch.cuda.max_memory_allocated() < 40 * 2 ** 30
