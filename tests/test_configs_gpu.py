"""Full-size checks for every BASELINE.json configuration on the GPU (through the C ABI).

For each config: (a) a small-batch slice at the config's FULL width / resolution / schedule against the CPU oracle (sized so
the oracle finishes in seconds), and (b) the config's full batch through size-independent properties -- per-sample
results equal the small-batch results (no cross-sample coupling anywhere: GroupNorm / LayerNorm / attention are per
sample, reference models/unet/blocks.py:57-60,79,126-134) and run-to-run bit stability.

Tolerances: rel = max|a-b| / max|b| over the tensor; one UNet forward is held to 5e-5
(BASELINE.json bar: 1e-3), chains to 1e-4 absolute.

cfg2 and cfg4 full-size checks live in test_unet_gpu.py (full_width goldens, batch independence at B=32) and
test_sampler_gpu.py (cfg4-shaped 50-step chain); this file adds cfg1, cfg3, cfg5 and the vlb / hybrid objectives.
"""
import numpy as np
import pytest
import torch

from helpers import ddpm_cfg, dddpm_cfg, det_load, golden, rel_err
from utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 5e-5


def _cpu_state(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


# ---------------------------------------------------------------------------------------------------- cfg1
def test_cfg1_mnist_full_width_chain_vs_oracle():
    """cfg1: MNIST 32x32, C_in = 1, unet_chan 128, T = 200 linear schedule (betas 5e-4 .. 0.1).
    10 reverse steps from t = 199 with injected noise, B = 2, against the oracle's restatement of p_sample_loop
    (reference models/diffusion/ddpm.py:203-249); then the config's batch of 16: slices equal the B = 2 run."""
    from models import DDPM, Unet
    from oracle import diffusion_ref as D
    from oracle import unet_ref as U
    cfg = ddpm_cfg(128, 1, 32, T=200)
    m = det_load(DDPM(cfg, Unet(cfg), DEV, 1)).eval()
    sd = _cpu_state(m)
    m = m.to(DEV)
    steps, shape = 10, (2, 1, 32, 32)
    x_T = syn.synthetic_normal(shape, "cfg1.xT")
    noise = torch.stack([syn.synthetic_normal(shape, f"cfg1.n{k}") for k in range(steps)])
    got = m.p_sample_loop(shape, early_stop=200 - steps, x_T=x_T, noise=noise).cpu()
    buf = D.schedule_buffers("linear", 200)
    want, _ = D.p_sample_loop(buf, lambda a, b: U.unet_forward(sd, cfg, a, b, pre="latent_model."), x_T, list(noise), 200, 200 - steps)
    assert float((got - want).abs().max()) < 1e-4
    assert torch.equal(got.reshape(2, -1).argmax(dim=1), want.reshape(2, -1).argmax(dim=1))
    # full batch of the config (16): samples 6..7 of the batch reproduce the B = 2 chain when given its inputs
    big = (16, 1, 32, 32)
    xb = syn.synthetic_normal(big, "cfg1.big.xT")
    nb = torch.stack([syn.synthetic_normal(big, f"cfg1.big.n{k}") for k in range(steps)])
    xb[6:8] = x_T
    nb[:, 6:8] = noise
    full = m.p_sample_loop(big, early_stop=200 - steps, x_T=xb, noise=nb).cpu()
    assert float((full[6:8] - got).abs().max()) < 2e-5
    assert torch.equal(full, m.p_sample_loop(big, early_stop=200 - steps, x_T=xb, noise=nb).cpu())   # bit-stable


# ---------------------------------------------------------------------------------------------------- cfg3
def test_cfg3_celeba64_latents_and_decoder_vs_oracle():
    """cfg3: CelebA 64x64 dDDPM -downsample 2: full-width UNet on 8x16x16 latents + the x2 ConvResNet decoder to 3x64x64
    (reference models/diffusion/dddpm.py:76-112).  B = 2 vs the oracle, then the config's batch of 64 by properties."""
    from models import DownsampleDDPM, Unet
    from oracle import resampler_ref as R
    from oracle import unet_ref as U
    cfg = dddpm_cfg(128, 64, 2)
    m = det_load(DownsampleDDPM(cfg, Unet(cfg), DEV, 3)).eval()
    sd = _cpu_state(m)
    m = m.to(DEV)
    assert m.sample_shape == [8, 16, 16]
    z = syn.synthetic_normal((2, 8, 16, 16), "cfg3.z")
    t = torch.tensor([17, 903])
    with torch.no_grad():
        eps = m.latent_model(z.to(DEV), t.to(DEV)).cpu()
        x = m.rescaled_upsample(torch.tanh(z).to(DEV)).cpu()
    assert rel_err(eps, U.unet_forward(sd, cfg, z, t, pre="latent_model.")) < TOL
    assert x.shape == (2, 3, 64, 64)
    assert rel_err(x, R.rescaled_upsample(sd, cfg, torch.tanh(z))) < 2e-5
    # the config's batch: 64 latents
    zb = syn.synthetic_normal((64, 8, 16, 16), "cfg3.big.z")
    zb[40:42] = z
    tb = (torch.arange(64) * 15) % 1000
    tb[40:42] = t
    with torch.no_grad():
        eb = m.latent_model(zb.to(DEV), tb.to(DEV))
        xb = m.rescaled_upsample(torch.tanh(zb).to(DEV))
        assert rel_err(eb[40:42].cpu(), eps) < 2e-5
        assert rel_err(xb[40:42].cpu(), x) < 2e-5
        assert torch.equal(eb, m.latent_model(zb.to(DEV), tb.to(DEV)))
        assert torch.equal(xb, m.rescaled_upsample(torch.tanh(zb).to(DEV)))
    # and the sampling entry point at the config's shapes: (x, z) of 64 images, finite, z in the chain's range
    m.use_graph = True
    xs, zs = m.sample(64, early_stop=997)
    assert xs.shape == (64, 3, 64, 64) and zs.shape == (64, 8, 16, 16)
    assert torch.isfinite(xs).all() and float(xs.abs().max()) <= 1.0


@pytest.mark.parametrize("d_chans", [32, 96])
def test_resamplers_with_d_chans_not_a_multiple_of_64(d_chans):
    """`d_chans` is free in the reference (`ConvResNet(dim, ...)`, convblocks.py:133-159: the blocks' inner width is int(dim / 2)); the
    HIP path takes every multiple of 32 -- an inner width of 16 or 48 runs on a padded pitch of 32 / 64 channels whose
    padding stays exactly zero (zero weight rows, zero bias, Mish(0) = 0).  Encoder and decoder of a tiny dDDPM against the oracle."""
    from models import DownsampleDDPMAutoencoder, Unet
    from oracle import resampler_ref as R
    cfg = dddpm_cfg(32, 32, 2)
    cfg["d_chans"] = d_chans
    m = det_load(DownsampleDDPMAutoencoder(cfg, Unet(cfg), DEV, 3)).eval()
    sd = _cpu_state(m)
    m = m.to(DEV)
    x = syn.synthetic_normal((3, 3, 32, 32), f"dch{d_chans}.x").clamp(-1, 1)
    with torch.no_grad():
        z = m.rescaled_downsample(x.to(DEV)).cpu()
        xr = m.rescaled_upsample(z.to(DEV)).cpu()
    zr = R.rescaled_downsample(sd, cfg, x)
    assert z.shape == (3, 8, 8, 8) and rel_err(z, zr) < 2e-5
    assert xr.shape == (3, 3, 32, 32) and rel_err(xr, R.rescaled_upsample(sd, cfg, z)) < 2e-5
    if d_chans % 64:
        # round 5: these widths train as well (padded parameter copies; gradients vs torch autograd in test_generic_width_train_gpu.py)
        obj, extra = m.train()(x.to(DEV))
        obj.backward()
        assert bool(torch.isfinite(obj)) and all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.downsample.parameters())


# ---------------------------------------------------------------------------------------------------- cfg5
def test_cfg5_full_resolution_256_vs_oracle_and_batch8():
    """cfg5: CelebAMask-HQ 256x256 DDPM -downsample 0: the full-resolution UNet (C_in 3, 65536 pixels per sample in the
    linear attention, GroupNorm slabs of 1 M elements -> the streamed statistics path).  B = 1 vs the oracle
    (reference models/unet/unet.py:74-104), then the config's per-GPU batch of 8: batch independence + bit stability."""
    from models import Unet
    from oracle import unet_ref as U
    cfg = dict(unet_chan=128, unet_in=3, unet_dims=(1, 2, 2, 2), unet_dropout=0.0)
    u = Unet(cfg)
    u.load_state_dict(syn.fill_state_dict(u.state_dict(), 5))
    sd = _cpu_state(u)
    u = u.to(DEV).eval()
    x = syn.synthetic_normal((1, 3, 256, 256), "cfg5.x256")
    t = torch.tensor([421])
    with torch.no_grad():
        y = u(x.to(DEV), t.to(DEV)).cpu()
        ref = U.unet_forward(sd, cfg, x, t)
    assert rel_err(y, ref) < TOL
    xb = syn.synthetic_normal((8, 3, 256, 256), "cfg5.big")
    xb[5] = x[0]
    tb = torch.tensor([0, 999, 250, 7, 600, 421, 33, 871])
    with torch.no_grad():
        yb = u(xb.to(DEV), tb.to(DEV))
        assert rel_err(yb[5:6].cpu(), y) < 2e-5
        assert torch.equal(yb, u(xb.to(DEV), tb.to(DEV)))


# ---------------------------------------------------------------------------------------------------- objectives
@pytest.mark.parametrize("loss_type", ["simple", "vlb", "hybrid"])
def test_loss_type_on_device_vs_golden(loss_type):
    """loss_ddpm for every loss_type (reference models/diffusion/ddpm.py:275-288: vlb_weights[t] * loss, hybrid =
    loss + 1e-4 * vlb) computed on the device against g7, which the reference itself produced."""
    from models import DDPM
    g = golden("g7_qsample_loss")
    cfg = ddpm_cfg(32, 3, 16, loss_type=loss_type)
    m = DDPM(cfg, torch.nn.Identity(), DEV, 3).to(DEV)
    eps = syn.synthetic_normal((4, 3, 16, 16), "g7.eps").to(DEV)
    eps_hat = syn.synthetic_normal((4, 3, 16, 16), "g7.eps_hat").to(DEV)
    t = torch.tensor([0, 1, 499, 999], device=DEV)
    with torch.no_grad():
        got = float(m.loss_ddpm(eps, eps_hat, t))
    assert abs(got / float(g[f"loss_{loss_type}"]) - 1) < 2e-6


def test_loss_flat_mean_on_device_vs_golden():
    from models import DDPM
    g = golden("g7_qsample_loss")
    cfg = ddpm_cfg(32, 3, 16)
    cfg["loss_flat"] = "mean"
    m = DDPM(cfg, torch.nn.Identity(), DEV, 3).to(DEV)
    eps = syn.synthetic_normal((4, 3, 16, 16), "g7.eps").to(DEV)
    eps_hat = syn.synthetic_normal((4, 3, 16, 16), "g7.eps_hat").to(DEV)
    t = torch.tensor([0, 1, 499, 999], device=DEV)
    with torch.no_grad():
        assert abs(float(m.loss_ddpm(eps, eps_hat, t)) / float(g["loss_simple_meanflat"]) - 1) < 2e-6


# ---------------------------------------------------------------------------------------------------- sampler graph cache
def test_sampler_graph_cache_reuse_and_reseed():
    """The plan keeps the captured step: a second chain on the same buffers with another seed must give different draws,
    the same seed the same bits, and repacked weights must not replay a stale time-shift table."""
    from models import DDPM, Unet
    cfg = ddpm_cfg(32, 3, 16)
    m = det_load(DDPM(cfg, Unet(cfg), DEV, 3)).to(DEV).eval()
    shape = (2, 3, 16, 16)
    x_T = syn.synthetic_normal(shape, "cache.xT")
    a = m.p_sample_loop(shape, early_stop=990, x_T=x_T, seed=11).cpu()
    b = m.p_sample_loop(shape, early_stop=990, x_T=x_T, seed=12).cpu()
    c = m.p_sample_loop(shape, early_stop=990, x_T=x_T, seed=11).cpu()
    assert torch.equal(a, c) and not torch.equal(a, b)
    m.use_graph = False
    assert torch.equal(a, m.p_sample_loop(shape, early_stop=990, x_T=x_T, seed=11).cpu())   # graph == eager
    m.use_graph = True
    with torch.no_grad():
        for p in m.latent_model.parameters():
            p.mul_(1.02)
    d = m.p_sample_loop(shape, early_stop=990, x_T=x_T, seed=11).cpu()
    m.use_graph = False
    assert torch.equal(d, m.p_sample_loop(shape, early_stop=990, x_T=x_T, seed=11).cpu())
    assert not torch.equal(a, d)


def test_cfg2_cifar_batch64_cin3_properties():
    """cfg2 at its full batch (CIFAR-10 32x32 DDPM, bs 64, C_in 3): size-independent properties of the forward -- every sample's
    output equals that sample run in a batch of 2 (no cross-sample coupling; the B = 2 slice is pinned by g3.full_c3), the run is
    bit-stable, and a 10-step chain from the same x_T / noise is bit-identical for the samples two different batch compositions
    share (reference: models/unet/unet.py:74-104, models/diffusion/ddpm.py:229-249)."""
    from models import DDPM, Unet
    cfg = ddpm_cfg(128, 3, 32)
    m = det_load(DDPM(cfg, Unet(cfg), DEV, 3)).to(DEV).eval()
    u = m.latent_model
    x = syn.synthetic_normal((64, 3, 32, 32), "cfg2.prop.x").to(DEV)
    t = (torch.arange(64, device=DEV) * 15) % 1000
    with torch.no_grad():
        y = u(x, t)
        assert torch.equal(y, u(x, t))
        for lo in (0, 31, 62):
            y2 = u(x[lo:lo + 2].contiguous(), t[lo:lo + 2].contiguous())
            assert rel_err(y[lo:lo + 2].cpu(), y2.cpu()) < 2e-5
        assert torch.isfinite(y).all()
        noise = torch.stack([syn.synthetic_normal((64, 3, 32, 32), f"cfg2.prop.n{k}") for k in range(10)]).to(DEV)
        full = m.p_sample_loop((64, 3, 32, 32), early_stop=990, x_T=x, noise=noise)
        again = m.p_sample_loop((64, 3, 32, 32), early_stop=990, x_T=x, noise=noise)
        assert torch.equal(full, again)
        part = m.p_sample_loop((8, 3, 32, 32), early_stop=990, x_T=x[40:48].contiguous(), noise=noise[:, 40:48].contiguous())
        assert (full[40:48] - part).abs().max() < 1e-4



def test_sampler_default_batch_192_properties():
    """The drop-in sampler's own default batch (reference generate_model_samples.py:16: batch_size = 192) at cfg4: 192 latents of
    8x32x32 through a 6-step chain and the x3 decoder (its 64-channel 256x256 tensors are 3.2 GB each at this batch).  Size-independent
    properties: the chain and the decoded images of the samples two batch compositions share are equal to those of a batch of 2 with the
    same x_T / noise (no cross-sample coupling, reference models/diffusion/dddpm.py:76-90), and the whole run is bit-stable."""
    from models import DownsampleDDPM, Unet
    cfg = dddpm_cfg(128, 256, 3)
    m = det_load(DownsampleDDPM(cfg, Unet(cfg), DEV, 3)).to(DEV).eval()
    B, steps = 192, 6
    shape = (B, 8, 32, 32)
    x_T = syn.synthetic_normal(shape, "b192.x").to(DEV)
    noise = torch.stack([syn.synthetic_normal(shape, f"b192.n{k}") for k in range(steps)]).to(DEV)
    with torch.no_grad():
        z = m.p_sample_loop(shape, early_stop=1000 - steps, x_T=x_T, noise=noise)
        img = m.rescaled_upsample(z)
        assert tuple(img.shape) == (B, 3, 256, 256) and bool(torch.isfinite(img).all())
        z_again = m.p_sample_loop(shape, early_stop=1000 - steps, x_T=x_T, noise=noise)
        assert torch.equal(z, z_again) and torch.equal(img, m.rescaled_upsample(z_again))
        for lo in (0, 95, 190):
            zp = m.p_sample_loop((2, 8, 32, 32), early_stop=1000 - steps, x_T=x_T[lo:lo + 2].contiguous(),
                                 noise=noise[:, lo:lo + 2].contiguous())
            assert (z[lo:lo + 2] - zp).abs().max() < 1e-4
            ip = m.rescaled_upsample(z[lo:lo + 2].contiguous())
            assert rel_err(img[lo:lo + 2].cpu(), ip.cpu()) < 2e-5
        del img
    torch.cuda.empty_cache()
