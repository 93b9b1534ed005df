"""Evaluation-time variational bound on the GPU: the fused ddk_vlb_terms kernel against the oracle,
and DDPM.test_losses (T x {q_sample, HIP UNet, fused VLB kernel}) against the reference's own test_losses_ output (g9).
Reference: models/diffusion/ddpm.py:317-446, models/utils/losses.py:17-109."""
import numpy as np
import pytest
import torch

from helpers import ddpm_cfg, dddpm_cfg, det_load, golden, rel_err
from utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _x():
    x = syn.synthetic_input((2, 3, 16, 16), "g9.x").clamp(-1, 1)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    return x


def test_vlb_terms_kernel_vs_oracle():
    """every branch: t == 0 (NLL incl. the x < -0.999 / x > 0.999 / clamp(1e-12) cases), t > 0 (KL), L_simple sums"""
    from ddk import ops
    from oracle import diffusion_ref as D
    buf = D.schedule_buffers("linear", 1000)
    B = 6
    x = syn.synthetic_input((B, 3, 16, 16), "vlb.x").clamp(-1, 1)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    x[1, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    eps = syn.synthetic_normal((B, 3, 16, 16), "vlb.eps")
    eps_hat = eps + 0.3 * syn.synthetic_normal((B, 3, 16, 16), "vlb.d")
    eps_hat[1] = 40.0 * syn.synthetic_normal((3, 16, 16), "vlb.far")       # far-off prediction: cdf_delta underflows -> clamp
    t = torch.tensor([0, 0, 1, 17, 500, 999])
    x_t = D.q_sample(buf, x, t, eps)
    want = D.vlb_terms(buf, x, x_t, t, eps_hat)
    want_sq = ((eps - eps_hat) ** 2).sum(dim=(1, 2, 3))
    dv = {k: v.to(DEV) for k, v in buf.items()}
    got, sq = ops.vlb_terms(x.to(DEV), x_t.to(DEV), eps_hat.to(DEV), t.to(DEV), dv["sqrt_recip_alphas_cumprod"],
                            dv["sqrt_recipm1_alphas_cumprod"], dv["posterior_mean_coef1"], dv["posterior_mean_coef2"],
                            dv["posterior_log_variance_clipped"], eps=eps.to(DEV))
    assert torch.isfinite(got).all()
    for b in range(B):
        assert abs(float(got[b]) - float(want[b])) <= 2e-5 * max(1.0, abs(float(want[b]))), (b, float(got[b]), float(want[b]))
    assert rel_err(sq.cpu(), want_sq) < 2e-6
    got2, none = ops.vlb_terms(x.to(DEV), x_t.to(DEV), eps_hat.to(DEV), t.to(DEV), dv["sqrt_recip_alphas_cumprod"],
                               dv["sqrt_recipm1_alphas_cumprod"], dv["posterior_mean_coef1"], dv["posterior_mean_coef2"],
                               dv["posterior_log_variance_clipped"])
    assert none is None and torch.equal(got, got2)                           # deterministic, eps optional


@pytest.mark.parametrize("loss_type", ["simple", "hybrid"])
def test_test_losses_vs_reference(loss_type):
    from models import DDPM, Unet
    g = golden("g9_test_losses")
    cfg = ddpm_cfg(32, 3, 16, T=50, loss_type=loss_type)
    m = det_load(DDPM(cfg, Unet(cfg), DEV, 3)).to(DEV).eval()
    draws = iter([syn.synthetic_normal((2, 3, 16, 16), f"g9.eps{k}").to(DEV) for k in range(50)])
    orig = torch.randn_like
    torch.randn_like = lambda z: next(draws)
    try:
        res = m.test_losses(_x().to(DEV))
    finally:
        torch.randn_like = orig
    assert set(res) == {"vlb_t", "prior", "vlb", "L_simple_t", "L_simple"}
    assert res["vlb_t"].shape == (2, 50) and res["L_simple_t"].shape == (50,)
    for k, v in res.items():
        assert rel_err(v.cpu(), g[f"{loss_type}_{k}"]) < 5e-5, k


def test_vlb_terms_module_api_matches_fused_and_torch_paths():
    """DDPM.vlb_terms (reference signature): the no-grad fused path equals the reference-shaped torch expression."""
    from models import DDPM, Unet
    cfg = ddpm_cfg(32, 3, 16, T=50)
    m = det_load(DDPM(cfg, Unet(cfg), DEV, 3)).to(DEV).eval()
    x = _x().to(DEV)
    t = torch.tensor([0, 31], device=DEV)
    x_t = m.q_sample(x, t, syn.synthetic_normal((2, 3, 16, 16), "g9.eps3").to(DEV))
    with torch.no_grad():
        fused = m.vlb_terms(x, x_t, t)
    with torch.enable_grad():
        plain = m.vlb_terms(x, x_t, t).detach()
    assert rel_err(fused.cpu(), plain.cpu()) < 2e-5


def test_dddpm_test_losses_runs_on_latents():
    from models import DownsampleDDPM, Unet
    cfg = dddpm_cfg(32, 32, 2, T=50)
    m = det_load(DownsampleDDPM(cfg, Unet(cfg), DEV, 3)).to(DEV).eval()
    res = m.test_losses(syn.synthetic_input((2, 3, 32, 32), "g9.dd").to(DEV))
    assert res["vlb_t"].shape == (2, 50) and all(torch.isfinite(v).all() for v in res.values())
