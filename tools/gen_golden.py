#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU) in the build container.

Build-container tooling only: needs /root/reference, never runs on the GPU box.  The reference
ships no tests or fixtures, so these vectors are what pins oracle/ (and through it
the HIP path) to the reference's behaviour.  Weights and inputs come from the closed-form
fill in downsampled-diffusion_amd/utils/synthetic.py, so only OUTPUTS are stored.

Loader recipe: SURVEY.md Appendix C (a synthetic top-level ``utils`` package avoids the
torchvision / tensorflow / wandb imports that the hot-path modules do not need).

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DDK_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


syn = _load(os.path.join(ROOT, "downsampled-diffusion_amd", "utils", "synthetic.py"), "ddk_synthetic")

uu = _load(f"{REF}/utils/utils.py", "utils.utils")
pkg = types.ModuleType("utils")
pkg.__path__ = []
for n in ("modify_config", "min_max_norm_image", "min_max_norm_batch", "reduce_mean", "reduce_sum",
          "flat_bits", "get_model_state_dict"):
    setattr(pkg, n, getattr(uu, n))
sys.modules["utils"] = pkg
sys.modules["utils.utils"] = uu
sys.path.insert(0, REF)
import models  # noqa: E402  (the reference package)
from models import DDPM, DownsampleDDPM, DownsampleDDPMAutoencoder, Unet  # noqa: E402
from models.unet import blocks as rb  # noqa: E402
import models.diffusion.ddpm as ref_ddpm_mod  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def det_load(module, prefix=""):
    """Fill a reference module with the deterministic weights (keys prefixed like the full model's)."""
    sd = module.state_dict()
    new = {k: (v.clone() if k in syn.SCHEDULE_KEYS else syn.fill_tensor(prefix + k, v.shape))
           for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module


def unet_cfg(chan, cin):
    return dict(unet_chan=chan, unet_in=cin, unet_dims=(1, 2, 2, 2), unet_dropout=0.0)


def ddpm_cfg(chan, cin, size, T=1000, schedule="linear", loss_type="simple"):
    c = unet_cfg(chan, cin)
    c.update(image_size=size, T=T, loss_type=loss_type, beta_schedule=schedule, loss_flat="sum")
    return c


def dddpm_cfg(chan, size, n_down, T=1000):
    c = ddpm_cfg(chan, 8, size, T)
    c.update(d_mode="convolutional_res", u_mode="convolutional_res", d_dropout=0, d_chans=64,
             d_n_blocks=3, u_n_blocks=3, unet_in=8, ae_loss=True, t_rec_max=100, force_latent=True,
             n_downsamples=n_down)
    return c


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ---------------------------------------------------------------- G0 state_dict layout
def g0_keys():
    """Key -> shape of the reference modules' state_dict (the checkpoint compatibility contract)."""
    import json
    out = {}
    cfg = ddpm_cfg(128, 3, 32)
    out["ddpm_c3"] = {k: list(v.shape) for k, v in DDPM(cfg, Unet(cfg), "cpu", 3).state_dict().items()}
    cfgd = dddpm_cfg(128, 256, 3)
    out["dddpm_x3"] = {k: list(v.shape) for k, v in DownsampleDDPM(cfgd, Unet(cfgd), "cpu", 3).state_dict().items()}
    cfgt = dddpm_cfg(32, 32, 2)
    out["dddpm_tiny_x2"] = {k: list(v.shape) for k, v in
                            DownsampleDDPMAutoencoder(cfgt, Unet(cfgt), "cpu", 3).state_dict().items()}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "g0_state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("g0_state_dict_keys:", {k: len(v) for k, v in out.items()})


# ---------------------------------------------------------------- G1 schedule
def g1_schedule():
    out = {}
    for kind, T in (("linear", 1000), ("linear", 200), ("cosine", 1000)):
        m = DDPM(ddpm_cfg(16, 3, 16, T, kind), torch.nn.Identity(), "cpu", 3)
        for k in syn.SCHEDULE_KEYS:
            out[f"{kind}_{T}_{k}"] = m.state_dict()[k].numpy()
        out[f"{kind}_{T}_vlb_weights"] = m.vlb_weights.numpy()
    save("g1_schedule", **out)


# ---------------------------------------------------------------- G2 blocks
@torch.no_grad()
def g2_blocks():
    out = {}
    x32 = syn.synthetic_input((2, 32, 8, 8), "g2.x32")
    x64 = syn.synthetic_input((2, 64, 8, 8), "g2.x64")
    x64s = syn.synthetic_input((2, 64, 4, 4), "g2.x64s")
    temb = syn.synthetic_input((2, 32), "g2.temb")

    out["block_32_64"] = det_load(rb.Block(32, 64), "g2.block.")(x32).numpy()
    out["res_32_64"] = det_load(rb.ResnetBlock(32, 64, time_emb_dim=32), "g2.res_a.")(x32, temb).numpy()
    out["res_64_64"] = det_load(rb.ResnetBlock(64, 64, time_emb_dim=32), "g2.res_b.")(x64s, temb).numpy()
    attn = det_load(rb.Residual(rb.PreNorm(64, rb.LinearAttention(64))), "g2.attn.")
    out["attn_64_8x8"] = attn(x64).numpy()
    out["attn_64_4x4"] = attn(x64s).numpy()
    out["down_64"] = det_load(rb.Downsample(64), "g2.down.")(x64).numpy()
    out["up_64"] = det_load(rb.Upsample(64), "g2.up.")(x64s).numpy()
    ln = det_load(rb.LayerNorm(64), "g2.ln.")
    out["ln_64"] = ln(x64).numpy()
    t = torch.tensor([0, 1, 500, 999])
    out["sinus_32"] = rb.SinusoidalPosEmb(32)(t).numpy()
    out["sinus_128"] = rb.SinusoidalPosEmb(128)(t).numpy()
    out["mish"] = torch.nn.Mish()(torch.linspace(-30, 30, 241)).numpy()
    save("g2_blocks", **out)


# ---------------------------------------------------------------- G3 UNet forward
@torch.no_grad()
def g3_unet():
    out = {}
    for cin in (1, 3, 8):
        u = det_load(Unet(unet_cfg(32, cin)).eval(), "latent_model.")
        x = syn.synthetic_normal((2, cin, 16, 16), f"g3.x{cin}")
        out[f"tiny_c{cin}"] = u(x, torch.tensor([3, 977])).numpy()
    for cin in (3, 8):
        u = det_load(Unet(unet_cfg(128, cin)).eval(), "latent_model.")
        x = syn.synthetic_normal((2, cin, 32, 32), f"g3.full{cin}")
        out[f"full_c{cin}"] = u(x, torch.tensor([999, 17])).numpy()
    save("g3_unet", **out)


# ---------------------------------------------------------------- G4/G5 p_sample trajectory
@torch.no_grad()
def run_chain(model, shape, key, steps, snaps=(1, 10, 50)):
    """Drive the reference's own p_sample with injected noise (noise_like is patched)."""
    T = model.timesteps
    x = syn.synthetic_normal(shape, key + ".xT")
    noises = iter([syn.synthetic_normal(shape, f"{key}.n{k}") for k in range(steps)])
    orig = ref_ddpm_mod.noise_like
    ref_ddpm_mod.noise_like = lambda shp, device, repeat=False: next(noises)
    got = {}
    try:
        for k, i in enumerate(reversed(range(T - steps, T))):
            t = torch.full((shape[0],), i, dtype=torch.long)
            x = model.p_sample(x, t)
            if (k + 1) in snaps:
                got[k + 1] = x.clone()
    finally:
        ref_ddpm_mod.noise_like = orig
    return x, got


@torch.no_grad()
def g4_chain():
    out = {}
    # plain DDPM, tiny width
    cfg = ddpm_cfg(32, 3, 16)
    m = det_load(DDPM(cfg, Unet(cfg), "cpu", 3).eval())
    x, got = run_chain(m, (2, 3, 16, 16), "g4.tiny", 50)
    for s, v in got.items():
        out[f"tiny_step{s}"] = v.numpy()
    out["tiny_fixed"] = uu.min_max_norm_image(x).mul(255.).numpy().transpose(0, 2, 3, 1)   # eval_helpers.py:37-41
    out["tiny_argmax"] = x.reshape(2, -1).argmax(dim=1).numpy()
    # a whole T=50 chain (t = 49 .. 0): its last step exercises the t == 0 noise mask
    cfg5 = ddpm_cfg(32, 3, 16, T=50)
    m5 = det_load(DDPM(cfg5, Unet(cfg5), "cpu", 3).eval())
    x5, got5 = run_chain(m5, (2, 3, 16, 16), "g4.t50", 50, snaps=(49, 50))
    out["t50_step49"] = got5[49].numpy()
    out["t50_step50"] = got5[50].numpy()
    # full width, cfg4-shaped latent (8 x 32 x 32), 50 steps
    cfgf = ddpm_cfg(128, 8, 32)
    mf = det_load(DDPM(cfgf, Unet(cfgf), "cpu", 8).eval())
    xf, gotf = run_chain(mf, (2, 8, 32, 32), "g4.full", 50)
    for s, v in gotf.items():
        out[f"full_step{s}"] = v.numpy()
    out["full_argmax"] = xf.reshape(2, -1).argmax(dim=1).numpy()
    # dDDPM: latent chain + tanh(upsample(z)); image 32 -> latent 8 (x2 downsamples)
    cfgd = dddpm_cfg(32, 32, 2)
    md = det_load(DownsampleDDPM(cfgd, Unet(cfgd), "cpu", 3).eval())
    z, gotz = run_chain(md, (2, 8, 8, 8), "g4.dd", 50)
    out["dd_z"] = z.numpy()
    out["dd_x"] = md.rescaled_upsample(z).numpy()
    out["dd_x_fixed"] = uu.min_max_norm_image(md.rescaled_upsample(z)).mul(255.).numpy().transpose(0, 2, 3, 1)
    save("g4_chain", **out)


# ---------------------------------------------------------------- G7 q_sample / losses
@torch.no_grad()
def g7_qsample_loss():
    out = {}
    x = syn.synthetic_input((4, 3, 16, 16), "g7.x")
    eps = syn.synthetic_normal((4, 3, 16, 16), "g7.eps")
    eps_hat = syn.synthetic_normal((4, 3, 16, 16), "g7.eps_hat")
    t = torch.tensor([0, 1, 499, 999])
    for lt in ("simple", "vlb", "hybrid"):
        cfg = ddpm_cfg(16, 3, 16, loss_type=lt)
        m = DDPM(cfg, torch.nn.Identity(), "cpu", 3)
        out[f"loss_{lt}"] = m.loss_ddpm(eps, eps_hat, t).numpy()
    out["q_sample"] = m.q_sample(x, t, eps).numpy()
    out["x0_clip"] = m.predict_x_from_eps(x, t, eps, clip=True).numpy()
    out["x0_noclip"] = m.predict_x_from_eps(x, t, eps, clip=False).numpy()
    mean, var, logvar = m.q_posterior(x, eps, t)
    out["post_mean"] = mean.numpy()
    out["post_logvar"] = logvar.numpy()
    cfgm = ddpm_cfg(16, 3, 16)
    cfgm["loss_flat"] = "mean"
    out["loss_simple_meanflat"] = DDPM(cfgm, torch.nn.Identity(), "cpu", 3).loss_ddpm(eps, eps_hat, t).numpy()
    save("g7_qsample_loss", **out)


# ---------------------------------------------------------------- G8 dDDPM resamplers
@torch.no_grad()
def g8_resamplers():
    out = {}
    for n_down, size in ((2, 32), (3, 32)):
        cfg = dddpm_cfg(32, size, n_down)
        m = det_load(DownsampleDDPMAutoencoder(cfg, Unet(cfg), "cpu", 3).eval())
        x = syn.synthetic_input((2, 3, size, size), f"g8.x{n_down}")
        z = m.rescaled_downsample(x)
        out[f"down{n_down}_z"] = z.numpy()
        out[f"down{n_down}_raw"] = m.downsample(x).numpy()
        out[f"up{n_down}_x"] = m.rescaled_upsample(z).numpy()
    save("g8_resamplers", **out)


# ---------------------------------------------------------------- G6 training step
def g6_train():
    """trainer_ddpm.py:113-158 restated around the reference MODEL classes (dropout 0):
    2 micro-batches, obj/2 backward, clip_grad_norm_ 1.0, Adam(lr), EMA reset / lerp."""
    from copy import deepcopy
    out = {}
    for tag in ("ddpm", "dddpm_ae", "dddpm"):
        if tag == "ddpm":
            cfg = ddpm_cfg(32, 3, 16)
            model = det_load(DDPM(cfg, Unet(cfg), "cpu", 3))
            xshape, eshape = (4, 3, 16, 16), (4, 3, 16, 16)
        else:
            cfg = dddpm_cfg(32, 32, 2)
            cls = DownsampleDDPMAutoencoder if tag == "dddpm_ae" else DownsampleDDPM
            model = det_load(cls(cfg, Unet(cfg), "cpu", 3))
            xshape, eshape = (4, 3, 32, 32), (4, 8, 8, 8)
        model.train()
        lr = 2e-4
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        names = [k for k, _ in model.named_parameters()]
        probe = [names[0], names[len(names) // 3], names[len(names) // 2], names[-3], names[-1]]
        out[f"{tag}_probe_names"] = np.array(probe)
        ema = None
        for step in range(2):
            objs = []
            for mb in range(2):
                x = syn.synthetic_input(xshape, f"g6.{tag}.x{step}{mb}")
                tt = torch.tensor([0, 40 + step, 500, 999 - mb])      # covers t < t_rec_max and t >= t_rec_max
                eps = syn.synthetic_normal(eshape, f"g6.{tag}.eps{step}{mb}")
                model.t_sample = lambda n, tt=tt: tt
                orig = torch.randn_like
                torch.randn_like = lambda z, eps=eps: eps
                try:
                    res = model(x)
                finally:
                    torch.randn_like = orig
                obj = res[0] if isinstance(res, tuple) else res
                if isinstance(res, tuple) and step == 0:
                    out[f"{tag}_latent{mb}"] = res[1]["latent"].detach().numpy()
                    out[f"{tag}_recon{mb}"] = res[1]["recon"].detach().numpy()
                (obj / 2).backward()
                objs.append(obj.item())
            out[f"{tag}_obj{step}"] = np.array(objs, dtype=np.float64)
            if step == 0:
                for n in probe:
                    g = dict(model.named_parameters())[n].grad
                    out[f"{tag}_grad_{n}"] = g.numpy().copy()
            total = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            out[f"{tag}_gradnorm{step}"] = total.numpy()
            opt.step()
            opt.zero_grad()
            for n in probe:
                out[f"{tag}_param{step}_{n}"] = dict(model.named_parameters())[n].detach().numpy().copy()
            # EMA: reset (deepcopy) at step 0, then one lerp with decay 0.995 (ema.py:33-44)
            if step == 0:
                ema = deepcopy(model)
            else:
                for pe, pn in zip(ema.parameters(), model.parameters()):
                    pe.data = pe.data * 0.995 + (1 - 0.995) * pn.data
                for n in probe:
                    out[f"{tag}_ema_{n}"] = dict(ema.named_parameters())[n].detach().numpy().copy()
    save("g6_train", **out)


# ---------------------------------------------------------------- G9 evaluation-time VLB (test_losses_)
@torch.no_grad()
def g9_test_losses():
    """The reference's own test_losses_ (models/diffusion/ddpm.py:393-442: T x {q_sample, vlb_terms, L_simple}) on the tiny
    model with T = 50 (T = 20 makes the linear schedule end at beta = 1: NaN at t = T-1) and the per-step noise injected (torch.randn_like patched, one draw per timestep, t = T-1 .. 0);
    plus dDDPM.test_losses (dddpm.py: test_losses_ on the tanh-rescaled latents).  x has values beyond +-0.999 so every
    branch of discretized_gaussian_log_likelihood (models/utils/losses.py:79-109) is exercised at t = 0."""
    out = {}
    for lt in ("simple", "hybrid"):
        cfg = ddpm_cfg(32, 3, 16, T=50, loss_type=lt)
        m = det_load(DDPM(cfg, Unet(cfg), "cpu", 3)).eval()
        x = syn.synthetic_input((2, 3, 16, 16), "g9.x").clamp(-1, 1)
        x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
        draws = iter([syn.synthetic_normal((2, 3, 16, 16), f"g9.eps{k}") for k in range(50)])
        orig = torch.randn_like
        torch.randn_like = lambda z: next(draws)
        try:
            res = m.test_losses(x)
        finally:
            torch.randn_like = orig
        for k, v in res.items():
            out[f"{lt}_{k}"] = v.numpy()
    # the pieces, on fixed inputs (for the kernel-level test): kl / nll per element before flat_bits
    from models.utils import discretized_gaussian_log_likelihood, normal_kl
    x = syn.synthetic_input((2, 3, 16, 16), "g9.x").clamp(-1, 1)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    mean1 = syn.synthetic_normal((2, 3, 16, 16), "g9.mean1") * 0.5
    mean2 = syn.synthetic_normal((2, 3, 16, 16), "g9.mean2") * 0.5
    lv1 = torch.tensor([-3.0, -0.5]).view(2, 1, 1, 1)
    lv2 = torch.tensor([-2.5, -0.75]).view(2, 1, 1, 1)
    out["piece_kl"] = normal_kl(mean1, lv1, mean2, lv2).numpy()
    out["piece_ll"] = discretized_gaussian_log_likelihood(x, means=mean2, log_scales=0.5 * lv2).numpy()
    out["piece_prior"] = uu.flat_bits(normal_kl(mean1, lv1, 0., 0.)).numpy()
    save("g9_test_losses", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g6", "g7", "g8", "g9"]
    table = dict(g0=g0_keys, g1=g1_schedule, g2=g2_blocks, g3=g3_unet, g4=g4_chain, g6=g6_train, g7=g7_qsample_loss,
                 g8=g8_resamplers, g9=g9_test_losses)
    for w in which:
        table[w]()
