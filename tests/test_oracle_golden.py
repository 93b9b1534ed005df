"""Pins oracle/ (the CPU restatement) to the golden vectors captured from the reference itself
(tools/gen_golden.py).  Tolerances are fp32 round-off: the oracle uses the same ATen ops on the same
inputs; where it restates a formula by hand (Mish, LayerNorm, softmax attention) last-bit differences appear."""
import numpy as np
import pytest
import torch

from helpers import ddpm_cfg, dddpm_cfg, det_state, golden, golden_keys, rel_err, unet_cfg
from oracle import diffusion_ref as D
from oracle import resampler_ref as R
from oracle import train_ref as TR
from oracle import unet_ref as U
from utils import synthetic as syn

torch.set_num_threads(8)


def shapes_for(tag, prefix_filter=None):
    ks = golden_keys()[tag]
    return {k: v for k, v in ks.items() if prefix_filter is None or k.startswith(prefix_filter)}


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize("kind,T", [("linear", 1000), ("linear", 200), ("cosine", 1000)])
def test_schedule_bit_exact(kind, T):
    g = golden("g1_schedule")
    buf = D.schedule_buffers(kind, T)
    for k in D.SCHEDULE_KEYS + ("vlb_weights",):
        assert np.array_equal(buf[k].numpy(), g[f"{kind}_{T}_{k}"]), k


def test_linear_T200_range():
    b = D.beta_schedule("linear", 200)
    assert abs(b[0] - 5e-4) < 1e-12 and abs(b[-1] - 0.1) < 1e-12


# ---------------------------------------------------------------- G2
def _blk_sd(mod_shapes, prefix):
    return {k: syn.fill_tensor(prefix + k, s) for k, s in mod_shapes.items()}


def test_blocks():
    g = golden("g2_blocks")
    x32 = syn.synthetic_input((2, 32, 8, 8), "g2.x32")
    x64 = syn.synthetic_input((2, 64, 8, 8), "g2.x64")
    x64s = syn.synthetic_input((2, 64, 4, 4), "g2.x64s")
    temb = syn.synthetic_input((2, 32), "g2.temb")

    sd = _blk_sd({"block.0.weight": (64, 32, 3, 3), "block.0.bias": (64,), "block.1.weight": (64,), "block.1.bias": (64,)},
                 "g2.block.")
    assert rel_err(U.block(sd, "", x32), g["block_32_64"]) < 2e-6

    def res_shapes(ci, co, td):
        s = {"mlp.1.weight": (co, td), "mlp.1.bias": (co,)}
        for b, c in (("block1", ci), ("block2", co)):
            s.update({f"{b}.block.0.weight": (co, c, 3, 3), f"{b}.block.0.bias": (co,), f"{b}.block.1.weight": (co,),
                      f"{b}.block.1.bias": (co,)})
        if ci != co:
            s.update({"res_conv.weight": (co, ci, 1, 1), "res_conv.bias": (co,)})
        return s
    # the reference's ResnetBlock applies Mish inside its mlp, on the raw time vector
    assert rel_err(U.resnet_block(_blk_sd(res_shapes(32, 64, 32), "g2.res_a."), "", x32, temb), g["res_32_64"]) < 2e-6
    assert rel_err(U.resnet_block(_blk_sd(res_shapes(64, 64, 32), "g2.res_b."), "", x64s, temb), g["res_64_64"]) < 2e-6

    asd = _blk_sd({"fn.fn.to_qkv.weight": (384, 64, 1, 1), "fn.fn.to_out.weight": (64, 128, 1, 1), "fn.fn.to_out.bias": (64,),
                   "fn.norm.g": (1, 64, 1, 1), "fn.norm.b": (1, 64, 1, 1)}, "g2.attn.")
    assert rel_err(U.attention_block(asd, "", x64), g["attn_64_8x8"]) < 2e-6
    assert rel_err(U.attention_block(asd, "", x64s), g["attn_64_4x4"]) < 2e-6

    dsd = _blk_sd({"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, "g2.down.")
    assert rel_err(U.downsample(dsd, "", x64), g["down_64"]) < 2e-6
    usd = _blk_sd({"conv.weight": (64, 64, 4, 4), "conv.bias": (64,)}, "g2.up.")
    assert rel_err(U.upsample(usd, "", x64s), g["up_64"]) < 2e-6
    lsd = _blk_sd({"g": (1, 64, 1, 1), "b": (1, 64, 1, 1)}, "g2.ln.")
    assert rel_err(U.chan_layernorm(x64, lsd["g"], lsd["b"]), g["ln_64"]) < 2e-6

    t = torch.tensor([0, 1, 500, 999])
    assert np.array_equal(U.sinusoidal_embedding(t, 32).numpy(), g["sinus_32"])
    assert np.array_equal(U.sinusoidal_embedding(t, 128).numpy(), g["sinus_128"])
    assert np.abs(U.mish(torch.linspace(-30, 30, 241)).numpy() - g["mish"]).max() < 1e-6


# ---------------------------------------------------------------- G3
@pytest.mark.parametrize("cin", [1, 3, 8])
def test_unet_tiny(cin):
    g = golden("g3_unet")
    shapes = {k[len("latent_model."):]: v for k, v in shapes_for("dddpm_tiny_x2", "latent_model.").items()}
    shapes["downs.0.0.block1.block.0.weight"] = [32, cin, 3, 3]
    shapes["downs.0.0.res_conv.weight"] = [32, cin, 1, 1]
    shapes["final_conv.1.weight"] = [cin, 32, 1, 1]
    shapes["final_conv.1.bias"] = [cin]
    sd = det_state(shapes, "latent_model.")
    x = syn.synthetic_normal((2, cin, 16, 16), f"g3.x{cin}")
    y = U.unet_forward(sd, unet_cfg(32, cin), x, torch.tensor([3, 977]))
    assert rel_err(y, g[f"tiny_c{cin}"]) < 5e-6


def test_unet_full_width():
    g = golden("g3_unet")
    shapes = {k[len("latent_model."):]: v for k, v in shapes_for("dddpm_x3", "latent_model.").items()}
    sd = det_state(shapes, "latent_model.")
    x = syn.synthetic_normal((2, 8, 32, 32), "g3.full8")
    y = U.unet_forward(sd, unet_cfg(128, 8), x, torch.tensor([999, 17]))
    assert rel_err(y, g["full_c8"]) < 5e-6


# ---------------------------------------------------------------- G4 / G5
def _chain(sd, cfg, shape, key, steps, T, pre="latent_model."):
    buf = D.schedule_buffers(cfg["beta_schedule"], T)
    x_T = syn.synthetic_normal(shape, key + ".xT")
    noises = [syn.synthetic_normal(shape, f"{key}.n{k}") for k in range(steps)]
    eps_model = lambda x, t: U.unet_forward(sd, cfg, x, t, pre=pre)
    return D.p_sample_loop(buf, eps_model, x_T, noises, T, t_end=T - steps)


def test_chain_tiny_50_steps():
    g = golden("g4_chain")
    cfg = ddpm_cfg(32, 3, 16)
    shapes = {k: v for k, v in golden_keys()["dddpm_tiny_x2"].items() if k.startswith("latent_model.")}
    shapes["latent_model.downs.0.0.block1.block.0.weight"] = [32, 3, 3, 3]
    shapes["latent_model.downs.0.0.res_conv.weight"] = [32, 3, 1, 1]
    shapes["latent_model.final_conv.1.weight"] = [3, 32, 1, 1]
    shapes["latent_model.final_conv.1.bias"] = [3]
    sd = det_state(shapes)
    x, snaps = _chain(sd, cfg, (2, 3, 16, 16), "g4.tiny", 50, 1000)
    for s in (1, 10, 50):
        assert np.abs(snaps[s].numpy() - g[f"tiny_step{s}"]).max() < 2e-5, s
    assert np.array_equal(x.reshape(2, -1).argmax(dim=1).numpy(), g["tiny_argmax"])
    fixed = D.fix_samples(x)
    assert np.abs(fixed - g["tiny_fixed"]).max() < 5e-3      # [0,255] scale
    assert np.array_equal(np.round(fixed).astype(np.uint8), np.round(g["tiny_fixed"]).astype(np.uint8)) or \
        (np.round(fixed) != np.round(g["tiny_fixed"])).mean() < 1e-3
    # whole T=50 chain incl. the t == 0 step (noise masked)
    cfg50 = ddpm_cfg(32, 3, 16, T=50)
    x50, s50 = _chain(sd, cfg50, (2, 3, 16, 16), "g4.t50", 50, 50)
    assert np.abs(s50[49].numpy() - g["t50_step49"]).max() < 2e-5
    assert np.abs(s50[50].numpy() - g["t50_step50"]).max() < 2e-5


# ---------------------------------------------------------------- G7
def test_qsample_and_losses():
    g = golden("g7_qsample_loss")
    buf = D.schedule_buffers("linear", 1000)
    x = syn.synthetic_input((4, 3, 16, 16), "g7.x")
    eps = syn.synthetic_normal((4, 3, 16, 16), "g7.eps")
    eps_hat = syn.synthetic_normal((4, 3, 16, 16), "g7.eps_hat")
    t = torch.tensor([0, 1, 499, 999])
    assert np.array_equal(D.q_sample(buf, x, t, eps).numpy(), g["q_sample"])
    assert np.array_equal(D.predict_x_from_eps(buf, x, t, eps, True).numpy(), g["x0_clip"])
    assert np.array_equal(D.predict_x_from_eps(buf, x, t, eps, False).numpy(), g["x0_noclip"])
    for lt in ("simple", "vlb", "hybrid"):
        assert abs(float(D.loss_ddpm(buf, eps, eps_hat, t, lt)) / float(g[f"loss_{lt}"]) - 1) < 1e-6, lt
    assert abs(float(D.loss_ddpm(buf, eps, eps_hat, t, "simple", "mean")) / float(g["loss_simple_meanflat"]) - 1) < 1e-6


# ---------------------------------------------------------------- G8
@pytest.mark.parametrize("n_down", [2, 3])
def test_resamplers(n_down):
    g = golden("g8_resamplers")
    cfg = dddpm_cfg(32, 32, n_down)
    keys = golden_keys()["dddpm_tiny_x2" if n_down == 2 else "dddpm_x3"]
    shapes = {k: v for k, v in keys.items() if k.startswith(("downsample.", "upsample."))}
    sd = det_state(shapes)
    x = syn.synthetic_input((2, 3, 32, 32), f"g8.x{n_down}")
    z = R.rescaled_downsample(sd, cfg, x)
    assert rel_err(z, g[f"down{n_down}_z"]) < 5e-6
    assert rel_err(R.rescaled_upsample(sd, cfg, torch.from_numpy(g[f"down{n_down}_z"])), g[f"up{n_down}_x"]) < 5e-6


def test_dddpm_chain_and_decode():
    g = golden("g4_chain")
    cfg = dddpm_cfg(32, 32, 2)
    sd = det_state(golden_keys()["dddpm_tiny_x2"])
    z, _ = _chain(sd, cfg, (2, 8, 8, 8), "g4.dd", 50, 1000)
    assert np.abs(z.numpy() - g["dd_z"]).max() < 2e-5
    xx = R.rescaled_upsample(sd, cfg, z)
    assert np.abs(xx.numpy() - g["dd_x"]).max() < 2e-5


# ---------------------------------------------------------------- G6 training step
@pytest.mark.parametrize("tag", ["ddpm", "dddpm_ae", "dddpm"])
def test_train_step(tag):
    g = golden("g6_train")
    if tag == "ddpm":
        cfg = ddpm_cfg(32, 3, 16)
        shapes = {k: v for k, v in golden_keys()["dddpm_tiny_x2"].items() if k.startswith("latent_model.")}
        shapes["latent_model.downs.0.0.block1.block.0.weight"] = [32, 3, 3, 3]
        shapes["latent_model.downs.0.0.res_conv.weight"] = [32, 3, 1, 1]
        shapes["latent_model.final_conv.1.weight"] = [3, 32, 1, 1]
        shapes["latent_model.final_conv.1.bias"] = [3]
        xshape, eshape = (4, 3, 16, 16), (4, 3, 16, 16)
    else:
        cfg = dddpm_cfg(32, 32, 2)
        shapes = {k: v for k, v in golden_keys()["dddpm_tiny_x2"].items() if k not in syn.SCHEDULE_KEYS}
        xshape, eshape = (4, 3, 32, 32), (4, 8, 8, 8)
    params = det_state(shapes)
    buf = D.schedule_buffers("linear", 1000)

    def objective(p, x, t, eps):
        if tag == "ddpm":
            return TR.ddpm_objective(p, buf, cfg, x, t, eps)
        fn = R.dddpm_ae_losses if tag == "dddpm_ae" else R.dddpm_losses
        return fn(p, buf, cfg, x, t, eps)[0]

    probe = [str(n) for n in g[f"{tag}_probe_names"]]
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    lr = 2e-4
    ema = None
    for step in range(2):
        mbs = []
        for mb in range(2):
            x = syn.synthetic_input(xshape, f"g6.{tag}.x{step}{mb}")
            t = torch.tensor([0, 40 + step, 500, 999 - mb])
            eps = syn.synthetic_normal(eshape, f"g6.{tag}.eps{step}{mb}")
            mbs.append((x, t, eps))
        grads, objs = TR.accumulate_grads(params, objective, mbs)
        assert np.allclose(objs, g[f"{tag}_obj{step}"], rtol=2e-5), (objs, g[f"{tag}_obj{step}"])
        if step == 0:
            for n in probe:
                assert rel_err(grads[n], g[f"{tag}_grad_{n}"]) < 5e-4, n
        clipped, total = TR.clip_grads(grads)
        assert abs(float(total) / float(g[f"{tag}_gradnorm{step}"]) - 1) < 5e-4
        for k in params:
            params[k], m[k], v[k] = TR.adam_step(params[k], clipped[k], m[k], v[k], step + 1, lr)
        # Adam's first steps are ~ lr * g / (|g| + 1e-8): where |g| is at round-off level the update is
        # ill-conditioned, so bound the worst element by a fraction of lr and the mean tightly.
        for n in probe:
            d = np.abs(params[n].numpy() - g[f"{tag}_param{step}_{n}"])
            assert d.max() < 0.25 * lr and d.mean() < 2e-3 * lr, (n, d.max(), d.mean())
        if step == 0:
            ema = {k: p.clone() for k, p in params.items()}      # EMA reset == copy (trainer_ddpm.py:108-109)
        else:
            ema = TR.ema_update(ema, params, 0.995)
            for n in probe:
                d = np.abs(ema[n].numpy() - g[f"{tag}_ema_{n}"])
                assert d.max() < 0.25 * lr and d.mean() < 2e-3 * lr, (n, d.max(), d.mean())


# ---------------------------------------------------------------- G9 evaluation-time VLB
def test_vlb_pieces_vs_reference():
    g = golden("g9_test_losses")
    x = syn.synthetic_input((2, 3, 16, 16), "g9.x").clamp(-1, 1)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    mean1 = syn.synthetic_normal((2, 3, 16, 16), "g9.mean1") * 0.5
    mean2 = syn.synthetic_normal((2, 3, 16, 16), "g9.mean2") * 0.5
    lv1 = torch.tensor([-3.0, -0.5]).view(2, 1, 1, 1)
    lv2 = torch.tensor([-2.5, -0.75]).view(2, 1, 1, 1)
    assert np.array_equal(D.normal_kl(mean1, lv1, mean2, lv2).numpy(), g["piece_kl"])
    assert np.array_equal(D.discretized_gaussian_log_likelihood(x, mean2, 0.5 * lv2).numpy(), g["piece_ll"])
    assert np.array_equal(D.flat_bits(D.normal_kl(mean1, lv1, 0.0, 0.0)).numpy(), g["piece_prior"])


def test_test_losses_vs_reference():
    """oracle restatement of ddpm.py:393-442 vs the reference's own test_losses_ (T = 50, injected draws)"""
    g = golden("g9_test_losses")
    cfg = ddpm_cfg(32, 3, 16, T=50)
    keys = {k: v for k, v in golden_keys()["ddpm_c3"].items()}
    from oracle import unet_ref as UU
    from models import DDPM, Unet
    m = DDPM(cfg, Unet(cfg), "cpu", 3)
    sd = det_state({k: tuple(v.shape) for k, v in m.state_dict().items()})
    buf = D.schedule_buffers("linear", 50)
    x = syn.synthetic_input((2, 3, 16, 16), "g9.x").clamp(-1, 1)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])
    noises = [syn.synthetic_normal((2, 3, 16, 16), f"g9.eps{k}") for k in range(50)]
    res = D.test_losses(buf, lambda a, b: UU.unet_forward(sd, cfg, a, b, pre="latent_model."), x, noises, 50)
    for k in ("vlb_t", "prior", "vlb", "L_simple_t", "L_simple"):
        assert rel_err(res[k], g[f"simple_{k}"]) < 2e-5, k
    assert np.array_equal(g["simple_vlb_t"], g["hybrid_vlb_t"])     # the hybrid detach does not change values
