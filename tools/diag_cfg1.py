import os, sys
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from helpers import ddpm_cfg, det_load
from models import DDPM, Unet
from utils import synthetic as syn
DEV = "cuda"
cfg = ddpm_cfg(128, 1, 32, T=200)
m = det_load(DDPM(cfg, Unet(cfg), DEV, 1)).eval().to(DEV)
steps, shape = 10, (2, 1, 32, 32)
x_T = syn.synthetic_normal(shape, "cfg1.xT")
noise = torch.stack([syn.synthetic_normal(shape, f"cfg1.n{k}") for k in range(steps)])
big = (16, 1, 32, 32)
xb = syn.synthetic_normal(big, "cfg1.big.xT")
nb = torch.stack([syn.synthetic_normal(big, f"cfg1.big.n{k}") for k in range(steps)])
xb[6:8] = x_T
nb[:, 6:8] = noise
plan = m.latent_model.plan()
for name, opts in (("all on", {}), ("first_gn off", {8: 0}), ("chain no edges", {7: 2}), ("chain off", {7: 0}), ("cluster off", {1: 0})):
    for k, v in ((8, 1), (7, 1), (1, 1)):
        plan.set_option(k, v)
    for k, v in opts.items():
        plan.set_option(k, v)
    got = m.p_sample_loop(shape, early_stop=200 - steps, x_T=x_T, noise=noise).cpu()
    full = m.p_sample_loop(big, early_stop=200 - steps, x_T=xb, noise=nb).cpu()
    print(f"{name:16s}: B=16 slice vs B=2 chain max diff {float((full[6:8] - got).abs().max()):.3e}", flush=True)
