#!/usr/bin/env python3
"""Reverse-step time of the other BASELINE.json sampling configs (cfg1 MNIST 32x32 bs16, cfg2 CIFAR-10 32x32 bs64, cfg3 CelebA-64
dDDPM-x2 bs64, cfg5-shape full-resolution DDPM 256x256 bs8) on one GPU, synthetic weights.  GPU-box tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops
from models import DDPM, DownsampleDDPM, Unet
from utils import synthetic as syn

DEV = "cuda"


def cfg(cin, size, T=1000, down=0):
    c = dict(unet_chan=128, unet_in=cin, unet_dims=(1, 2, 2, 2), unet_dropout=0.1, image_size=size, T=T, loss_type="simple",
             beta_schedule="linear", loss_flat="sum", ema_decay=0.995)
    if down:
        c.update(d_mode="convolutional_res", u_mode="convolutional_res", d_dropout=0, d_chans=64, d_n_blocks=3, u_n_blocks=3, unet_in=8,
                 ae_loss=True, t_rec_max=100, force_latent=True, n_downsamples=down)
    return c


def run(name, model, B, steps=100):
    model = model.to(DEV).eval()
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    C, S, _ = model.sample_shape
    T = model.timesteps
    unet = model.latent_model
    plan, tables = unet.plan(), model._tables()
    x = ops.randn((B, S, S, C), DEV, seed=1, step=T, stream_id=0)
    with torch.no_grad():
        plan.sample_nhwc(x, tables, T - 1, T - 10, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.sample_nhwc(x, tables, T - 1, T - steps, seed=1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    fl = unet.flops(B, S, S)
    print(f"{name}: {dt * 1e3:7.3f} ms per reverse step (batch {B}, latent {C}x{S}x{S}), {fl / dt / 1e12:5.1f} TFLOP/s, "
          f"{B / (T * dt):6.2f} images/s at T={T}", flush=True)


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else None     # e.g. "cfg5": that configuration alone (for a rocprofv3 --stats run)
    if only == "cfg5":
        c = cfg(3, 256); run("cfg5-shape DDPM 256x256 bs8      ", DDPM(c, Unet(c), DEV, 3), 8, steps=20)
        sys.exit(0)
    if only == "cfg2":
        c = cfg(3, 32); run("cfg2 CIFAR-10 DDPM 32x32 bs64   ", DDPM(c, Unet(c), DEV, 3), 64)
        sys.exit(0)
    if only == "cfg3":
        c = cfg(8, 64, down=2); run("cfg3 CelebA-64 dDDPM-x2 bs64     ", DownsampleDDPM(c, Unet(c), DEV, 3), 64)
        sys.exit(0)
    c = cfg(1, 32, T=200); run("cfg1 MNIST DDPM 32x32 bs16      ", DDPM(c, Unet(c), DEV, 1), 16)
    c = cfg(3, 32); run("cfg2 CIFAR-10 DDPM 32x32 bs64   ", DDPM(c, Unet(c), DEV, 3), 64)
    c = cfg(8, 64, down=2); run("cfg3 CelebA-64 dDDPM-x2 bs64     ", DownsampleDDPM(c, Unet(c), DEV, 3), 64)
    c = cfg(8, 256, down=3); run("cfg4 CelebA-HQ-256 dDDPM-x3 bs32 ", DownsampleDDPM(c, Unet(c), DEV, 3), 32)
    c = cfg(3, 256); run("cfg5-shape DDPM 256x256 bs8      ", DDPM(c, Unet(c), DEV, 3), 8, steps=20)
