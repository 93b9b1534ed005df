"""Oracle: dDDPM down/up-sampling ConvResNet and dDDPM losses (test infrastructure only).

Follows reference models/downsampled/convblocks.py:92-159, models/downsampled/wrapper.py:6-59
(mode 'convolutional_res' only -- the mode train.py:34-35 selects) and
models/diffusion/dddpm.py:76-177.
"""
import torch
import torch.nn.functional as F

from .diffusion_ref import loss_ddpm, predict_x_from_eps, q_sample
from .unet_ref import mish, unet_forward


def conv_res_block(sd, pre, x, mode, mask=None):
    """convblocks.py:112-130: Mish->1x1->Mish->3x3->Mish->3x3->Mish->1x1, Dropout2d, +x, then resample.

    mode: 'down' (avg_pool2d 2), 'up' (nearest x2) or None.  mask: None = Dropout2d(p=0) / eval mode (identity); else the
    nn.Dropout2d draw of convblocks.py:121-124 restated with an injected mask [B, C] of {0, 1 / (1 - p)} (whole channels of a sample).
    """
    h = F.conv2d(mish(x), sd[pre + "c1.weight"], sd[pre + "c1.bias"])
    h = F.conv2d(mish(h), sd[pre + "c2.weight"], sd[pre + "c2.bias"], padding=1)
    h = F.conv2d(mish(h), sd[pre + "c3.weight"], sd[pre + "c3.bias"], padding=1)
    h = F.conv2d(mish(h), sd[pre + "c4.weight"], sd[pre + "c4.bias"])
    if mask is not None:
        h = h * mask[:, :, None, None]
    out = x + h
    if mode == "up":
        out = F.interpolate(out, scale_factor=2)
    elif mode == "down":
        out = F.avg_pool2d(out, kernel_size=2, stride=2)
    return out


def conv_res_net(sd, pre, x, n_levels, n_blocks, upsample, masks=None):
    """convblocks.py:133-159: 1x1 explode, per level [resampling block + (n_blocks-1) plain], 1x1 condense.
    masks: {block key prefix: Dropout2d mask} for train-mode parity with injected draws (see conv_res_block)."""
    masks = masks or {}
    x = F.conv2d(x, sd[pre + "conv.0.weight"], sd[pre + "conv.0.bias"])
    i = 1
    for _ in range(n_levels):
        x = conv_res_block(sd, f"{pre}conv.{i}.", x, "up" if upsample else "down", masks.get(f"{pre}conv.{i}."))
        i += 1
        for _ in range(n_blocks - 1):
            x = conv_res_block(sd, f"{pre}conv.{i}.", x, None, masks.get(f"{pre}conv.{i}."))
            i += 1
    return F.conv2d(x, sd[f"{pre}conv.{i}.weight"], sd[f"{pre}conv.{i}.bias"])


def rescaled_downsample(sd, cfg, x, masks=None):
    """dddpm.py:92-101: z = tanh(downsample(x)) when force_latent."""
    z = conv_res_net(sd, "downsample.", x, cfg["n_downsamples"], cfg["d_n_blocks"], False, masks)
    return torch.tanh(z) if cfg["force_latent"] else z


def rescaled_upsample(sd, cfg, z, masks=None):
    """dddpm.py:103-112."""
    x = conv_res_net(sd, "upsample.", z, cfg["n_downsamples"], cfg["u_n_blocks"], True, masks)
    return torch.tanh(x) if cfg["force_latent"] else x


def loss_recon(sd, cfg, x, z_hat, t):
    """dddpm.py:114-120: per-sample sum((x - up(z_hat))^2), zeroed where t >= t_rec_max."""
    x_hat = rescaled_upsample(sd, cfg, z_hat)
    per = (x - x_hat) ** 2
    dims = list(range(1, per.dim()))
    per = per.sum(dim=dims) if cfg["loss_flat"] == "sum" else per.mean(dim=dims)
    t_rec_max = cfg["T"] - 1 if cfg["t_rec_max"] == -1 else cfg["t_rec_max"]
    return torch.where(t < t_rec_max, per, torch.zeros_like(per))


def dddpm_ae_losses(sd, buf, cfg, x, t, eps):
    """dddpm.py:155-177 (DownsampleDDPMAutoencoder.losses) with injected t and eps."""
    z = rescaled_downsample(sd, cfg, x)
    l_rec = loss_recon(sd, cfg, x, z, t)
    z = z.detach()
    z_t = q_sample(buf, z, t, eps)
    eps_hat = unet_forward(sd, cfg, z_t, t, pre="latent_model.")
    l_ddpm = loss_ddpm(buf, eps, eps_hat, t, cfg["loss_type"], cfg["loss_flat"])
    obj = (l_ddpm + l_rec).mean()
    return obj, {"latent": l_ddpm.mean(), "recon": l_rec.mean()}


def dddpm_losses(sd, buf, cfg, x, t, eps):
    """dddpm.py:122-143 (DownsampleDDPM.losses) with injected t and eps."""
    z = rescaled_downsample(sd, cfg, x)
    z_t = q_sample(buf, z, t, eps)
    eps_hat = unet_forward(sd, cfg, z_t, t, pre="latent_model.")
    l_ddpm = loss_ddpm(buf, eps, eps_hat, t, cfg["loss_type"], cfg["loss_flat"])
    z_hat = predict_x_from_eps(buf, z_t, t, eps_hat, clip=False)
    l_rec = loss_recon(sd, cfg, x, z_hat, t)
    obj = (l_ddpm + l_rec).mean()
    return obj, {"latent": l_ddpm.mean(), "recon": l_rec.mean()}
