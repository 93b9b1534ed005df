"""Downsampled DDPM (dDDPM): a DDPM over tanh-squashed latents produced by a learned ConvResNet encoder,
decoded by a ConvResNet decoder (reference models/diffusion/dddpm.py:11-177).  Same constructor and
return conventions: ``sample`` -> (x, z), ``forward`` -> (objective, {'latent', 'recon'}).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ddk import ops
from models.downsampled import get_downsampling, get_upsampling
from .ddpm import DDPM


class DownsampleDDPM(DDPM):
    def __init__(self, config: dict, denoise_model: nn.Module, device: str, color_channels: int = 3):
        super().__init__(config, denoise_model, device, color_channels)
        self.t_rec_max = int(self.timesteps - 1) if config['t_rec_max'] == -1 else config['t_rec_max']
        self.x_shape = [self.in_channels, self.image_size, self.image_size]
        self.force_latent = config['force_latent']
        unet_in = config['unet_in']
        self.dim_reduc = np.power(2, config['n_downsamples']).astype(int)
        z_size = int(self.image_size / self.dim_reduc)
        self.sample_shape = [unet_in, z_size, z_size]
        assert unet_in >= self.in_channels, (f'Input channels to DDPM-Unet {unet_in} should be equal or larger to '
                                             f'data color channels {self.in_channels}.')
        self.downsample = get_downsampling(config, self.x_shape)
        self.upsample = get_upsampling(config, self.x_shape)

    # ------------------------------------------------------------------ encoder / decoder (dddpm.py:92-112)
    def rescaled_downsample(self, x):
        """z = tanh(downsample(x)) (tanh only when force_latent)."""
        self._check_device(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.downsample.parameters()):
            from trainers.autograd_unet import resnet_forward_autograd
            return resnet_forward_autograd(self.downsample, x, self.force_latent)
        z = ops.nhwc_to_nchw(self.downsample.forward_nhwc(ops.nchw_to_nhwc(x.contiguous().float(), ops.pad32(x.shape[1])),
                                                          final_tanh=self.force_latent))
        assert list(z.shape)[1:] == self.sample_shape, f'mismatch between {list(z.shape)[1:]} and {self.sample_shape}'
        return z

    def rescaled_upsample(self, z):
        """x = tanh(upsample(z))."""
        self._check_device(z)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.upsample.parameters()):
            from trainers.autograd_unet import resnet_forward_autograd
            return resnet_forward_autograd(self.upsample, z, self.force_latent)
        x = ops.nhwc_to_nchw(self.upsample.forward_nhwc(ops.nchw_to_nhwc(z.contiguous().float(), ops.pad32(z.shape[1])),
                                                        final_tanh=self.force_latent))
        assert list(x.shape)[1:] == self.x_shape, f'mismatch between {list(x.shape)[1:]} and {self.x_shape}'
        return x

    # ------------------------------------------------------------------ sampling (dddpm.py:76-90)
    @torch.no_grad()
    def sample(self, batch_size=16, every=1, early_stop=None):
        z_sample = self.p_sample_loop((batch_size, *self.sample_shape), every, early_stop)
        x_sample = self.rescaled_upsample(z_sample)
        assert list(z_sample.shape)[1:] == self.sample_shape
        assert list(x_sample.shape)[1:] == self.x_shape
        return x_sample, z_sample

    @torch.no_grad()
    def reconstruct(self, x, n):
        """dddpm.py:33-74 (visualisation only)."""
        assert x.shape[0] >= n, f'batch size ({x.shape[0]}) is below {n}'
        x = x[:n]
        t = torch.linspace(0, self.timesteps - 1, n, device=x.device, dtype=torch.long)
        z = self.rescaled_downsample(x)
        eps = torch.randn_like(z)
        z_t = self.q_sample(z, t, eps)
        eps_hat = self.latent_model(z_t, t)
        z_recon = self.predict_x_from_eps(z_t, t, eps_hat, clip=False)
        x_recon = self.rescaled_upsample(z_recon)
        assert list(x_recon.shape)[1:] == self.x_shape
        return x_recon, z_recon

    # ------------------------------------------------------------------ losses (dddpm.py:114-143)
    def loss_recon(self, x, z_hat, t):
        x_hat = self.rescaled_upsample(z_hat)
        assert x_hat.shape == x.shape, f'mismatch between {x_hat.shape} and {x.shape}'
        loss = self._per_sample_sq_err(x, x_hat)
        return torch.where(t < self.t_rec_max, loss, torch.zeros_like(loss))

    def losses(self, x, t):
        z = self.rescaled_downsample(x)
        eps = torch.randn_like(z)
        z_t = self.q_sample(z, t, eps)
        eps_hat = self.latent_model(z_t, t)
        L_ddpm = self.loss_ddpm(eps, eps_hat, t)
        z_hat = self.predict_x_from_eps(z_t, t, eps_hat, clip=False)
        L_rec = self.loss_recon(x, z_hat, t)
        obj = (L_ddpm + L_rec).mean()
        return obj, {'latent': L_ddpm.mean(), 'recon': L_rec.mean()}

    @torch.no_grad()
    def test_losses(self, x):
        return self.test_losses_(self.rescaled_downsample(x))


RECON_SIDE_STREAM = True          # see DownsampleDDPMAutoencoder.losses; the tests switch it off to compare
FUSED_OBJECTIVE = True            # idem: the objective of the 'simple' loss in one launch
_recon_streams = {}               # device -> torch.cuda.Stream (module level: a stream must not end up in a deepcopy of the model)


class DownsampleDDPMAutoencoder(DownsampleDDPM):
    """Reconstruction loss taken directly through the autoencoder, latent detached for the DDPM term
    (dddpm.py:151-177; selected by ae_loss=True, train.py:44)."""

    def __init__(self, config: dict, denoise_model: nn.Module, device: str, color_channels: int = 3):
        super().__init__(config, denoise_model, device, color_channels)

    def losses(self, x, t):
        z = self.rescaled_downsample(x)
        # The reconstruction branch (decoder + loss, and with it the backward of both) does not meet the denoiser branch before the two
        # losses are added (the latent is detached): in training on the device it runs on its own stream -- its large memory-bound
        # launches beside the UNet's small latency-bound ones.  autograd runs each backward on the stream of its forward.
        fork = RECON_SIDE_STREAM and x.is_cuda and torch.is_grad_enabled()
        # 'simple' loss in training on the device: objective and the two report values from the per-sample losses in one launch
        # (ddk_ae_objective) instead of where / add / three means and their autograd counterparts
        fused = FUSED_OBJECTIVE and self.L == 'simple' and x.is_cuda and torch.is_grad_enabled()
        recon = (lambda: self._per_sample_sq_err(x, self.rescaled_upsample(z))) if fused else (lambda: self.loss_recon(x, z, t))
        if fork:
            main = torch.cuda.current_stream()
            side = _recon_streams.get(x.device)
            if side is None:
                side = _recon_streams[x.device] = torch.cuda.Stream(device=x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                L_rec = recon()
        else:
            L_rec = recon()
        z = z.detach()
        eps = torch.randn_like(z)
        z_t = self.q_sample(z, t, eps)
        eps_hat = self.latent_model(z_t, t)
        L_ddpm = self._per_sample_sq_err(eps, eps_hat) if fused else self.loss_ddpm(eps, eps_hat, t)
        if fork:
            main.wait_stream(side)
        if fused:
            from ddk import autograd as AG
            obj, latent, rec = AG.AEObjectiveFn.apply(L_ddpm, L_rec, t.contiguous(), math.ceil(self.t_rec_max))    # t < t_rec_max for integer t, also for a fractional setting
            return obj, {'latent': latent, 'recon': rec}
        obj = (L_ddpm + L_rec).mean()
        return obj, {'latent': L_ddpm.mean(), 'recon': L_rec.mean()}
