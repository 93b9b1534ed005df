"""DDPM / dDDPM training loops with the reference's semantics (reference trainers/trainer_ddpm.py:13-265):
2 accumulation micro-batches, (obj / 2).backward(), clip_grad_norm_(1.0), Adam step, zero_grad, EMA (copy while
step < 2000, lerp every 10th step after), eval toggles, checkpoint dict schema.

MI355X specifics: forward + backward are HIP kernels (trainers/autograd_unet.py); clip + Adam + EMA are three fused
kernels over flat buffers; under torchrun the flat gradient bucket is all-reduced ONCE per optimiser step over RCCL,
before the clip, so every rank applies the identical update; losses are read back once per step, not per micro-batch.
"""
import os

import numpy as np
import torch

from parallel import all_reduce_flat_
from utils import LOGGING_DIR, min_max_norm_image
from .ema import EMA
from .trainer import Trainer
from .train_helpers import cycle


def _invalidate(model):
    for m in model.modules():
        if hasattr(m, "invalidate_plan"):
            m.invalidate_plan()


class TrainerDDPM(Trainer):
    def __init__(self, config: dict, model, train_loader, val_loader=None, device: str = 'cpu', wandb_name: str = 'tmp',
                 mute: bool = True, n_channels: int = None):
        # the EMA copy must be taken before the parameters are re-homed into the optimiser's flat buffer
        self.use_ema = config['ema_decay'] > 0
        ema = EMA(model.to(device), config['ema_decay']) if self.use_ema else None
        super().__init__(config, model, train_loader, val_loader, device, wandb_name, mute, n_channels)
        self.train_loader = cycle(self.train_loader)
        if config['val_split'] > 0:
            self.val_loader = cycle(self.val_loader)
            self.val_batch = next(self.val_loader)[0][0].repeat(self.n_samples, 1, 1, 1).to(self.device)
        else:
            self.val_batch = next(self.train_loader)[0][0].repeat(self.n_samples, 1, 1, 1).to(self.device)
        self.step = 0
        self.gradient_accumulate_every = 2
        self.logging_every = 10000
        if self.use_ema:
            self.step_start_ema = 2000
            self.update_ema_every = 10
            self.ema = ema
            self.ema.eval()
        self.name += f'_{config["T"]}'

    # ------------------------------------------------------------------ checkpoints (trainer_ddpm.py:49-72)
    def save_checkpoint(self) -> None:
        save_data = {
            'optimizer': self.opt.state_dict(),
            'model': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
            'config': self.config,
            'train_losses': self.train_losses,
            'step': self.step,
        }
        if self.use_ema:
            save_data['ema_model'] = self.ema.state_dict()
        torch.save(save_data, self.checkpoint_name)
        if self.logger:
            self.logger.save(self.checkpoint_name, policy='live')

    def load_checkpoint(self, checkpoint: dict) -> None:
        self.opt.load_state_dict(checkpoint['optimizer'])
        self.model.load_state_dict(checkpoint['model'])
        _invalidate(self.model)
        self.config = checkpoint['config']
        self.train_losses = checkpoint['train_losses']
        self.step = checkpoint['step']
        if 'ema_model' in checkpoint and self.use_ema:
            self.ema.load_state_dict(checkpoint['ema_model'])

    @torch.no_grad()
    def sample(self):
        return self.ema.sample(self.n_samples) if self.use_ema else self.model.sample(self.n_samples)

    @torch.no_grad()
    def recon(self, x):
        return self.ema.reconstruct(x, self.n_samples) if self.use_ema else self.model.reconstruct(x, self.n_samples)

    @torch.no_grad()
    def log_wandb(self, x, commit: bool = True) -> None:
        """trainer_ddpm.py:91-105: sample + reconstruction grids, stored as .npy (no torchvision / wandb offline)."""
        samples, recon = self.sample(), self.recon(x)
        samples = samples[0] if isinstance(samples, tuple) else samples
        recon = recon[0] if isinstance(recon, tuple) else recon
        log_name = f'{self.step}_{self.name}_{self.config["dataset"]}'
        os.makedirs(LOGGING_DIR, exist_ok=True)
        np.save(os.path.join(LOGGING_DIR, f'sample_{log_name}.npy'), min_max_norm_image(samples).cpu().numpy())
        np.save(os.path.join(LOGGING_DIR, f'recon_{log_name}.npy'), min_max_norm_image(recon).cpu().numpy())

    def update_ema(self):
        """trainer_ddpm.py:107-111"""
        if self.step < self.step_start_ema:
            self.ema.reset(self.model)
        elif self.step % self.update_ema_every == 0:
            self.ema.update(self.model)

    # ------------------------------------------------------------------ one optimiser step
    def _micro_batch(self):
        x, _ = next(self.train_loader)
        x = x.to(self.device, non_blocking=True)
        out = self.model(x)
        obj, extra = (out[0], out[1]) if isinstance(out, tuple) else (out, None)
        (obj / self.gradient_accumulate_every).backward()
        return obj.detach(), extra

    def optimizer_step(self):
        """all-reduce (data parallel) -> clip_grad_norm_(1.0) -> Adam -> zero_grad (trainer_ddpm.py:142-144)"""
        all_reduce_flat_(self.opt.fp.grad, average=True)
        norm = self.opt.step()
        self.opt.zero_grad()
        _invalidate(self.model)
        return norm

    def train_loop(self) -> None:
        while self.step < self.n_steps:
            self.model.train()
            objs = [self._micro_batch()[0] for _ in range(self.gradient_accumulate_every)]
            train_obj = float(torch.stack(objs).mean())          # one device->host read per step
            self.train_losses.append(train_obj)
            is_log = self.step != 0 and self.step % self.logging_every == 0
            self.logger.log({'train_obj': train_obj}, commit=(not is_log))
            self.optimizer_step()
            if self.use_ema:
                self.update_ema()
            self.model.eval()
            if is_log:
                self.save_checkpoint()
                self.log_wandb(self.val_batch)
            self.step += 1
        return self.train_losses


class TrainerDownsampleDDPM(TrainerDDPM):
    """trainer_ddpm.py:161-265: same loop, the model returns (objective, {'latent', 'recon'})."""

    def train_loop(self):
        while self.step < self.n_steps:
            self.model.train()
            objs, lats, recs = [], [], []
            for _ in range(self.gradient_accumulate_every):
                obj, extra = self._micro_batch()
                objs.append(obj / self.gradient_accumulate_every)        # the reference logs objective.item() here
                lats.append(extra['latent'].detach())
                recs.append(extra['recon'].detach())
            vals = torch.stack([torch.stack(objs).mean(), torch.stack(lats).mean(), torch.stack(recs).mean()]).tolist()
            self.train_losses.append(vals[0])
            is_log = self.step != 0 and self.step % self.logging_every == 0
            self.logger.log({'train_obj': vals[0], 'train_latent': vals[1], 'train_recon': vals[2]}, commit=(not is_log))
            self.optimizer_step()
            if self.use_ema:
                self.update_ema()
            self.model.eval()
            if is_log:
                self.save_checkpoint()
                self.log_wandb(self.val_batch)
            self.step += 1
        return self.train_losses
