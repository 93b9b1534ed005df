"""Trainer base with the reference's constructor and attributes (reference trainers/trainer.py:10-115)."""
import json
import os

import numpy as np
import torch

from parallel import is_main_rank
from utils import LOGGING_DIR, reduce_mean
from .optim import FusedAdam
from .train_helpers import RunLogger


class Trainer(object):
    def __init__(self, config: dict, model, train_loader, val_loader=None, device: str = 'cpu', wandb_name: str = 'tmp',
                 mute: bool = True, n_channels: int = None, n_samples: int = 25):
        self.lr = config['lr']
        self.n_steps = config['n_steps']
        self.batch_size = config['batch_size']
        self.image_size = config['image_size']
        self.name = config['model']
        self.config = config
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.device = device
        self.wandb_name = wandb_name
        self.mute = mute
        self.n_channels = n_channels
        self.n_samples = config.get('n_samples', n_samples)       # `--n_samples` extension

        # trainer.py:47-51
        self.n_rows = np.sqrt(self.n_samples).astype(int)
        if self.n_rows ** 2 != self.n_samples:
            raise ValueError(f'Number of samples ({self.n_samples}) has to be a square number.')
        if self.n_samples > self.batch_size:
            raise ValueError(f'Number of samples ({self.n_samples}) has to be lower than batch size ({self.batch_size}).')

        self.loss_handle = reduce_mean
        self.train_losses = []
        self.x_dim = int(self.n_channels * self.image_size * self.image_size)

        if str(device) == 'cpu' or not torch.cuda.is_available():
            raise RuntimeError("the HIP training path needs a ROCm device (no CPU fallback)")
        self.model = model.to(self.device)
        # Adam(params, lr) with torch defaults (trainer.py:69), on flat buffers, clip_grad_norm_(1.0) folded into step()
        self.opt = FusedAdam(self.model, lr=self.lr, max_grad_norm=1.0)
        self.model._flat_params = self.opt.fp
        self.logger = None

    def save_losses(self) -> None:
        if not is_main_rank():
            return
        file_path = os.path.join(LOGGING_DIR, f'loss_{self.name}_{self.config["dataset"]}.json')
        print(f'Saving losses to file {file_path}')
        os.makedirs(LOGGING_DIR, exist_ok=True)
        with open(file_path, 'w') as f:
            json.dump(self.train_losses, f)

    def init_wandb(self) -> None:
        """trainer.py:78-92 with the JSONL logger standing in for wandb (same resume-by-id behaviour)."""
        # rank 0 owns the JSONL log and the checkpoint file; the run id is rank 0's, shared so every rank names the same paths
        self.logger = RunLogger(LOGGING_DIR, self.wandb_name, self.config, run_id=self.config.get('wandb_id'), enabled=is_main_rank())
        self.wandb_id = self._shared_run_id(self.logger.id)
        self.logger.id = self.wandb_id
        self.config['wandb_id'] = self.wandb_id
        os.makedirs(LOGGING_DIR, exist_ok=True)
        self.checkpoint_name = os.path.join(LOGGING_DIR, f'checkpoint_{self.name}_{self.wandb_id}.pt')

    @staticmethod
    def _shared_run_id(run_id):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [run_id]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return run_id

    def finalize(self) -> None:
        """trainer.py:94-99; the local checkpoint is KEPT (the reference deletes it after its wandb upload)."""
        self.save_checkpoint()
        self.logger.finish()
        print(f"Training of {self.name} completed!")

    def train(self):
        self.init_wandb()
        losses = self.train_loop()
        self.finalize()
        return losses

    def train_loop(self):
        raise NotImplementedError('Implement in subclass.')

    def load_checkpoint(self):
        raise NotImplementedError('Implement in subclass.')

    def save_checkpoint(self):
        raise NotImplementedError('Implement in subclass.')
