"""setup_trainer with the reference's signature (reference trainers/wrapper.py:10-49)."""
import os
from functools import reduce
from operator import mul

import torch

from models import DDPM, DownsampleDDPM, DownsampleDDPMAutoencoder, Unet
from parallel import broadcast_module_, init_from_env
from utils import get_color_channels, get_dataloader, seed_everything
from .trainer_ddpm import TrainerDDPM, TrainerDownsampleDDPM


def setup_trainer(config: dict, mute: bool, data_root: str, wandb_project: str = 'tmp', seed: int = None):
    """Instantiate a trainer for the model the config names.  Under torchrun every rank builds the same model (same
    seed), rank 0's weights are broadcast once, and each rank draws its own data shard."""
    rank, world = init_from_env()
    seed_everything(seed)
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm device: the HIP training path has no CPU fallback")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = f'cuda:{local}' if world > 1 else 'cuda'
    train_loader, val_loader = get_dataloader(config, 'cuda', True, data_root, config['val_split'])
    color_channels = get_color_channels(config['dataset'])
    x_shape = [color_channels, config['image_size'], config['image_size']]
    _ = reduce(mul, x_shape, 1)

    train_args = [train_loader, val_loader, device, wandb_project, mute, color_channels]
    if config['model'] == 'ddpm':
        print('Instantiating DDPM')
        config['unet_in'] = color_channels
        latent_model = Unet(config)
        model = DDPM(config, latent_model, device, color_channels)
        cls = TrainerDDPM
    elif config['model'] == 'dddpm':
        print('Instantiating DownsampledDDPM')
        latent_model = Unet(config)
        mcls = DownsampleDDPMAutoencoder if config['ae_loss'] else DownsampleDDPM
        model = mcls(config, latent_model, device, color_channels)
        cls = TrainerDownsampleDDPM
    else:
        raise NotImplementedError('Specified model not implemented.')
    model = model.to(device)
    broadcast_module_(model, src=0)
    if seed is not None and world > 1:
        seed_everything(seed + 1000 * (rank + 1))       # distinct data / noise / dropout streams per rank
    trainer = cls(config, model, *train_args)
    config['model_size'] = sum(p.numel() for p in model.parameters())
    return trainer, config
