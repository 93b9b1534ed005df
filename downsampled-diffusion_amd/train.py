#!/usr/bin/env python3
"""Training entry point with the reference's command line (reference train.py:1-85):

    python train.py -m ddpm -d celeba_hq -bs 32 -is 256 -downsample 3 -e 800000 -mute

Model hyper-parameters are the reference's CONFIG_MODEL dicts (train.py:19-46); `-downsample N > 0` switches to dDDPM.
Multi-GPU: `torchrun --nproc-per-node N train.py ...` (data parallel, one gradient all-reduce per optimiser step).
"""
import json

from models import MODEL_NAMES
from trainers import setup_trainer
from utils import DATASETS, get_args, modify_config

DATA_ROOT = '../data/'
WANDB_PROJECT = 'ddpm-test'

CONFIG = {
    'lr': 1e-3,
    'rnd_flip': False,
}

CONFIG_MODEL = {
    'ddpm': {
        'lr': 2e-4,
        'unet_chan': 128,
        'unet_dims': (1, 2, 2, 2),
        'unet_dropout': 0.1,
        'T': 1000,
        'loss_type': 'simple',
        'beta_schedule': 'linear',
        'ema_decay': 0.995,
        'loss_flat': 'sum',
        'val_split': 0,
    },
    'dddpm': {
        'd_mode': 'convolutional_res',
        'u_mode': 'convolutional_res',
        'd_dropout': 0,
        'd_chans': 64,
        'd_n_blocks': 3,
        'u_n_blocks': 3,
        'unet_in': 8,
        'ae_loss': True,
        't_rec_max': 100,
        'force_latent': True,
    },
}

if __name__ == '__main__':
    config, mute = get_args(CONFIG, DATASETS, MODEL_NAMES)
    t_override = config.pop('T_override', None)
    config = modify_config(config, CONFIG_MODEL[config['model']])
    if t_override is not None:
        config['T'] = t_override
    if config['model'] == 'ddpm' and config['n_downsamples'] > 0:
        config['model'] = 'dddpm'
        config = modify_config(config, CONFIG_MODEL['dddpm'])

    trainer, config = setup_trainer(config, mute, DATA_ROOT, WANDB_PROJECT, 0)
    print('\nTraining configuration dict:')
    print(json.dumps(config, sort_keys=False, indent=4, default=str) + '\n')
    _ = trainer.train()
    print("train.py script finished!")
