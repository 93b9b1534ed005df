#!/usr/bin/env python3
"""Resume training from a checkpoint (reference train_from_checkpoint.py:11-24): rebuild the trainer from the stored
config, load optimiser / model / EMA / step, continue."""
import argparse
import os

from trainers import setup_trainer
from utils import LOGGING_DIR, load_checkpoint_file

DATA_ROOT = '../data/'
WANDB_PROJECT = 'ddpm-test'

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('checkpoint', help='file name inside LOGGING_DIR, or a path')
    ap.add_argument('-e', dest='n_steps', type=int, default=None, help='new total number of train steps')
    ap.add_argument('-mute', action='store_true')
    args = ap.parse_args()
    path = args.checkpoint if os.path.exists(args.checkpoint) else os.path.join(LOGGING_DIR, args.checkpoint)
    checkpoint = load_checkpoint_file(path)
    config = checkpoint['config']
    if args.n_steps is not None:
        config['n_steps'] = args.n_steps
    trainer, config = setup_trainer(config, args.mute, DATA_ROOT, WANDB_PROJECT, 0)
    trainer.load_checkpoint(checkpoint)
    if args.n_steps is not None:
        trainer.n_steps = args.n_steps
    trainer.train()
    print("train_from_checkpoint.py script finished!")
