"""Filesystem locations (reference utils/paths.py:1-8 hard-codes a DTU cluster; here every path is an
environment variable with a local default)."""
import os

WORK_DIR = os.environ.get('DDPM_WORK_DIR', './results/')
SAMPLE_DIR = os.environ.get('DDPM_SAMPLE_DIR', os.path.join(WORK_DIR, 'samples/'))
SAMPLE_LATENT_DIR = os.environ.get('DDPM_SAMPLE_LATENT_DIR', os.path.join(WORK_DIR, 'samples_latent/'))
CHECKPOINT_DIR = os.environ.get('DDPM_CHECKPOINT_DIR', os.path.join(WORK_DIR, 'checkpoints/'))
REFERENCE_DIR = os.environ.get('DDPM_REFERENCE_DIR', os.path.join(WORK_DIR, 'reference/'))
LOGGING_DIR = os.environ.get('DDPM_LOGGING_DIR', './results/logging/')
DATA_DIR = os.environ.get('DDPM_DATA_DIR', '../data')
