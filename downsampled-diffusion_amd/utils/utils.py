"""Host-side helpers with the reference's names (reference utils/utils.py:5-54)."""
import numpy as np


def modify_config(config, model_config):
    """utils.py:5-8: overlay model_config onto config (in place) and return it."""
    config.update(model_config)
    return config


def min_max_norm_batch(x):
    """utils.py:11-13: min-max over the whole batch."""
    lo, hi = x.min(), x.max()
    return (x - lo) / (hi - lo)


def min_max_norm_image(x):
    """utils.py:16-24: min-max per image."""
    flat = x.reshape(x.shape[0], -1)
    lo = flat.min(dim=1).values.reshape(-1, 1, 1, 1)
    hi = flat.max(dim=1).values.reshape(-1, 1, 1, 1)
    return (x - lo) / (hi - lo)


def _non_batch_dims(x):
    return list(range(1, x.dim()))


def reduce_mean(x):
    """utils.py:26-31"""
    return x.mean(dim=_non_batch_dims(x))


def reduce_sum(x):
    """utils.py:34-40"""
    return x.sum(dim=_non_batch_dims(x))


def flat_bits(x):
    """utils.py:43-48: mean over non-batch dims in bits (divide by ln 2)."""
    return reduce_mean(x) / np.log(2.)


def get_model_state_dict(save_data):
    """utils.py:51-54: prefer the EMA weights of a checkpoint."""
    return save_data['ema_model'] if 'ema_model' in save_data else save_data['model']


def load_checkpoint_file(path, map_location='cpu'):
    """The ONE checkpoint reader of the CLIs (generate_model_samples.py, train_from_checkpoint.py).  Checkpoints written by
    the reference trainer (trainers/trainer_ddpm.py:49-62) carry numpy scalars in `train_losses` (np.mean) and a config dict
    with tuples: torch >= 2.6's default weights_only=True refuses to unpickle them, so the full unpickler is requested
    explicitly -- these are the user's own training checkpoints, exactly what the reference's torch.load read."""
    import torch
    return torch.load(path, map_location=map_location, weights_only=False)
