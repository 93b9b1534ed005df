"""Sampler output stage (reference utils/eval_helpers.py:37-41)."""
import numpy as np

from .utils import min_max_norm_image


def fix_samples(samples):
    """Per-image min-max -> [0,255] -> host NHWC float32: the on-disk format of generate_model_samples.py."""
    samples = min_max_norm_image(samples) * 255.
    return np.moveaxis(samples.cpu().numpy(), 1, -1)
