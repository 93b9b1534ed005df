"""Small tensor helpers with the reference's names (models/utils/helpers.py:7-40)."""
from inspect import isfunction

import torch


def exists(x):
    return x is not None


def default(val, d):
    """helpers.py:25-28"""
    if val is not None:
        return val
    return d() if isfunction(d) else d


def get_ones_like(x):
    """helpers.py:20-22"""
    return torch.ones_like(x)


def get_identity_like(x):
    """helpers.py:10-17: an identity matrix per (sample, channel)."""
    n, c, _, w = x.shape
    return torch.eye(w, device=x.device, dtype=x.dtype).expand(n, c, w, w).contiguous()


def extract(a, t, x_shape):
    """helpers.py:31-34: per-sample gather of a [T] table, shaped to broadcast over x."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    """helpers.py:37-40: N(0,1) of `shape`; `repeat` draws one sample and tiles it over the batch."""
    if repeat:
        one = torch.randn((1, *shape[1:]), device=device)
        return one.repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)
