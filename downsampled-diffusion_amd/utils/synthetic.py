"""Deterministic synthetic weights and inputs (no datasets / checkpoints exist offline).

Every tensor of a reference-shaped ``state_dict`` is filled from a counter-based
closed form (a 64-bit integer mix of the element index and a CRC of the key), so
the same values can be produced on any box, by the golden generator
(tools/gen_golden.py) and by the tests/bench, without shipping weight files.
This is SURVEY.md section 8c "G0 weights formula".

Scales are fan-in based so activations stay O(1) through the 61 conv layers.
"""
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised over uint64 arrays."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return x ^ (x >> np.uint64(31))


def uniform_pm1(n: int, key: str, salt: int = 0) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (key, salt, index)."""
    base = np.uint64(zlib.crc32(key.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (base << np.uint64(20))
        h = _mix64(idx)
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def fill_tensor(key: str, shape, salt: int = 0) -> torch.Tensor:
    """Deterministic fp32 tensor for one state_dict entry, scaled by its role."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform_pm1(n, key, salt)
    leaf = key.split(".")[-1]
    if leaf in ("g",):                                  # channel-LayerNorm gain (1,C,1,1)
        v = 1.0 + 0.1 * u
    elif leaf in ("b",):                                # channel-LayerNorm bias
        v = 0.1 * u
    elif leaf == "bias":
        v = 0.1 * u
    elif leaf == "weight" and len(shape) == 1:          # GroupNorm affine gain
        v = 1.0 + 0.1 * u
    elif leaf == "weight":
        if len(shape) == 4 and ".conv." in key and ("ups." in key) and shape[2] == 4:
            fan_in = shape[0] * 4                       # ConvTranspose k4 s2: 4 taps reach each output
        else:
            fan_in = int(np.prod(shape[1:]))
        v = u * np.sqrt(3.0 / fan_in)                   # unit-variance-preserving uniform
    else:
        v = u
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state_dict(state_dict, salt: int = 0, skip=()):
    """Return a new dict with every tensor replaced by its deterministic fill.

    Keys listed in ``skip`` (e.g. the DDPM schedule buffers) keep their values.
    """
    out = {}
    for k, v in state_dict.items():
        if k in skip or not torch.is_floating_point(v):
            out[k] = v.clone()
        else:
            out[k] = fill_tensor(k, v.shape, salt).to(v.dtype)
    return out


SCHEDULE_KEYS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def synthetic_input(shape, key: str, salt: int = 0, scale: float = 1.0) -> torch.Tensor:
    """Deterministic input tensor in [-scale, scale)."""
    n = int(np.prod(shape))
    return torch.from_numpy((uniform_pm1(n, "input:" + key, salt) * scale).astype(np.float32).reshape(shape))


def synthetic_normal(shape, key: str, salt: int = 0) -> torch.Tensor:
    """Deterministic ~N(0,1) tensor (Box-Muller on the counter-based uniforms)."""
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = (uniform_pm1(m, "n1:" + key, salt) + 1.0) * 0.5
    u2 = (uniform_pm1(m, "n2:" + key, salt) + 1.0) * 0.5
    r = np.sqrt(-2.0 * np.log(np.maximum(u1, 1e-12)))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return torch.from_numpy(z.astype(np.float32).reshape(shape))
