"""Variance schedules (reference models/diffusion/beta_schedule.py:5-33), float64 numpy on the host."""
import numpy as np


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """'linear': Ho et al. rescaled to any T (betas = linspace * 1000/T);
    'cosine': improved-DDPM alpha-bar, betas clipped to [0, 0.999]."""
    if schedule == "linear":
        k = 1000 / n_timestep
        return np.linspace(k * linear_start, k * linear_end, n_timestep, dtype=np.float64)
    if schedule == "cosine":
        import torch  # the reference evaluates the cosine in torch float64; keep its rounding
        s = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        abar = torch.cos(s / (1 + cosine_s) * np.pi / 2).pow(2)
        abar = abar / abar[0]
        return np.clip((1 - abar[1:] / abar[:-1]).numpy(), a_min=0, a_max=0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")
