"""Loss helpers with the reference's names (models/utils/losses.py:7-109).

l1/l2 are thin F.* wrappers; normal_kl and the discretised Gaussian log-likelihood are only used by
the evaluation-time VLB (ddpm.py:317-446), which is outside the accelerated path and stays plain torch.
"""
import numpy as np
import torch
import torch.nn.functional as F


def l1_loss(target, output):
    """losses.py:7-9"""
    return (target - output).abs().mean()


def l2_loss(target, output, reduction='mean'):
    """losses.py:12-14"""
    return F.mse_loss(target, output, reduction=reduction)


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL( N(mean1, exp logvar1) || N(mean2, exp logvar2) ), elementwise (losses.py:17-53)."""
    as_t = lambda v, ref: v if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=ref.dtype, device=ref.device)
    ref = next(v for v in (mean1, logvar1, mean2, logvar2) if isinstance(v, torch.Tensor))
    logvar1, logvar2 = as_t(logvar1, ref), as_t(logvar2, ref)
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2)
                  + (mean1 - mean2) ** 2 * torch.exp(-logvar2))


def _std_normal_cdf(x):
    """tanh approximation of the standard normal CDF (losses.py:56-64)."""
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """log-likelihood of uint8 images rescaled to [-1,1] under a discretised Gaussian (losses.py:67-109)."""
    centered = x - means
    inv_std = torch.exp(-log_scales)
    cdf_plus = _std_normal_cdf(inv_std * (centered + 1.0 / 255.0))
    cdf_min = _std_normal_cdf(inv_std * (centered - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    delta = cdf_plus - cdf_min
    out = torch.where(x < -0.999, log_cdf_plus,
                      torch.where(x > 0.999, log_one_minus_cdf_min, torch.log(delta.clamp(min=1e-12))))
    assert out.shape == x.shape
    return out
