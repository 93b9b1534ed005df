"""Oracle: one optimiser step of the reference training loop (test infrastructure only).

Follows reference trainers/trainer_ddpm.py:107-158 (2 accumulation micro-batches, obj/2 backward,
clip_grad_norm_(params, 1.0), Adam step, zero_grad, EMA schedule), trainers/trainer.py:69
(Adam(params, lr) with torch defaults) and trainers/ema.py:33-44.  Gradients come from torch
autograd over the functional oracle forward, so no reference or product code is involved.
"""
import math

import torch

from .diffusion_ref import loss_ddpm, q_sample
from .unet_ref import unet_forward

GRAD_ACCUM = 2           # trainer_ddpm.py:35
CLIP_NORM = 1.0          # trainer_ddpm.py:142
ADAM_BETAS = (0.9, 0.999)
ADAM_EPS = 1e-8
EMA_START = 2000         # trainer_ddpm.py:41
EMA_EVERY = 10           # trainer_ddpm.py:42


def ddpm_objective(sd, buf, cfg, x, t, eps, pre="latent_model."):
    """ddpm.py:290-315 with injected t, eps: q_sample -> UNet -> loss_ddpm."""
    x_t = q_sample(buf, x, t, eps)
    eps_hat = unet_forward(sd, cfg, x_t, t, pre=pre)
    return loss_ddpm(buf, eps, eps_hat, t, cfg["loss_type"], cfg["loss_flat"])


def accumulate_grads(params, objective_fn, micro_batches):
    """trainer_ddpm.py:118-131: sum over micro-batches of grad(obj / GRAD_ACCUM)."""
    names = list(params.keys())
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    objs = []
    for mb in micro_batches:
        obj = objective_fn(leaves, *mb)
        g = torch.autograd.grad(obj / GRAD_ACCUM, [leaves[k] for k in names], allow_unused=True)
        for k, gi in zip(names, g):
            if gi is not None:
                grads[k] += gi
        objs.append(float(obj.detach()))
    return grads, objs


def global_grad_norm(grads):
    """torch.nn.utils.clip_grad_norm_: 2-norm of the per-tensor 2-norms."""
    return torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()


def clip_grads(grads, max_norm=CLIP_NORM):
    """clip_grad_norm_: coef = max_norm / (total + 1e-6), clamped to 1."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adam_step(p, g, m, v, step, lr):
    """torch.optim.Adam (no amsgrad, no weight decay), single tensor; step counts from 1."""
    b1, b2 = ADAM_BETAS
    m = m * b1 + (1 - b1) * g
    v = v * b2 + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + ADAM_EPS
    return p - (lr / bc1) * m / denom, m, v


def ema_update(ema, params, decay):
    """ema.py:36-44: p_ema = p_ema*decay + (1-decay)*p over parameters."""
    return {k: ema[k] * decay + (1 - decay) * params[k] for k in params}


def ema_schedule(step):
    """trainer_ddpm.py:107-111: 'reset' while step < 2000, 'update' every 10th step after, else None."""
    if step < EMA_START:
        return "reset"
    if step % EMA_EVERY == 0:
        return "update"
    return None
