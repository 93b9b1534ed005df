"""Exponential moving average of the model parameters with the reference's interface
(reference trainers/ema.py:7-61): reset = copy of the live model, update = p_ema*decay + (1-decay)*p over parameters
(buffers are not averaged).  Both run on flat parameter buffers: one device copy / one fused lerp kernel."""
from copy import deepcopy

import torch
from torch import nn

from ddk import ops
from .optim import FlatParams


class EMA():
    def __init__(self, model: nn.Module, decay: float = 0.99):
        self.decay = decay
        self.ema_model = deepcopy(model)
        for p in self.ema_model.parameters():
            p.requires_grad_(False)
        self._flat = None

    def _flat_ema(self):
        if self._flat is None:
            self._flat = FlatParams(self.ema_model, with_grad=False)
        return self._flat

    @staticmethod
    def _live_flat(model):
        fp = getattr(model, "_flat_params", None)
        return fp.flat if fp is not None else torch.cat([p.detach().reshape(-1) for p in model.parameters()])

    def _touch(self):
        for m in self.ema_model.modules():
            if hasattr(m, "invalidate_plan"):
                m.invalidate_plan()

    def eval(self):
        self.ema_model.eval()

    def reset(self, model):
        """trainers/ema.py:33-34 (deepcopy of the live model) as a device-to-device parameter copy."""
        self._flat_ema().flat.copy_(self._live_flat(model))
        self._touch()

    def update(self, model):
        """trainers/ema.py:36-44"""
        ops.ema_update_(self._flat_ema().flat, self._live_flat(model), self.decay)
        self._touch()

    def forward(self, x):
        return self.ema_model(x)

    @torch.no_grad()
    def sample(self, n: int):
        return self.ema_model.sample(n)

    @torch.no_grad()
    def reconstruct(self, x, n: int):
        return self.ema_model.reconstruct(x, n)

    def load_state_dict(self, state_dict) -> None:
        self.ema_model.load_state_dict(state_dict)
        self._touch()

    def state_dict(self):
        return self.ema_model.state_dict()
