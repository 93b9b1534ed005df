"""Flat-buffer optimiser state for the HIP training path.

All parameters of the model are re-homed into ONE contiguous fp32 buffer (each nn.Parameter becomes a view of it) and
so are the gradients.  The global-norm clip, the Adam update and the EMA then are single fused kernels over the flat
buffers (ddk_grad_norm_clip / ddk_adam_step / ddk_ema_update), and the data-parallel gradient exchange is ONE
all-reduce of the flat gradient bucket.

Semantics follow reference trainers/trainer.py:69 (Adam(params, lr), torch defaults) and
trainers/trainer_ddpm.py:142-144 (clip_grad_norm_(params, 1.0) -> opt.step() -> opt.zero_grad()).
"""
import torch

from ddk import ops


class FlatParams:
    """Re-homes module parameters (and optionally their .grad) into flat buffers; keeps (name, offset, numel)."""

    def __init__(self, module, with_grad=True):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatParams: move the model to the ROCm device before building optimiser state")
        self.params = params
        self.names = [n for n, _ in module.named_parameters()]
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32) if with_grad else None
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                if with_grad:
                    p.grad = self.grad[off:off + n].view_as(p)
                self.offsets.append(off)
                off += n

    def views(self, flat):
        return [flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]


class FusedAdam:
    """torch.optim.Adam semantics (no amsgrad / weight decay) with the preceding clip_grad_norm_ folded in."""

    def __init__(self, module, lr, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0):
        self.fp = FlatParams(module, with_grad=True)
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.step_count = 0
        self.last_norm = None      # device tensor [grad_norm, clip_coef] of the latest step (no host sync)

    def zero_grad(self):
        self.fp.grad.zero_()

    def step(self):
        """global-norm clip (max_grad_norm) + Adam; gradients must already be reduced across ranks."""
        self.last_norm = ops.grad_norm_clip(self.fp.grad, self.max_grad_norm)
        self.step_count += 1
        ops.adam_step_(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.step_count, clip=self.last_norm,
                       betas=self.betas, eps=self.eps)
        ops.weights_changed()      # in-place update behind torch's version counters: drop cached kernel-layout copies
        return self.last_norm

    # ---- torch.optim.Adam-compatible (de)serialisation, so checkpoints interchange with the reference trainer
    def state_dict(self):
        m, v = self.fp.views(self.exp_avg), self.fp.views(self.exp_avg_sq)
        state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": m[i].clone(), "exp_avg_sq": v[i].clone()}
                 for i in range(len(self.fp.params))} if self.step_count > 0 else {}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                     capturable=False, differentiable=False, fused=None, params=list(range(len(self.fp.params))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the dict torch.optim.Adam.state_dict() writes for the same parameter list (the reference trainer's
        checkpoints, trainers/trainer_ddpm.py:51) as well as our own; parameter count and shapes are validated."""
        if len(sd.get("param_groups", [])) != 1:
            raise ValueError(f"FusedAdam: expected one param group, got {len(sd.get('param_groups', []))}")
        group = sd["param_groups"][0]
        if group.get("amsgrad") or group.get("weight_decay", 0) or group.get("maximize"):
            raise ValueError("FusedAdam: amsgrad / weight_decay / maximize are not supported (the reference uses Adam defaults)")
        if len(group["params"]) != len(self.fp.params):
            raise ValueError(f"FusedAdam: optimizer state is for {len(group['params'])} parameters, the model has {len(self.fp.params)}")
        state = sd.get("state", {})
        for i, st in state.items():
            p = self.fp.params[int(i)]
            for k in ("exp_avg", "exp_avg_sq"):
                if tuple(st[k].shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdam: state[{i}].{k} has shape {tuple(st[k].shape)}, parameter '{self.fp.names[int(i)]}' is "
                                     f"{tuple(p.shape)}")
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0
        if state:
            m, v = self.fp.views(self.exp_avg), self.fp.views(self.exp_avg_sq)
            with torch.no_grad():
                for i, st in state.items():
                    i = int(i)
                    m[i].copy_(st["exp_avg"])
                    v[i].copy_(st["exp_avg_sq"])
                    self.step_count = max(self.step_count, int(float(st["step"])))
