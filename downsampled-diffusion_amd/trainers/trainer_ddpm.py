"""DDPM / dDDPM training loops with the reference's semantics (reference trainers/trainer_ddpm.py:13-265):
2 accumulation micro-batches, (obj / 2).backward(), clip_grad_norm_(1.0), Adam step, zero_grad, EMA (copy while
step < 2000, lerp every 10th step after), eval toggles, checkpoint dict schema.

MI355X specifics: forward + backward are HIP kernels (trainers/autograd_unet.py); clip + Adam + EMA are three fused
kernels over flat buffers; under torchrun the flat gradient bucket is all-reduced ONCE per optimiser step over RCCL,
before the clip, so every rank applies the identical update; losses are read back once per step, not per micro-batch.
"""
import os

import numpy as np
import torch

from parallel import all_ranks_agree, all_reduce_flat_, is_main_rank, main_rank_does
from utils import LOGGING_DIR, min_max_norm_image
from .ema import EMA
from .trainer import Trainer
from .train_helpers import cycle


def _is_oom(text):
    t = (text or "").lower()
    return "out of memory" in t or "outofmemory" in t or "hiperroroutofmemory" in t


def _invalidate(model):
    for m in model.modules():
        if hasattr(m, "invalidate_plan"):
            m.invalidate_plan()


class TrainerDDPM(Trainer):
    def __init__(self, config: dict, model, train_loader, val_loader=None, device: str = 'cpu', wandb_name: str = 'tmp',
                 mute: bool = True, n_channels: int = None):
        # the EMA copy must be taken before the parameters are re-homed into the optimiser's flat buffer
        self.use_ema = config['ema_decay'] > 0
        ema = EMA(model.to(device), config['ema_decay']) if self.use_ema else None
        super().__init__(config, model, train_loader, val_loader, device, wandb_name, mute, n_channels)
        self.train_loader = cycle(self.train_loader)
        if config['val_split'] > 0:
            self.val_loader = cycle(self.val_loader)
            self.val_batch = next(self.val_loader)[0][0].repeat(self.n_samples, 1, 1, 1).to(self.device)
        else:
            self.val_batch = next(self.train_loader)[0][0].repeat(self.n_samples, 1, 1, 1).to(self.device)
        self.step = 0
        self._graph = None
        self._merge = None              # merged micro-batches: None = not decided yet, then True / False (see _accumulate)
        self._merge_proven = False
        self.gradient_accumulate_every = 2
        self.logging_every = 10000
        if self.use_ema:
            self.step_start_ema = 2000
            self.update_ema_every = 10
            self.ema = ema
            self.ema.eval()
        self.name += f'_{config["T"]}'

    # ------------------------------------------------------------------ checkpoints (trainer_ddpm.py:49-72)
    def save_checkpoint(self) -> None:
        """Rank 0 writes the file (every rank holds the identical replicated state after the all-reduced step); the others
        wait for its verdict (parallel.main_rank_does: success or the error text is broadcast, so a failed write raises on
        every rank instead of leaving them parked at a barrier) -- nobody races ahead into a resume / the next save."""
        def write():
            save_data = {
                'optimizer': self.opt.state_dict(),
                'model': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'config': self.config,
                'train_losses': self.train_losses,
                'step': self.step,
            }
            if self.use_ema:
                save_data['ema_model'] = self.ema.state_dict()
            tmp = self.checkpoint_name + '.tmp'
            torch.save(save_data, tmp)
            os.replace(tmp, self.checkpoint_name)          # never leave a half-written checkpoint behind
            if self.logger:
                self.logger.save(self.checkpoint_name, policy='live')
        main_rank_does(write, "save_checkpoint")

    def load_checkpoint(self, checkpoint: dict) -> None:
        self.opt.load_state_dict(checkpoint['optimizer'])
        self.model.load_state_dict(checkpoint['model'])
        _invalidate(self.model)
        self.config = checkpoint['config']
        self.train_losses = checkpoint['train_losses']
        self.step = checkpoint['step']
        if 'ema_model' in checkpoint and self.use_ema:
            self.ema.load_state_dict(checkpoint['ema_model'])

    @torch.no_grad()
    def sample(self):
        return self.ema.sample(self.n_samples) if self.use_ema else self.model.sample(self.n_samples)

    @torch.no_grad()
    def recon(self, x):
        return self.ema.reconstruct(x, self.n_samples) if self.use_ema else self.model.reconstruct(x, self.n_samples)

    @torch.no_grad()
    def log_wandb(self, x, commit: bool = True) -> None:
        """trainer_ddpm.py:91-105: sample + reconstruction grids, stored as .npy (no torchvision / wandb offline)."""
        if not is_main_rank():       # image logs are rank 0's job; the other ranks skip the T-step sampling too
            return
        samples, recon = self.sample(), self.recon(x)
        samples = samples[0] if isinstance(samples, tuple) else samples
        recon = recon[0] if isinstance(recon, tuple) else recon
        log_name = f'{self.step}_{self.name}_{self.config["dataset"]}'
        os.makedirs(LOGGING_DIR, exist_ok=True)
        np.save(os.path.join(LOGGING_DIR, f'sample_{log_name}.npy'), min_max_norm_image(samples).cpu().numpy())
        np.save(os.path.join(LOGGING_DIR, f'recon_{log_name}.npy'), min_max_norm_image(recon).cpu().numpy())

    def update_ema(self):
        """trainer_ddpm.py:107-111"""
        if self.step < self.step_start_ema:
            self.ema.reset(self.model)
        elif self.step % self.update_ema_every == 0:
            self.ema.update(self.model)

    @property
    def merge_micro_batches(self):
        """True while the micro-batches of a step run as one merged pass (decided at the first step; see _accumulate)"""
        if self._merge is None:
            return bool(self.config.get('merge_micro_batches', True)) and self.gradient_accumulate_every > 1
        return bool(self._merge)

    # ------------------------------------------------------------------ one optimiser step
    def _micro_batch(self, x=None, passes=None):
        """One forward + backward; `passes` = how many such passes make up the optimiser step (the objective enters the gradient
        bucket as obj / passes; default: gradient_accumulate_every)."""
        if x is None:
            x, _ = next(self.train_loader)
            x = x.to(self.device, non_blocking=True)
        passes = self.gradient_accumulate_every if passes is None else passes
        from ddk import ops
        with ops.deferred_wgrad():                 # the slab reduces of this backward pass: one launch when the block ends
            out = self.model(x)
            obj, extra = (out[0], out[1]) if isinstance(out, tuple) else (out, None)
            obj.backward(torch.full_like(obj, 1.0 / passes))    # = (obj / passes).backward(), two launches less
        return obj.detach(), extra

    def _accumulate(self):
        """The `gradient_accumulate_every` forward/backward passes of one step -> tensor [accumulate, 1 or 3] of
        (objective[, latent, recon]) per micro-batch.  On the GPU the passes are captured once into a device graph
        (trainers/graph_step.py) and replayed; config['graph_train'] = False keeps the eager sequence."""
        acc = self.gradient_accumulate_every
        batches = [next(self.train_loader)[0].to(self.device, non_blocking=True) for _ in range(acc)]
        # The objective is a MEAN of per-sample terms (ddpm.py:290-315, dddpm.py:152-177: no batch statistics anywhere -- GroupNorm,
        # LayerNorm and the attention are per sample), so sum_k d(obj_k / acc) over the `acc` micro-batches IS d(obj of their
        # concatenation): the micro-batches of a step go through ONE forward + backward as one batch of acc x batch_size samples --
        # the same gradient up to fp32 summation order (tests/test_trainer_gpu.py), half the launches, kernels twice as full.
        # config['merge_micro_batches'] = False keeps the reference's pass-by-pass sequence (trainer_ddpm.py:118-128).
        #
        # Memory: a merged pass holds the activations of acc x batch_size samples at once -- what accumulation exists to avoid
        # (cfg4 -bs 32: 30.3 GiB instead of 15.2).  So the merge is an optimisation that must be able to fail: when the first merged
        # pass (graph capture, or the first eager pass) runs out of device memory on ANY rank, every rank drops to the pass-by-pass
        # sequence for the rest of the run.  config['merge_micro_batches']: True (default) = merge with that fall-back, False = never.
        micro = batches
        if self._merge is None:
            self._merge = bool(self.config.get('merge_micro_batches', True)) and acc > 1
        merged = self._merge and all(b.shape == micro[0].shape for b in micro)
        if merged:
            batches = [torch.cat(micro)]
        use_graph = self.config.get('graph_train', True) and str(self.device).startswith('cuda')

        def capture(bs):
            from .graph_step import GraphedAccumulation
            try:
                return GraphedAccumulation(self.model, len(bs)).capture(bs), None
            except Exception as e:       # noqa: BLE001 -- e.g. a model whose forward synchronises with the host
                return None, f"{type(e).__name__}: {e}"

        def unmerge(why):
            print(f"[trainer] the merged pass over {acc} x {micro[0].shape[0]} samples ran out of device memory ({why}); "
                  f"this run accumulates pass by pass (config['merge_micro_batches'] = False says so up front)")
            self._merge = False
            self.opt.zero_grad()
            if str(self.device).startswith('cuda'):
                torch.cuda.synchronize()
                torch.cuda.empty_cache()

        if use_graph and self._graph is None:
            graph, err = capture(batches)
            # The outcome is agreed on by ALL ranks before anyone acts on it: a rank that raised (or went eager) alone would
            # leave the others parked in all_reduce_flat_ -- either every rank replays the graph, or every rank raises /
            # runs eagerly / drops the merge.
            ok, first_err = all_ranks_agree(err is None, err)
            if not ok and merged and _is_oom(first_err):
                del graph
                unmerge(first_err)
                merged, batches = False, micro
                graph, err = capture(batches)
                ok, first_err = all_ranks_agree(err is None, err)
            if ok:
                self._graph = graph
            else:
                # A silent fall-back to ~3000 eager launches per step is a performance regression nobody would notice:
                # only config['graph_train'] == 'auto' may degrade; the default (True) treats a failed capture as an error.
                if self.config.get('graph_train', True) != 'auto':
                    raise RuntimeError(f"device-graph capture of the training step failed ({first_err}); set "
                                       "config['graph_train'] = 'auto' to fall back to eager launches, or False to disable "
                                       "graph replay")
                print(f"[trainer] device-graph capture of the training step failed ({first_err}); running eagerly")
                self._graph = False
                torch.cuda.synchronize()
            self.opt.zero_grad()          # the warm-up / capture passes accumulated gradients of their own
        if (use_graph and self._graph and len(batches) == len(self._graph.static_x)
                and all(b.shape == s.shape for b, s in zip(batches, self._graph.static_x))):
            rows = self._graph.replay(batches)
        else:
            def eager(bs):
                rows = []
                for x in bs:
                    obj, extra = self._micro_batch(x, passes=len(bs))
                    rec = [obj] if extra is None else [obj, extra['latent'].detach(), extra['recon'].detach()]
                    rows.append(torch.stack([r.reshape(()) for r in rec]))
                return torch.stack(rows)
            if merged and not self._merge_proven:
                # the first eager merged pass of the run: its outcome (did it fit?) is agreed on by all ranks, as above
                rows, err = None, None
                try:
                    rows = eager(batches)
                except Exception as e:   # noqa: BLE001
                    err = f"{type(e).__name__}: {e}"
                ok, first_err = all_ranks_agree(err is None, err)
                if ok:
                    self._merge_proven = True
                elif _is_oom(first_err):
                    del rows
                    unmerge(first_err)
                    merged, batches = False, micro
                    rows = eager(batches)
                else:
                    raise RuntimeError(f"the training pass failed: {first_err}")
            else:
                rows = eager(batches)
        # one row per micro-batch as the loggers expect; a merged pass reports the mean over all its samples in each
        return rows.expand(acc, -1) if merged else rows

    def optimizer_step(self):
        """all-reduce (data parallel) -> clip_grad_norm_(1.0) -> Adam -> zero_grad (trainer_ddpm.py:142-144)"""
        all_reduce_flat_(self.opt.fp.grad, average=True)
        norm = self.opt.step()
        self.opt.zero_grad()
        _invalidate(self.model)
        return norm

    def train_loop(self) -> None:
        while self.step < self.n_steps:
            self.model.train()
            train_obj = float(self._accumulate()[:, 0].mean())   # one device->host read per step
            self.train_losses.append(train_obj)
            is_log = self.step != 0 and self.step % self.logging_every == 0
            self.logger.log({'train_obj': train_obj}, commit=(not is_log))
            self.optimizer_step()
            if self.use_ema:
                self.update_ema()
            self.model.eval()
            if is_log:
                self.save_checkpoint()
                self.log_wandb(self.val_batch)
            self.step += 1
        return self.train_losses


class TrainerDownsampleDDPM(TrainerDDPM):
    """trainer_ddpm.py:161-265: same loop, the model returns (objective, {'latent', 'recon'})."""

    def train_loop(self):
        while self.step < self.n_steps:
            self.model.train()
            rows = self._accumulate()                                    # [accumulate, (objective, latent, recon)]
            # the reference logs objective.item() / accumulate per micro-batch and the means of latent / recon
            vals = torch.stack([rows[:, 0].mean() / self.gradient_accumulate_every, rows[:, 1].mean(), rows[:, 2].mean()]).tolist()
            self.train_losses.append(vals[0])
            is_log = self.step != 0 and self.step % self.logging_every == 0
            self.logger.log({'train_obj': vals[0], 'train_latent': vals[1], 'train_recon': vals[2]}, commit=(not is_log))
            self.optimizer_step()
            if self.use_ema:
                self.update_ema()
            self.model.eval()
            if is_log:
                self.save_checkpoint()
                self.log_wandb(self.val_batch)
            self.step += 1
        return self.train_losses
