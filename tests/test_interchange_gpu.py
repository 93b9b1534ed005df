"""Checkpoint interchange with the reference trainer and the sampler output stage (N2), on the GPU.

N3: torch.optim.Adam <-> FusedAdam state dicts in both directions (reference trainers/trainer.py:69,
trainers/trainer_ddpm.py:49-72), the EMA preference of the sampling CLI (utils/utils.py:51-54), resume = bit-identical next step
(train_from_checkpoint.py:11-24).
N2: asynchronous double-buffered output stage == the synchronous fix_samples; batch-sharded CLI run merged into the single
.npy == the 1-process run, bit for bit (generate_model_samples.py:48-69, evaluate_ddpm.py:52)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import ddpm_cfg, dddpm_cfg, det_load
from utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    from models import DDPM, Unet
    cfg = ddpm_cfg(32, 3, 16)
    return det_load(DDPM(cfg, Unet(cfg), DEV, 3)), cfg


def _grads(model, k):
    return [syn.synthetic_normal(tuple(p.shape), f"ic.g{k}.{n}") * 0.05 for n, p in model.named_parameters()]


def test_torch_adam_state_loads_into_fused_adam_and_continues_identically():
    """2 steps of torch.optim.Adam on the CPU, its state_dict into FusedAdam, then step 3 on both with the same gradient:
    parameters agree to fp32 rounding of one lr-sized update."""
    from trainers.optim import FusedAdam
    ref, _ = _model()
    ref = ref.cpu()
    opt_t = torch.optim.Adam(ref.parameters(), lr=2e-4)
    for k in range(2):
        for p, g in zip(ref.parameters(), _grads(ref, k)):
            p.grad = g.clone()
        opt_t.step()
    ours, _ = _model()
    ours.load_state_dict(ref.state_dict())
    ours = ours.to(DEV)
    opt_f = FusedAdam(ours, lr=1.0, max_grad_norm=1e30)          # lr comes from the loaded state; clip disabled
    opt_f.load_state_dict(opt_t.state_dict())
    assert opt_f.step_count == 2 and opt_f.lr == 2e-4
    for i, p in enumerate(ref.parameters()):
        assert torch.equal(opt_f.fp.views(opt_f.exp_avg)[i].cpu(), opt_t.state[p]["exp_avg"])
        assert torch.equal(opt_f.fp.views(opt_f.exp_avg_sq)[i].cpu(), opt_t.state[p]["exp_avg_sq"])
    g3 = _grads(ref, 2)
    for p, g in zip(ref.parameters(), g3):
        p.grad = g.clone()
    opt_t.step()
    for v, g in zip(opt_f.fp.views(opt_f.fp.grad), g3):
        v.copy_(g.to(DEV))
    opt_f.step()
    for (n, a), b in zip(ref.named_parameters(), ours.parameters()):
        assert float((a.detach() - b.detach().cpu()).abs().max()) < 2e-4 * 1e-3, n        # 0.1 % of one lr step

    # and back: FusedAdam.state_dict() is a valid torch.optim.Adam state for the same parameter list
    back = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in ours.parameters()], lr=1.0)
    back.load_state_dict(opt_f.state_dict())
    assert back.param_groups[0]["lr"] == 2e-4
    for i, p in enumerate(back.param_groups[0]["params"]):
        assert float(back.state[p]["step"]) == 3.0
        assert torch.equal(back.state[p]["exp_avg"].cpu(), opt_f.fp.views(opt_f.exp_avg)[i].cpu())


def test_fused_adam_rejects_foreign_state():
    from trainers.optim import FusedAdam
    m, _ = _model()
    opt = FusedAdam(m.to(DEV), lr=2e-4)
    other = torch.nn.Linear(4, 4)
    o2 = torch.optim.Adam(other.parameters(), lr=1e-3)
    other(torch.ones(1, 4)).sum().backward()
    o2.step()
    with pytest.raises(ValueError, match="parameters"):
        opt.load_state_dict(o2.state_dict())
    sd = opt.state_dict()
    opt.step()
    sd = opt.state_dict()
    sd["state"][0]["exp_avg"] = torch.zeros(3, 3)
    with pytest.raises(ValueError, match="shape"):
        opt.load_state_dict(sd)


def test_resume_from_reference_shaped_checkpoint_is_bit_identical(tmp_path, monkeypatch):
    """Trainer A: 2 optimiser steps, checkpoint through torch.save with numpy losses (as the reference writes them); trainer B:
    load_checkpoint_file + load_checkpoint; step 3 on the same micro-batches -> identical flat parameters, moments, EMA."""
    import dp_worker as W
    import trainers.trainer as T
    import trainers.trainer_ddpm as TD
    from trainers import setup_trainer
    from utils import load_checkpoint_file
    for mod in (T, TD):
        monkeypatch.setattr(mod, "LOGGING_DIR", str(tmp_path) + "/", raising=True)

    def step(trainer, k):
        trainer.model.train()
        trainer.opt.zero_grad()
        for mb in range(trainer.gradient_accumulate_every):
            x, t, eps = W.fixed_inputs(mb)
            W.run_micro_batch(trainer, x + 0.01 * k, t, eps)
        trainer.optimizer_step()
        trainer.update_ema()
        trainer.step += 1

    a, _ = setup_trainer(W.config(), True, None, "resume_a", seed=0)
    a.init_wandb()
    for k in range(2):
        step(a, k)
    a.train_losses = [np.mean([1.0, 2.0]), np.float64(0.25)]          # numpy scalars, like np.mean in the reference loop
    a.save_checkpoint()
    ck = load_checkpoint_file(a.checkpoint_name)
    assert isinstance(ck["train_losses"][0], np.floating) and ck["step"] == 2
    b, _ = setup_trainer(dict(ck["config"]), True, None, "resume_b", seed=123)        # different init: everything must come from the file
    b.load_checkpoint(ck)
    assert b.step == 2 and b.opt.step_count == 2
    step(a, 2)
    step(b, 2)
    assert torch.equal(a.opt.fp.flat, b.opt.fp.flat)
    assert torch.equal(a.opt.exp_avg, b.opt.exp_avg) and torch.equal(a.opt.exp_avg_sq, b.opt.exp_avg_sq)
    for (k, v), (_, w) in zip(a.ema.state_dict().items(), b.ema.state_dict().items()):
        assert torch.equal(v, w), k


def test_output_stage_matches_fix_samples():
    from utils import OutputStage, fix_samples
    batches = [syn.synthetic_normal((3, 3, 16, 16), f"os.{k}").to(DEV) for k in range(5)]
    stage = OutputStage()
    for b in batches:
        stage.submit(b)
    got = stage.finish()
    assert len(got) == 5 and stage.finish() == []
    for g, b in zip(got, batches):
        want = fix_samples(b)
        assert g.shape == (3, 16, 16, 3) and g.dtype == np.float32
        assert np.array_equal(g, want)
        ref = (b - b.reshape(3, -1).min(dim=1).values[:, None, None, None]) / \
              (b.reshape(3, -1).max(dim=1).values - b.reshape(3, -1).min(dim=1).values)[:, None, None, None] * 255.
        assert np.array_equal(g, np.moveaxis(ref.cpu().numpy(), 1, -1))          # the reference expression, bit for bit


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_sampling_merges_to_the_single_process_file(tmp_path):
    """generate_model_samples.py: 1 process vs 2 ranks (gloo, both on cuda:0): the merged {saved_model}.npy -- the ONE file
    evaluate_ddpm.py:52 loads -- is bit-identical, because a batch's draws depend only on its global index."""
    cfg = dddpm_cfg(32, 32, 2)
    cfg.update(model="dddpm", dataset="celeba", T=100)
    cfg_path = tmp_path / "cfg.json"
    cfg_path.write_text(json.dumps(cfg))
    script = os.path.join(ROOT, "downsampled-diffusion_amd", "generate_model_samples.py")
    base = [sys.executable, script, "--synthetic", str(cfg_path), "--saved_model", "shardtest", "--fid_samples", "10", "--batch_size", "2",
            "--early_stop", "92"]
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "downsampled-diffusion_amd"))
    single, multi = tmp_path / "single", tmp_path / "multi"
    r = subprocess.run(base + ["--out_dir", str(single)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    port = _free_port()
    procs = [subprocess.Popen(base + ["--out_dir", str(multi)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(env, RANK=str(rk), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                       DDK_DIST_BACKEND="gloo")) for rk in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2500:]
    a, b = np.load(single / "shardtest.npy"), np.load(multi / "shardtest.npy")
    assert a.shape == (5, 2, 32, 32, 3) and np.array_equal(a, b)
    assert np.array_equal(np.load(single / "shardtest_latent.npy"), np.load(multi / "shardtest_latent.npy"))
    assert not list(multi.glob("*.rank*.npy"))                  # shards removed after the merge
