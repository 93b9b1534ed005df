"""Data-parallel training on the product path: two fresh child processes (tests/dp_worker.py),
gloo backend, both ranks on cuda:0, run setup_trainer + one optimiser step on half of a fixed global batch each.

Asserted: (a) the post-broadcast state_dict is bitwise equal on every rank although the ranks initialised differently;
(b) the all-reduced flat gradient equals the single-process gradient of the full batch within 3e-5 (rel. to max);
(c) the flat parameters, Adam moments and the EMA are bitwise equal across ranks after the step;
(d) only rank 0 wrote the checkpoint, and it holds the stepped weights.
Reference loop being parallelised: trainers/trainer_ddpm.py:118-144."""
import glob
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_dp_two_ranks_match_single_process(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "dp_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world))

    # (a) broadcast: bitwise equal state on both ranks
    assert r0["state_after_broadcast"].keys() == r1["state_after_broadcast"].keys()
    for k, v in r0["state_after_broadcast"].items():
        assert torch.equal(v, r1["state_after_broadcast"][k]), k
    # the ranks really did compute different local gradients, and agree after the all-reduce
    assert not torch.equal(r0["local_grad"], r1["local_grad"])
    assert torch.equal(r0["reduced_grad"], r1["reduced_grad"])

    # (b) single-process reference: the same trainer on the FULL global micro-batches, starting from the broadcast weights
    import dp_worker as W
    import utils
    import trainers.trainer as tr_mod
    import trainers.trainer_ddpm as td_mod
    from trainers import setup_trainer
    utils.LOGGING_DIR = tr_mod.LOGGING_DIR = td_mod.LOGGING_DIR = str(tmp_path)
    trainer, _ = setup_trainer(W.config(), True, None, "dp_ref", seed=0)
    trainer.model.load_state_dict(r0["state_after_broadcast"])
    if trainer.use_ema:
        trainer.ema.reset(trainer.model)
    trainer.model.train()
    trainer.opt.zero_grad()
    for mb in range(trainer.gradient_accumulate_every):
        W.run_micro_batch(trainer, *W.fixed_inputs(mb))
    full = trainer.opt.fp.grad.detach().cpu()
    err = float((r0["reduced_grad"] - full).abs().max() / full.abs().max())
    assert err < 3e-5, err
    assert float((0.5 * (r0["local_grad"] + r1["local_grad"]) - r0["reduced_grad"]).abs().max()) <= 1e-6 * float(full.abs().max())

    # (c) replicas stay bit-identical through clip + Adam + EMA
    assert torch.equal(r0["flat_after_step"], r1["flat_after_step"])
    assert torch.equal(r0["exp_avg"], r1["exp_avg"])
    for k, v in r0["ema_state"].items():
        assert torch.equal(v, r1["ema_state"][k]), k
    # ... and match the single-process step on the full batch to rounding
    trainer.optimizer_step()
    ref_flat = trainer.opt.fp.flat.detach().cpu()
    assert float((r0["flat_after_step"] - ref_flat).abs().max()) < 0.05 * 2e-4     # a fraction of one lr-sized Adam step

    # (d) one checkpoint file, written by rank 0, holding the stepped weights
    assert r0["checkpoint_name"] == r1["checkpoint_name"]
    files = glob.glob(os.path.join(tmp_path, "checkpoint_*.pt"))
    assert files == [r0["checkpoint_name"]], files
    ck = torch.load(files[0], weights_only=False)
    assert set(ck) >= {"optimizer", "model", "config", "train_losses", "step", "ema_model"}
    names = [n for n, _ in trainer.model.named_parameters()]
    flat_ck = torch.cat([ck["model"][n].reshape(-1).cpu() for n in names])
    assert torch.equal(flat_ck, r0["flat_after_step"])
