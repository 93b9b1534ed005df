"""Child process of tests/test_dp_gpu.py: ONE data-parallel rank of the product training path.

Started fresh (before any GPU call) by the test, `gloo` backend with every rank on cuda:0 -- the rehearsal of the one-process-
per-GPU RCCL layout that fits a 1-GPU box.  Runs trainers.setup_trainer (model build + rank-0 weight broadcast) and one
optimiser step of TrainerDDPM on this rank's half of a fixed global batch, then writes what the parent asserts on.
Reference semantics: trainers/trainer_ddpm.py:118-144 (2 micro-batches, obj/2 backward, clip, Adam, EMA)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, HERE]


def config():
    return dict(model="ddpm", dataset="cifar10", batch_size=4, image_size=16, n_steps=1, lr=2e-4, unet_chan=32, unet_dims=(1, 2, 2, 2),
                unet_dropout=0.0, T=1000, loss_type="simple", beta_schedule="linear", ema_decay=0.995, loss_flat="sum", val_split=0,
                n_downsamples=0, n_samples=4, graph_train=False)


def fixed_inputs(mb):
    """The GLOBAL micro-batch `mb` (4 samples): images, timesteps and noise every process can regenerate."""
    import torch
    from utils import synthetic as syn
    x = syn.synthetic_input((4, 3, 16, 16), f"dp.x{mb}")
    t = torch.tensor([3, 250 + mb, 700, 999 - mb])
    eps = syn.synthetic_normal((4, 3, 16, 16), f"dp.eps{mb}")
    return x, t, eps


def run_micro_batch(trainer, x, t, eps):
    """trainer._micro_batch with the model's random draws (t_sample, randn_like) replaced by the fixed ones."""
    import torch
    dev = trainer.device
    x, t, eps = x.to(dev), t.to(dev), eps.to(dev)
    trainer.model.t_sample = lambda n, t=t: t
    orig = torch.randn_like
    torch.randn_like = lambda z, eps=eps: eps
    try:
        obj, _ = trainer._micro_batch(x)
    finally:
        torch.randn_like = orig
    return obj


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["LOCAL_RANK"] = "0"                      # every rank on cuda:0
    import torch
    import torch.distributed as dist
    from parallel import init_from_env
    init_from_env("gloo")
    import utils
    utils.LOGGING_DIR = out_dir
    import trainers.trainer as tr_mod
    import trainers.trainer_ddpm as td_mod
    tr_mod.LOGGING_DIR = td_mod.LOGGING_DIR = out_dir
    torch.manual_seed(1234 + rank)                      # different initial weights per rank: the broadcast has work to do
    from trainers import setup_trainer
    trainer, cfg = setup_trainer(config(), True, None, "dp_test", seed=None)
    res = {"rank": rank}
    res["state_after_broadcast"] = {k: v.detach().cpu().clone() for k, v in trainer.model.state_dict().items()}

    trainer.model.train()
    trainer.opt.zero_grad()
    lo, hi = 2 * rank, 2 * rank + 2                      # this rank's half of every global micro-batch
    for mb in range(trainer.gradient_accumulate_every):
        x, t, eps = fixed_inputs(mb)
        run_micro_batch(trainer, x[lo:hi], t[lo:hi], eps[lo:hi])
    res["local_grad"] = trainer.opt.fp.grad.detach().cpu().clone()
    snap = {}
    orig_step = trainer.opt.step

    def step_and_snapshot():
        snap["reduced_grad"] = trainer.opt.fp.grad.detach().cpu().clone()    # after the all-reduce, before clip + Adam
        return orig_step()
    trainer.opt.step = step_and_snapshot
    trainer.optimizer_step()                             # product path: all-reduce -> clip -> Adam -> zero_grad
    trainer.update_ema()
    res["reduced_grad"] = snap["reduced_grad"]
    res["flat_after_step"] = trainer.opt.fp.flat.detach().cpu().clone()
    res["exp_avg"] = trainer.opt.exp_avg.detach().cpu().clone()
    res["ema_state"] = {k: v.detach().cpu().clone() for k, v in trainer.ema.state_dict().items()}
    # rank-0-only checkpoint
    trainer.init_wandb()
    trainer.save_checkpoint()
    res["checkpoint_name"] = trainer.checkpoint_name
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
