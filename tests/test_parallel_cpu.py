"""world_size-2 gloo tests of the multi-GPU plumbing (runs on CPU): weight broadcast bucket, gradient all-reduce bucket,
batch sharding.  The sampler's shard equivalence on real kernels is in tests/test_sampler_gpu.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "downsampled-diffusion_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from parallel import all_reduce_flat_, broadcast_module_, flatten_tensors, init_from_env, shard_batch, unflatten_into_
    r, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # different weights per rank before the broadcast
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.GroupNorm(2, 8), torch.nn.Linear(4, 4))
    model.register_buffer("sched", torch.arange(5, dtype=torch.float32) * (rank + 1))
    nbytes = broadcast_module_(model, src=0)
    flat = flatten_tensors(list(model.state_dict().values()))
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    # gradient bucket: mean over ranks
    grads = [torch.full_like(p, float(rank + 1)) for p in model.parameters()]
    gflat = flatten_tensors(grads)
    all_reduce_flat_(gflat)
    unflatten_into_(gflat, grads)
    ok_grad = all(torch.allclose(g, torch.full_like(g, (1 + world) / 2)) for g in grads)
    q.put((rank, same, nbytes, ok_grad, shard_batch(33, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_allreduce_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_float = (3 * 8 * 9 + 8) + 16 + (16 + 4) + 5
    for rank, same, nbytes, ok_grad, shard in res:
        assert same and ok_grad and nbytes == n_float * 4
    assert res[0][4] == (0, 17) and res[1][4] == (17, 33)


def test_shard_sizes_cover_everything():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "downsampled-diffusion_amd")]
    from parallel import shard_batch, shard_sizes
    for total in (0, 1, 7, 32, 50000):
        for world in (1, 2, 3, 8):
            assert sum(shard_sizes(total, world)) == total
            spans = [shard_batch(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_main_rank_does(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "downsampled-diffusion_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from parallel import init_from_env, main_rank_does
    init_from_env("gloo")
    ok = main_rank_does(lambda: "written", "a write that works")
    try:
        main_rank_does(lambda: (_ for _ in ()).throw(OSError("disk full")), "save_checkpoint")
        raised = None
    except RuntimeError as e:
        raised = str(e)
    q.put((rank, ok, raised))
    dist.barrier()          # every rank is still in step after the failure
    dist.destroy_process_group()


def test_rank0_failure_raises_on_every_rank_instead_of_hanging():
    """parallel.main_rank_does: a rank-0-only write that fails must raise on all ranks (a bare barrier behind it would park the
    others forever) -- the pattern behind TrainerDDPM.save_checkpoint and the sampler's shard merge"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_main_rank_does, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == "written" and res[1][1] is None
    for _, _, raised in res:
        assert raised and "save_checkpoint failed on rank 0" in raised and "disk full" in raised


def _worker_all_ranks_agree(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "downsampled-diffusion_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from parallel import all_ranks_agree, init_from_env
    init_from_env("gloo")
    both = all_ranks_agree(True)
    one = all_ranks_agree(rank != 1, "HIP error: capture failed")       # only rank 1 failed locally
    q.put((rank, both, one))
    dist.barrier()
    dist.destroy_process_group()


def test_a_local_failure_is_known_to_every_rank():
    """parallel.all_ranks_agree: the outcome of a rank-local step (the trainer's device-graph capture) is the same on every rank,
    so either all ranks replay the graph or all raise / run eagerly -- nobody is left alone in the gradient all-reduce"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_all_ranks_agree, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, both, one in res:
        assert both == (True, None)
        assert one[0] is False and one[1] == "rank 1: HIP error: capture failed"
