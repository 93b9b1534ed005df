"""Data-parallel plumbing for the DDPM path (the reference has none: SURVEY.md F8, section 2.1).

Sampling shards over the batch axis with NO collective in the step loop: rank 0 broadcasts the weights once as one
flat fp32 bucket (C1), every rank samples its own slice with its own Philox stream id, results stay per rank (or are
gathered once at the end).  Training all-reduces ONE flat gradient bucket per optimiser step (C2), before the
global-norm clip, so every rank applies the identical update.  xGMI is point-to-point (7 links per GPU): a single
large bucket lets RCCL use all links at once, which is why nothing here is bucketed per tensor.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:      # DDK_DIST_BACKEND=gloo: rehearse N ranks on one GPU (RCCL wants one device per rank)
            backend = os.environ.get("DDK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_sizes(total, world):
    """Contiguous, near-equal split of `total` items over `world` ranks (first ranks take the remainder)."""
    base, rem = divmod(total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def shard_batch(total, rank, world):
    """(start, stop) of this rank's slice of a global batch."""
    sizes = shard_sizes(total, world)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


def flatten_tensors(tensors):
    """One contiguous fp32 bucket holding every tensor (in order)."""
    return torch.cat([t.detach().reshape(-1).float() for t in tensors]) if tensors else torch.empty(0)


def unflatten_into_(flat, tensors):
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return tensors


def _active(force):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return force or dist.get_world_size() > 1


def broadcast_module_(module, src=0, force=False):
    """C1: rank `src`'s parameters + floating buffers to every rank as one flat bucket.  Returns bytes moved.
    force=True issues the collective even on a world of one rank (tests: walks the RCCL device-tensor path on a 1-GPU box)."""
    if not _active(force):
        return 0
    tensors = [t for t in module.state_dict().values() if torch.is_floating_point(t)]
    flat = flatten_tensors(tensors)
    dist.broadcast(flat, src=src)
    unflatten_into_(flat, tensors)
    return flat.numel() * 4


def all_reduce_flat_(flat, average=True, force=False):
    """C2: sum (then average) a flat gradient bucket over all ranks, in place."""
    if not _active(force):
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(dist.get_world_size())
    return flat


def is_main_rank():
    """True on rank 0 (and in single-process runs): the rank that writes checkpoints, losses and logs."""
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_ranks_agree(ok, detail=None):
    """Every rank reports whether a step it ran LOCALLY worked (`ok`, with an error text in `detail`); returns
    (all_ok, first_failure_text) -- the same pair on every rank.  For decisions that must not diverge between ranks (a rank
    that raises or changes its launch sequence alone leaves the others parked in the next collective)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok), (None if ok else detail)
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, None if ok else f"rank {dist.get_rank()}: {detail}")
    bad = [b for b in box if b is not None]
    return not bad, (bad[0] if bad else None)


def main_rank_does(fn, what="rank-0 work"):
    """Run `fn()` on rank 0 only and make EVERY rank learn whether it worked: rank 0's success / error text is broadcast
    afterwards, and all ranks raise together when it failed.  A bare barrier behind rank-0-only file I/O hangs the other
    ranks forever if rank 0 raises (disk full, permissions) before reaching its own barrier.  Returns fn()'s value on rank 0,
    None elsewhere."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    result, err = None, None
    if is_main_rank():
        try:
            result = fn()
        except Exception as e:      # noqa: BLE001 -- reported to every rank below, then re-raised
            if not multi:
                raise
            err = f"{type(e).__name__}: {e}"
    if multi:
        box = [err]
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            raise RuntimeError(f"{what} failed on rank 0: {box[0]}")
    return result
