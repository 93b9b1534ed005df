"""Sampler output stage (reference utils/eval_helpers.py:37-41, generate_model_samples.py:48-69)."""
import os

import numpy as np
import torch

from .utils import min_max_norm_image


class OutputStage:
    """Device -> host pipeline of the sampling driver: fused fix_samples kernel, then an ASYNCHRONOUS copy into one of two
    pinned staging buffers guarded by an event -- the host only waits for a buffer when it is about to be reused (two batches
    later) or at finish(), so the copy and the pageable memcpy of batch i overlap the sampling of batch i+1.
    submit() returns nothing; finish() returns the list of NHWC float32 arrays in submission order (the reference's
    ``sample_list``)."""

    def __init__(self):
        self._slots = {}          # (shape, device) -> [[pinned, event, pending], [pinned, event, pending]]
        self._turn = {}
        self._out = []            # results in submission order (None while in flight)

    def _retire(self, slot):
        if slot[2] is not None:
            slot[1].synchronize()
            self._out[slot[2]] = slot[0].numpy().copy()
            slot[2] = None

    def submit(self, samples):
        from ddk import ops
        if not samples.is_cuda:
            self._out.append(np.moveaxis((min_max_norm_image(samples) * 255.).numpy(), 1, -1))
            return
        dev = ops.fix_samples(samples.contiguous().float())
        key = (tuple(dev.shape), str(dev.device))
        if key not in self._slots:
            self._slots[key] = [[torch.empty(dev.shape, dtype=torch.float32, pin_memory=True), torch.cuda.Event(), None] for _ in range(2)]
            self._turn[key] = 0
        slot = self._slots[key][self._turn[key]]
        self._turn[key] ^= 1
        self._retire(slot)                           # waits only if this buffer's previous copy has not landed yet
        slot[0].copy_(dev, non_blocking=True)
        slot[1].record(torch.cuda.current_stream(samples.device))
        slot[2] = len(self._out)
        self._out.append(None)                       # (`dev` may be freed now: the allocator re-uses it in stream order, after the copy)

    def finish(self):
        for slots in self._slots.values():
            for slot in slots:
                self._retire(slot)
        out, self._out = self._out, []
        return out


def fix_samples(samples):
    """Per-image min-max -> [0,255] -> host NHWC float32: the on-disk format of generate_model_samples.py (reference
    utils/eval_helpers.py:37-41).  Synchronous, reference-shaped call: one batch through the OutputStage."""
    stage = OutputStage()
    stage.submit(samples)
    return stage.finish()[0]


def merge_rank_shards(base_path, world, remove=False):
    """Batch-sharded sampling writes ``{base}.rank{r}.npy`` per rank; the evaluator (reference evaluate_ddpm.py:52) loads
    ONE ``{base}.npy``.  Concatenates the shards in rank order (= global batch order: ranks take contiguous runs of the
    job's batches) into that file.  Returns the merged array."""
    parts, paths = [], []
    for r in range(world):
        path = f"{base_path}.rank{r}.npy"
        if not os.path.exists(path):
            raise FileNotFoundError(f"missing sampling shard {path} (the ranks must write to a filesystem rank 0 can read)")
        paths.append(path)
        a = np.load(path)
        if a.size:
            parts.append(a)
    merged = np.concatenate(parts, axis=0) if parts else np.zeros((0,), dtype=np.float32)
    np.save(base_path, merged, allow_pickle=False)
    if remove:
        for path in paths:          # exactly the shards that were merged: no glob (metacharacters in the name, stale shards of
            os.remove(path)         # an earlier, larger world are not ours to delete unseen)
    return merged
