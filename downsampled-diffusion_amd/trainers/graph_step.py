"""The gradient-accumulation part of a training step -- `accumulate` x [forward, (obj / accumulate).backward()] -- captured
into ONE device graph (torch.cuda.CUDAGraph, i.e. a hipGraph) and replayed per optimiser step.

Why: the training forward/backward is ~900 HIP kernel launches sequenced by Python + torch.autograd (host-bound:
40-50 ms per step at cfg2/cfg3 for ~6 ms of GPU work).  A replay costs one launch.

What makes the capture valid here:
  * inputs are copied into static buffers; gradients accumulate into the optimiser's flat gradient buffer (static);
  * t ~ randint and eps ~ randn come from torch's graph-safe CUDA generator;
  * dropout masks are keyed by (host seed baked at capture) + (device epoch bumped by a kernel inside the graph);
  * the kernel-layout weight copies are marked stale before capture, so their refresh is part of the graph (weights change every
    step): ONE ddk_pack_jobs launch over the persistent buffers the warm-up passes registered (ddk/ops.py:cached_pack); a copy
    first requested while capturing lives in the graph's private pool and is dropped from the cache afterwards, so no eager call is
    ever handed a tensor whose contents only a replay keeps current;
  * scratch workspaces requested while capturing come from the graph's private pool (ddk/ops.py:_ws), not from the shared
    per-tag buffers, so a later larger eager request cannot free memory the graph replays into;
  * the optimiser (all-reduce, clip, Adam, EMA) stays outside: its scalars (bias corrections) change per step.
"""
import torch

from ddk import lib as L


def _invalidate(model):
    from ddk import ops
    ops.weights_changed()
    for m in model.modules():
        if hasattr(m, "invalidate_plan"):
            m.invalidate_plan()


class GraphedAccumulation:
    def __init__(self, model, accumulate: int):
        self.model, self.accumulate = model, accumulate
        self.graph = None
        self.static_x = None
        self.outputs = None
        self._keep = []        # buffers the captured launches address through device tables (ddk.ops.graph_owner)

    def _run(self):
        from ddk import ops
        lib = L.load()
        outs = []
        for x in self.static_x:
            L.check(lib.ddk_dropout_epoch(0, 1, L.stream()), "dropout_epoch")
            with ops.deferred_wgrad():             # the slab reduces of this backward pass: one launch when the block ends
                out = self.model(x)
                obj, extra = (out[0], out[1]) if isinstance(out, tuple) else (out, None)
                # d(obj / accumulate): the scale goes in as the seed gradient (a division launch and its backward less; the same bits for a power of two such as the reference's 2)
                obj.backward(torch.full_like(obj, 1.0 / self.accumulate))
            rec = [obj.detach()]
            if extra is not None:
                rec += [extra['latent'].detach(), extra['recon'].detach()]
            outs.append(torch.stack([r.reshape(()) for r in rec]))
        return torch.stack(outs)           # [accumulate, 1 or 3]

    def capture(self, batches):
        """batches: `accumulate` example input tensors (device).  Runs two eager warm-up passes (their gradients are
        discarded by the caller's zero_grad) and captures the third."""
        self.static_x = [b.clone() for b in batches]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                _invalidate(self.model)
                self._run()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        _invalidate(self.model)
        from ddk import ops
        self.graph = torch.cuda.CUDAGraph()
        self._keep.clear()
        with ops.graph_owner(self._keep), torch.cuda.graph(self.graph):
            self.outputs = self._run()
        # mark the copies stale again (eager code refreshes them before use) and drop those allocated in the graph's private pool
        _invalidate(self.model)
        return self

    def replay(self, batches):
        for dst, src in zip(self.static_x, batches):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs
