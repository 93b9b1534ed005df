"""ddk_pack_jobs: every kernel-layout weight copy of the training path refreshed by one launch -- bit-identical to the
single-tensor pack entry points (reference trainers/trainer_ddpm.py:142-144 changes the weights once per optimiser step)."""
import pytest
import torch

from ddk import lib as L
from ddk import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _weights():
    g = torch.Generator().manual_seed(7)
    shapes = [(128, 128, 3, 3), (256, 384, 3, 3), (128, 8, 3, 3), (32, 3, 3, 3), (64, 128, 1, 1), (128, 128, 4, 4), (384, 128, 1, 1),
              (8, 128, 1, 1), (96, 72, 3, 3)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]


def _all_copies(ws):
    """every (tag, weight, fn) the autograd path asks for (ddk/autograd.py), on shapes that cover each job kind"""
    want = []
    for w in ws:
        o, i, kh, kw = w.shape
        want.append(("fwd", w, ops.pack_conv_weight))
        if (kh, kw) == (4, 4):
            want.append(("fwdT", w, ops.pack_convT_weight))
            continue
        want.append((("dgrad", ops.pad32(i)), w, lambda t, n=ops.pad32(i): ops.pack_conv_weight_dgrad(t, i_pad=n)))
        if (kh, kw) == (3, 3):
            want.append(("wino", w, ops.pack_conv_weight_wino))
            lo, hi = (0, i) if i < 64 else (32, i)
            def pk(t, lo=lo, hi=hi, o=o, i=i):
                out = torch.empty((ops.pad32(o) // 32, 16, hi - lo, 32), device=t.device)
                L.check(L.load().ddk_pack_conv_weight_wino_dgrad(L.ptr(t), L.ptr(out), o, i, lo, hi, ops.pad32(o), L.stream()), "wino_dgrad")
                return out
            want.append((("wino_dgrad", lo, hi), w, pk))
    return want


def test_one_launch_refreshes_every_copy_bit_exactly():
    ops.weights_changed()
    ws = _weights()
    want = _all_copies(ws)
    first = [ops.cached_pack(tag, w, fn) for tag, w, fn in want]
    ptrs = [t.data_ptr() for t in first]
    with torch.no_grad():
        for k, w in enumerate(ws):
            w.mul_(1.0 + 0.25 * (k + 1)).add_(0.125)
    ops.weights_changed()                                          # what FusedAdam.step does after its in-place update
    before = dict(ops.pack_stats)
    again = [ops.cached_pack(tag, w, fn) for tag, w, fn in want]
    assert ops.pack_stats["batched_launches"] == before["batched_launches"] + 1          # the first stale hit refreshed all of them
    assert ops.pack_stats["batched_jobs"] == before["batched_jobs"] + len(want)
    assert ops.pack_stats["single"] == before["single"]
    assert [t.data_ptr() for t in again] == ptrs                   # in place: a captured graph keeps reading the same buffers
    for (tag, w, fn), got in zip(want, again):
        ref = fn(w.detach())
        assert torch.equal(got, ref), tag
    # a second change of ONE weight: only that weight's copies are stale, the others are not rewritten
    with torch.no_grad():
        ws[0].mul_(0.5)
    n0 = ops.pack_stats["batched_jobs"]
    for tag, w, fn in want:
        assert torch.equal(ops.cached_pack(tag, w, fn), fn(w.detach())), tag
    assert ops.pack_stats["batched_jobs"] - n0 == sum(1 for _, w, _ in want if w is ws[0])


def test_refresh_inside_a_captured_graph():
    """trainers/graph_step.py: the accumulation graph starts with the refresh, so a replay after an optimiser step sees the new weights"""
    ops.weights_changed()
    ws = _weights()[:4]
    want = _all_copies(ws)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                                          # warm-up passes register the jobs and build the device table
            ops.weights_changed()
            outs = [ops.cached_pack(tag, w, fn) for tag, w, fn in want]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ops.weights_changed()
    before = dict(ops.pack_stats)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = [ops.cached_pack(tag, w, fn) for tag, w, fn in want]
        total = torch.stack([o.sum() for o in outs])
    assert ops.pack_stats["single"] == before["single"] and ops.pack_stats["batched_launches"] == before["batched_launches"] + 1
    with torch.no_grad():
        for w in ws:
            w.mul_(-2.0)
    g.replay()
    torch.cuda.synchronize()
    for (tag, w, fn), got in zip(want, outs):
        assert torch.equal(got, fn(w.detach())), tag
    assert torch.equal(total, torch.stack([fn(w.detach()).sum() for _, w, fn in want]))
    ops.weights_changed()


def test_a_second_model_at_the_same_addresses_gets_its_own_table():
    """train one model, drop it, build another of the same shapes: the allocator hands out the same addresses, so the cache keys
    (tag, address, shape) repeat -- the device job table must still be rebuilt for the new copies' buffers (a stale table made
    ddk_pack_jobs write through freed pointers: a memory fault in tools/train_bench.py's third configuration).  Emulated here by new
    Parameter objects over the SAME storage."""
    ops.weights_changed()
    store = [w.detach() for w in _weights()[:5]]
    for round_ in range(3):
        ws = [torch.nn.Parameter(t) for t in store]              # new tensor objects, same addresses and shapes
        want = _all_copies(ws)
        outs = [ops.cached_pack(tag, w, fn) for tag, w, fn in want]
        with torch.no_grad():
            for w in ws:
                w.mul_(1.0 + round_).add_(0.5)
        ops.weights_changed()
        outs = [ops.cached_pack(tag, w, fn) for tag, w, fn in want]              # one batched refresh, into THIS round's copies
        for (tag, w, fn), got in zip(want, outs):
            assert torch.equal(got, fn(w.detach())), (round_, tag)
        del ws, want, outs
        ops.weights_changed()                                                    # prunes the dead entries


def test_layout_rejects_bad_jobs():
    jobs = (L.PackJob * 1)()
    jobs[0].src, jobs[0].dst, jobs[0].kind = 256, 512, 3
    jobs[0].p[0], jobs[0].p[1], jobs[0].p[2] = 8, 8, 40                        # i_pad not a multiple of 32
    assert L.load().ddk_pack_jobs_layout(jobs, 1) < 0
    jobs[0].kind = 9
    assert L.load().ddk_pack_jobs_layout(jobs, 1) < 0
