"""Backward-kernel parity on the GPU: every HIP backward op (through ddk.autograd) against torch-CPU autograd of the
oracle's functional ops on the same inputs, then a whole training objective against the reference goldens (G6)."""
import os

import numpy as np
import pytest
import contextlib

import torch
import torch.nn.functional as F

from helpers import ddpm_cfg, dddpm_cfg, det_load, golden, rel_err, to_nchw, to_nhwc, unet_cfg
from oracle import unet_ref as U
from utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(scope="module")
def AG():
    from ddk import autograd as ag
    return ag


def grads_cpu(fn, *tensors):
    leaves = [t.clone().requires_grad_(True) for t in tensors]
    out = fn(*leaves)
    g = rnd(*out.shape, seed=999)
    gs = torch.autograd.grad(out, leaves, g, allow_unused=True)
    return out.detach(), g, gs


CONV_BWD = [
    ("s1", 2, 8, 8, 32, 0, 64), ("s1", 4, 16, 16, 128, 0, 128), ("s1", 2, 8, 8, 64, 64, 64), ("s1", 3, 5, 7, 32, 0, 32),
    ("s1", 32, 4, 4, 256, 0, 256), ("s2", 4, 16, 16, 64, 0, 64), ("s2", 2, 8, 8, 128, 0, 128),
    ("1x1", 4, 8, 8, 128, 0, 384), ("1x1", 2, 4, 4, 64, 64, 128), ("T", 4, 4, 4, 64, 0, 64), ("T", 2, 8, 8, 128, 0, 128),
    # halo weight-gradient kernel (N % 64 == 0, C % 64 == 0): row chunks at W = 32 / 8 (odd batch) / 64 (row segments)
    ("s1", 2, 32, 32, 64, 0, 64), ("s1", 3, 8, 8, 128, 0, 64), ("s1", 1, 64, 64, 64, 0, 128), ("s1", 6, 4, 4, 64, 64, 128),
    # the narrow convs of the dDDPM encoder / decoder at many pixels: one- and two-wave tiles, up to 256 pixel splits
    ("s1", 4, 64, 64, 32, 0, 32), ("1x1", 4, 64, 64, 64, 0, 32), ("1x1", 5, 64, 64, 32, 0, 64), ("1x1", 16, 64, 64, 64, 0, 64),
    # narrow halo weight-gradient kernel (N == C == 32, >= 256 chunks): row segments (W = 64), whole rows at W = 32 / 16 / 8, ragged
    # chunk counts per wave
    ("s1", 8, 32, 32, 32, 0, 32), ("s1", 64, 16, 16, 32, 0, 32), ("s1", 136, 8, 8, 32, 0, 32), ("s1", 3, 64, 64, 32, 0, 32),
]


@pytest.mark.parametrize("kind,B,H,W,c0,c1,N", CONV_BWD)
def test_conv_backward(AG, kind, B, H, W, c0, c1, N):
    from ddk import ops
    cin = c0 + c1
    x = rnd(B, cin, H, W, seed=1)
    bias = rnd(N, seed=3, scale=0.1)
    if kind == "T":
        w = rnd(cin, N, 4, 4, seed=2, scale=(cin * 4) ** -0.5)
        f = lambda xx, ww, bb: F.conv_transpose2d(xx, ww, bb, stride=2, padding=1)
        code = ops.CONVT4X4_S2
    else:
        k = 1 if kind == "1x1" else 3
        w = rnd(N, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
        f = lambda xx, ww, bb: F.conv2d(xx, ww, bb, stride=2 if kind == "s2" else 1, padding=k // 2)
        code = {"s1": ops.CONV3X3_S1, "s2": ops.CONV3X3_S2, "1x1": ops.CONV1X1}[kind]
    out_ref, g, (gx, gw, gb) = grads_cpu(f, x, w, bias)
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous().requires_grad_(True)
    x1 = xh[..., c0:].contiguous().requires_grad_(True) if c1 else None
    wd, bd = w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    out = AG.conv(code, x0, wd, bd, x2=x1)
    assert rel_err(to_nchw(out.detach().cpu()), out_ref) < 2e-5
    out.backward(to_nhwc(g).to(DEV))
    gxh = to_nhwc(gx)
    assert rel_err(x0.grad.cpu(), gxh[..., :c0]) < 3e-5
    if c1:
        assert rel_err(x1.grad.cpu(), gxh[..., c0:]) < 3e-5
    assert rel_err(wd.grad.cpu(), gw) < 3e-5
    assert rel_err(bd.grad.cpu(), gb) < 3e-5


def test_conv_backward_accumulates_into_existing_grads(AG):
    """two backward passes into pre-existing .grad buffers (the optimiser's flat buffer): weight AND bias gradients are added in
    place by the weight-gradient launches (ddk_conv_wgrad_bias)"""
    from ddk import ops
    x, w, b = rnd(3, 64, 16, 16, seed=31), rnd(32, 64, 3, 3, seed=32, scale=0.05), rnd(32, seed=33, scale=0.1)
    out_ref, g, (gx, gw, gb) = grads_cpu(lambda a, ww, bb: F.conv2d(a, ww, bb, padding=1), x, w, b)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    wd.grad, bd.grad = torch.ones_like(wd), torch.full_like(bd, 2.0)
    for _ in range(2):
        xh = to_nhwc(x).to(DEV).requires_grad_(True)
        AG.conv(ops.CONV3X3_S1, xh, wd, bd).backward(to_nhwc(g).to(DEV))
    assert rel_err(wd.grad.cpu(), 1 + 2 * gw) < 3e-5 and rel_err(bd.grad.cpu(), 2 + 2 * gb) < 3e-5


def test_deferred_reduces_leave_temporaries_alone(AG):
    """inside ops.deferred_wgrad() (the trainers' accumulation block) only gradients that land in pre-existing `.grad` buffers wait
    for the end of the pass; a parameter WITHOUT such a buffer gets its gradient through autograd as a temporary, which must be
    complete when the Function returns -- first pass (no .grad yet) and second pass (.grad exists) both against torch-CPU"""
    from ddk import ops
    x, w, b = rnd(3, 64, 16, 16, seed=51), rnd(32, 64, 3, 3, seed=52, scale=0.05), rnd(32, seed=53, scale=0.1)
    out_ref, g, (gx, gw, gb) = grads_cpu(lambda a, ww, bb: F.conv2d(a, ww, bb, padding=1), x, w, b)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    for k in (1, 2):
        with ops.deferred_wgrad():
            xh = to_nhwc(x).to(DEV).requires_grad_(True)
            AG.conv(ops.CONV3X3_S1, xh, wd, bd).backward(to_nhwc(g).to(DEV))
        assert rel_err(wd.grad.cpu(), k * gw) < 3e-5 and rel_err(bd.grad.cpu(), k * gb) < 3e-5
        assert rel_err(to_nchw(xh.grad.cpu()), gx) < 3e-5


@pytest.mark.parametrize("B,H,W,C,M", [(2, 16, 16, 64, 32), (3, 8, 12, 32, 32), (1, 64, 64, 64, 32), (64, 4, 4, 128, 64)])
def test_preact_conv_chain(AG, B, H, W, C, M):
    """conv1x1(mish(x)) -> conv3x3(mish(.)) -> conv1x1(mish(.)) + x with the Mish forward written by the producing conv's epilogue
    (ddk_conv_args.mish_out) and Mish' applied by the input-gradient convs (dmish_src) == torch autograd of the same chain
    (the dDDPM encoder / decoder block, convblocks.py:112-130)"""
    from ddk import ops
    x = rnd(B, C, H, W, seed=41)
    w1, b1 = rnd(M, C, 1, 1, seed=42, scale=C ** -0.5), rnd(M, seed=43, scale=0.1)
    w2, b2 = rnd(M, M, 3, 3, seed=44, scale=(9 * M) ** -0.5), rnd(M, seed=45, scale=0.1)
    w3, b3 = rnd(C, M, 1, 1, seed=46, scale=M ** -0.5), rnd(C, seed=47, scale=0.1)

    def ref(xx, a1, c1, a2, c2, a3, c3):
        h = F.conv2d(F.mish(xx), a1, c1)
        h = F.conv2d(F.mish(h), a2, c2, padding=1)
        return F.conv2d(F.mish(h), a3, c3) + xx
    out_ref, g, grads = grads_cpu(ref, x, w1, b1, w2, b2, w3, b3)
    xd = to_nhwc(x).to(DEV).requires_grad_(True)
    ps = [t.to(DEV).requires_grad_(True) for t in (w1, b1, w2, b2, w3, b3)]
    a = ops.mish(xd.detach())
    h, a = AG.preact_conv(ops.CONV1X1, xd, a, ps[0], ps[1])
    assert rel_err(a.cpu(), F.mish(h.detach().cpu())) < 5e-6
    h, a = AG.preact_conv(ops.CONV3X3_S1, h, a, ps[2], ps[3])
    out, none = AG.preact_conv(ops.CONV1X1, h, a, ps[4], ps[5], resid=xd, want_act=False)
    assert none is None and rel_err(to_nchw(out.detach().cpu()), out_ref) < 2e-5
    out.backward(to_nhwc(g).to(DEV))
    assert rel_err(xd.grad.cpu(), to_nhwc(grads[0])) < 3e-5
    for got, want in zip(ps, grads[1:]):
        assert rel_err(got.grad.cpu(), want) < 3e-5


def test_conv_backward_padded_input_and_residual(AG):
    """first UNet conv: 8 real channels inside a 32-channel padded tensor; to_out-style fused residual"""
    from ddk import ops
    x, w, b = rnd(2, 8, 8, 8, seed=4), rnd(64, 8, 3, 3, seed=5, scale=0.1), rnd(64, seed=6, scale=0.1)
    r = rnd(2, 64, 8, 8, seed=7)
    out_ref, g, (gx, gw, gb, gr) = grads_cpu(lambda a, ww, bb, rr: F.conv2d(a, ww, bb, padding=1) + rr, x, w, b, r)
    xp = ops.nchw_to_nhwc(x.to(DEV), 32).requires_grad_(True)
    wd, bd, rd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True), to_nhwc(r).to(DEV).requires_grad_(True)
    out = AG.conv(ops.CONV3X3_S1, xp, wd, bd, resid=rd)
    out.backward(to_nhwc(g).to(DEV))
    assert rel_err(to_nchw(out.detach().cpu()), out_ref) < 2e-5
    assert rel_err(xp.grad[..., :8].cpu(), to_nhwc(gx)) < 3e-5 and float(xp.grad[..., 8:].abs().max()) == 0.0
    assert rel_err(wd.grad.cpu(), gw) < 3e-5 and rel_err(bd.grad.cpu(), gb) < 3e-5 and rel_err(rd.grad.cpu(), to_nhwc(gr)) < 1e-6


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 32), (4, 16, 16, 128), (8, 32, 32, 128), (3, 4, 4, 256), (2, 16, 16, 256),
                                     (2, 48, 48, 128),     # 36864 elements per group: streamed large-slab path, 2 splits
                                     (1, 100, 90, 64)])    # 72000 per group: 4 splits, ragged last split
def test_groupnorm_mish_backward(AG, B, H, W, C):
    x = rnd(B, C, H, W, seed=10, scale=2.0) + 0.3
    g, b = 1 + 0.1 * rnd(C, seed=11), 0.1 * rnd(C, seed=12)
    temb, add = rnd(B, C, seed=13), rnd(B, C, H, W, seed=14)
    f = lambda xx, gg, bb, tt, aa: U.mish(F.group_norm(xx, 8, gg, bb, 1e-5)) + tt[:, :, None, None] + aa
    out_ref, go, (gx, gg, gb, gt, ga) = grads_cpu(f, x, g, b, temb, add)
    xd = to_nhwc(x).to(DEV).requires_grad_(True)
    gd, bd = g.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    td, ad = temb.to(DEV).requires_grad_(True), to_nhwc(add).to(DEV).requires_grad_(True)
    out = AG.groupnorm_mish(xd, gd, bd, temb=td, addend=ad)
    assert rel_err(to_nchw(out.detach().cpu()), out_ref) < 5e-6
    out.backward(to_nhwc(go).to(DEV))
    assert rel_err(xd.grad.cpu(), to_nhwc(gx)) < 3e-5
    assert rel_err(gd.grad.cpu(), gg) < 3e-5 and rel_err(bd.grad.cpu(), gb) < 3e-5
    assert rel_err(td.grad.cpu(), gt) < 3e-5 and rel_err(ad.grad.cpu(), to_nhwc(ga)) < 1e-6


def test_groupnorm_dropout_consistency(AG):
    """Dropout(p): ~p of the outputs are zeroed and scaled by 1/(1-p); the backward uses the identical mask."""
    B, H, W, C, p = 4, 16, 16, 128, 0.1
    x = (rnd(B, H, W, C, seed=20) + 3.0).to(DEV).requires_grad_(True)      # mish(gn)+temb is never exactly 0 here
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    temb = torch.full((B, C), 5.0, device=DEV).requires_grad_(True)
    y = AG.groupnorm_mish(x, g, b, temb=temb, drop_p=p, seed=1234, layer=3)
    y0 = AG.groupnorm_mish(x.detach(), g, b, temb=temb.detach(), drop_p=0.0)
    dropped = (y == 0)
    frac = float(dropped.float().mean())
    assert abs(frac - p) < 0.01
    assert torch.allclose(y[~dropped], y0[~dropped] / (1 - p), rtol=1e-6)
    assert torch.equal(y, AG.groupnorm_mish(x, g, b, temb=temb, drop_p=p, seed=1234, layer=3))          # deterministic
    assert not torch.equal(dropped, AG.groupnorm_mish(x, g, b, temb=temb, drop_p=p, seed=1234, layer=4) == 0)
    y.sum().backward()
    # dtemb[b][c] = sum_hw mask/(1-p)
    want = (~dropped).float().sum(dim=(1, 2)) / (1 - p)
    assert torch.allclose(temb.grad, want, rtol=1e-5)


def test_groupnorm_dropout_large_slab(AG):
    """Same mask contract on the streamed large-slab path (64x64 x 16 channels per group = 65536 elements)."""
    B, H, W, C, p = 2, 64, 64, 128, 0.1
    x = (rnd(B, H, W, C, seed=21) + 3.0).to(DEV).requires_grad_(True)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    temb = torch.full((B, C), 5.0, device=DEV).requires_grad_(True)
    y = AG.groupnorm_mish(x, g, b, temb=temb, drop_p=p, seed=99, layer=7)
    y0 = AG.groupnorm_mish(x.detach(), g, b, temb=temb.detach(), drop_p=0.0)
    dropped = (y == 0)
    assert abs(float(dropped.float().mean()) - p) < 0.005
    assert torch.allclose(y[~dropped], y0[~dropped] / (1 - p), rtol=1e-6)
    y.sum().backward()
    want = (~dropped).float().sum(dim=(1, 2)) / (1 - p)
    assert torch.allclose(temb.grad, want, rtol=1e-5)


@pytest.mark.parametrize("C", [32, 64, 128, 256])
def test_chan_layernorm_backward(AG, C):
    x = rnd(3, C, 6, 5, seed=30, scale=2.0) + 0.5
    g, b = 1 + 0.1 * rnd(1, C, 1, 1, seed=31), 0.1 * rnd(1, C, 1, 1, seed=32)
    out_ref, go, (gx, gg, gb) = grads_cpu(lambda a, gg_, bb_: U.chan_layernorm(a, gg_, bb_), x, g, b)
    xd = to_nhwc(x).to(DEV).requires_grad_(True)
    gd, bd = g.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = AG.ChanLayerNormFn.apply(xd, gd, bd, 1e-5)
    out.backward(to_nhwc(go).to(DEV))
    assert rel_err(xd.grad.cpu(), to_nhwc(gx)) < 3e-5
    assert rel_err(gd.grad.cpu(), gg) < 3e-5 and rel_err(bd.grad.cpu(), gb) < 3e-5
    # ddk_chan_layernorm_bwd_add: the Residual's gradient added by the same launch == a separate add, bit for bit
    from ddk import ops
    dy, extra = to_nhwc(go).to(DEV), to_nhwc(rnd(3, C, 6, 5, seed=33)).to(DEV)
    dx0, _, _ = ops.chan_layernorm_bwd(xd.detach(), gd.detach(), dy)
    dx1, _, _ = ops.chan_layernorm_bwd(xd.detach(), gd.detach(), dy, addend=extra)
    assert torch.equal(dx1, dx0 + extra)


@pytest.mark.parametrize("B,H,W", [(2, 4, 4), (3, 8, 8), (2, 16, 16), (1, 10, 13),
                                   (1, 64, 64), (2, 48, 40), (1, 128, 96)])     # pixel-split statistics / dctx (16, 7 ragged, 48 splits)
def test_linattn_backward(AG, B, H, W):
    qkv = rnd(B, 384, H, W, seed=40, scale=1.2)

    def f(t):
        q, k, v = t.reshape(B, 3, 4, 32, H * W).unbind(1)
        ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
        return torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, 128, H, W)
    out_ref, go, (gq,) = grads_cpu(f, qkv)
    qd = to_nhwc(qkv).to(DEV).requires_grad_(True)
    out = AG.LinAttnFn.apply(qd, 4)
    out.backward(to_nhwc(go).to(DEV))
    assert rel_err(to_nchw(out.detach().cpu()), out_ref) < 1e-5
    assert rel_err(qd.grad.cpu(), to_nhwc(gq)) < 3e-5
    # the statistics recomputed by the backward's own launches (ddk_linattn_bwd_recompute) == saved by a forward launch, bit for bit
    from ddk import ops
    _, cx, stats = ops.linattn_train(qd.detach(), 4)
    dq_saved = ops.linattn_bwd(qd.detach(), to_nhwc(go).to(DEV), cx, stats, 4)
    assert torch.equal(dq_saved, qd.grad)


def test_small_n_conv_and_elementwise_backward(AG):
    for C, n_out in ((128, 8), (128, 3), (32, 1), (64, 3)):
        a, w, b = rnd(2, C, 9, 7, seed=50), rnd(n_out, C, 1, 1, seed=51, scale=C ** -0.5), rnd(n_out, seed=52)
        out_ref, go, (ga, gw, gb) = grads_cpu(lambda x, ww, bb: F.conv2d(x, ww, bb), a, w, b)
        ad = to_nhwc(a).to(DEV).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        out = AG.SmallNConvFn.apply(ad, wd, bd)
        out.backward(to_nhwc(go).to(DEV))
        assert rel_err(ad.grad.cpu(), to_nhwc(ga)) < 3e-5 and rel_err(wd.grad.cpu(), gw) < 3e-5 and rel_err(bd.grad.cpu(), gb) < 3e-5
    x = rnd(2, 64, 8, 12, seed=53, scale=3.0)
    for fn_gpu, fn_cpu in ((AG.MishFn.apply, U.mish), (AG.TanhFn.apply, torch.tanh),
                           (AG.AvgPool2Fn.apply, lambda t: F.avg_pool2d(t, 2, 2)),
                           (AG.UpNearest2Fn.apply, lambda t: F.interpolate(t, scale_factor=2))):
        out_ref, go, (gx,) = grads_cpu(fn_cpu, x)
        xd = to_nhwc(x).to(DEV).requires_grad_(True)
        out = fn_gpu(xd)
        out.backward(to_nhwc(go).to(DEV))
        assert rel_err(xd.grad.cpu(), to_nhwc(gx)) < 1e-5


@pytest.mark.parametrize("mode,M,N,K", [(0, 64, 128, 4096), (0, 64, 512, 4736), (1, 64, 96, 2048), (0, 8, 32, 1000), (2, 96, 64, 64)])
def test_small_gemm_long_contraction(mode, M, N, K):
    """the time-embedding products: C = A B / A B^T / A^T B; a long contraction on few tiles is split over workgroups with a
    fixed-order sum (dact = dout Wcat has K = all block channels, ~5000)"""
    from ddk import ops
    a = rnd(K, M, seed=51) if mode == 2 else rnd(M, K, seed=51)
    b = rnd(N, K, seed=52) if mode == 1 else rnd(K, N, seed=52)
    ref = (a.t().double() @ b.double()) if mode == 2 else (a.double() @ (b.t().double() if mode == 1 else b.double()))
    ad, bd = a.to(DEV), b.to(DEV)
    out = torch.empty(M, N, device=DEV)
    ops.small_gemm(mode, ad, bd, out, M, N, K, ad.shape[1], bd.shape[1], N)
    assert rel_err(out.cpu().double(), ref) < 1e-5
    out2 = torch.empty(M, N, device=DEV)
    ops.small_gemm(mode, ad, bd, out2, M, N, K, ad.shape[1], bd.shape[1], N)
    assert torch.equal(out, out2)
    acc = torch.ones(M, N, device=DEV)
    ops.small_gemm(mode, ad, bd, acc, M, N, K, ad.shape[1], bd.shape[1], N, accumulate=True)
    assert rel_err(acc.cpu().double(), ref + 1) < 1e-5


def test_time_embedding_backward(AG):
    from ddk.plan import sinusoidal_freqs
    dim, B = 32, 5
    t = torch.tensor([0, 3, 250, 731, 999])
    w1, b1 = rnd(4 * dim, dim, seed=60, scale=dim ** -0.5), rnd(4 * dim, seed=61, scale=0.1)
    w2, b2 = rnd(dim, 4 * dim, seed=62, scale=(4 * dim) ** -0.5), rnd(dim, seed=63, scale=0.1)
    mlps = [(rnd(co, dim, seed=64 + i, scale=dim ** -0.5), rnd(co, seed=70 + i, scale=0.1)) for i, co in enumerate((32, 64, 64))]

    def f(w1_, b1_, w2_, b2_, *m):
        tv = F.linear(U.mish(F.linear(U.sinusoidal_embedding(t, dim), w1_, b1_)), w2_, b2_)
        a = U.mish(tv)
        return torch.cat([F.linear(a, m[2 * i], m[2 * i + 1]) for i in range(3)], dim=1)
    flat = [w1, b1, w2, b2] + [p for pair in mlps for p in pair]
    out_ref, go, gs = grads_cpu(f, *flat)
    dev = [p.to(DEV).requires_grad_(True) for p in flat]
    outs = AG.TimeEmbedFn.apply(t.to(DEV), sinusoidal_freqs(dim).to(DEV), *dev)      # one [B, C_out] output per block
    out = torch.cat(outs, dim=1)
    assert rel_err(out.detach().cpu(), out_ref) < 2e-5
    out.backward(go.to(DEV))
    for p, g in zip(dev, gs):
        assert rel_err(p.grad.cpu(), g) < 3e-5


def test_optimizer_kernels_match_oracle():
    from ddk import ops
    from oracle import train_ref as TR
    n = 100003
    p, g = rnd(n, seed=80), rnd(n, seed=81, scale=3.0)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = p.to(DEV), g.to(DEV), m.to(DEV), v.to(DEV)
    grads = {"a": g[:5000].clone(), "b": g[5000:].clone()}
    clipped, total = TR.clip_grads(grads)
    nc = ops.grad_norm_clip(gd, 1.0).cpu()
    assert abs(float(nc[0]) / float(total) - 1) < 1e-5
    gc = torch.cat([clipped["a"], clipped["b"]])
    for step in (1, 2, 3):
        p, m, v = TR.adam_step(p, gc, m, v, step, 2e-4)
        ops.adam_step_(pd, gd, md, vd, 2e-4, step, clip=nc.to(DEV))
        assert (pd.cpu() - p).abs().max() < 2e-7 and rel_err(md.cpu(), m) < 1e-5 and rel_err(vd.cpu(), v) < 1e-5
    e = rnd(n, seed=82)
    ed = e.to(DEV)
    ops.ema_update_(ed, pd, 0.995)
    assert (ed.cpu() - (e * 0.995 + (1 - 0.995) * pd.cpu())).abs().max() < 1e-6


# ---------------------------------------------------------------- whole objective vs the reference goldens (G6)
def _g6_model(tag):
    from models import DDPM, DownsampleDDPM, DownsampleDDPMAutoencoder, Unet
    if tag == "ddpm":
        cfg = ddpm_cfg(32, 3, 16)
        m = DDPM(cfg, Unet(cfg), DEV, 3)
        return det_load(m).to(DEV).train(), (4, 3, 16, 16), (4, 3, 16, 16)
    cfg = dddpm_cfg(32, 32, 2)
    cls = DownsampleDDPMAutoencoder if tag == "dddpm_ae" else DownsampleDDPM
    return det_load(cls(cfg, Unet(cfg), DEV, 3)).to(DEV).train(), (4, 3, 32, 32), (4, 8, 8, 8)


@pytest.mark.parametrize("tag", ["ddpm", "dddpm_ae"])
def test_objective_and_grads_vs_reference(tag):
    g = golden("g6_train")
    model, xshape, eshape = _g6_model(tag)
    probe = [str(n) for n in g[f"{tag}_probe_names"]]
    params = dict(model.named_parameters())
    objs = []
    for mb in range(2):
        x = syn.synthetic_input(xshape, f"g6.{tag}.x0{mb}").to(DEV)
        tt = torch.tensor([0, 40, 500, 999 - mb], device=DEV)
        eps = syn.synthetic_normal(eshape, f"g6.{tag}.eps0{mb}").to(DEV)
        model.t_sample = lambda n, tt=tt: tt
        orig = torch.randn_like
        torch.randn_like = lambda z, eps=eps: eps
        try:
            res = model(x)
        finally:
            torch.randn_like = orig
        obj = res[0] if isinstance(res, tuple) else res
        if isinstance(res, tuple):
            assert abs(float(res[1]["latent"]) / float(g[f"{tag}_latent{mb}"]) - 1) < 1e-4
            assert abs(float(res[1]["recon"]) / float(g[f"{tag}_recon{mb}"]) - 1) < 1e-4
        (obj / 2).backward()
        objs.append(float(obj))
    assert np.allclose(objs, g[f"{tag}_obj0"], rtol=1e-4), (objs, g[f"{tag}_obj0"])
    for n in probe:
        assert rel_err(params[n].grad.cpu(), g[f"{tag}_grad_{n}"]) < 1e-3, n
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
    assert abs(float(total) / float(g[f"{tag}_gradnorm0"]) - 1) < 1e-3


def test_reconstruction_branch_on_its_own_stream_is_bit_identical():
    """DownsampleDDPMAutoencoder.losses in training: decoder + reconstruction loss (and their backward) on a second stream, beside the
    denoiser branch == everything on one stream: objective and every gradient bit for bit, eager with `.grad` slots + deferred reduces."""
    from models.diffusion import dddpm as D
    from ddk import ops
    model, xshape, eshape = _g6_model("dddpm_ae")
    x = syn.synthetic_input(xshape, "fork.x").to(DEV)
    tt = torch.tensor([0, 40, 500, 999], device=DEV)
    eps = syn.synthetic_normal(eshape, "fork.eps").to(DEV)
    model.t_sample = lambda n: tt
    def run(fork):
        D.RECON_SIDE_STREAM = fork
        orig = torch.randn_like
        torch.randn_like = lambda z: eps
        try:
            for p in model.parameters():
                p.grad = torch.zeros_like(p)
            torch.manual_seed(5)
            with ops.deferred_wgrad():
                obj, extra = model(x)
                obj.backward()
            torch.cuda.synchronize()
            return obj.detach().clone(), extra["recon"].detach().clone(), [p.grad.clone() for p in model.parameters()]
        finally:
            torch.randn_like = orig
            D.RECON_SIDE_STREAM = True
    o0, r0, g0 = run(False)
    o1, r1, g1 = run(True)
    assert torch.equal(o0, o1) and torch.equal(r0, r1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)


def test_fused_autoencoder_objective_matches_the_torch_expression():
    """ddk_ae_objective / AEObjectiveFn: objective, report values and every gradient of DownsampleDDPMAutoencoder.losses == the
    where / add / mean expression of dddpm.py:155-177 (samples on both sides of t_rec_max)."""
    from models.diffusion import dddpm as D
    model, xshape, eshape = _g6_model("dddpm_ae")
    x = syn.synthetic_input(xshape, "obj.x").to(DEV)
    tt = torch.tensor([0, 40, 500, 999], device=DEV)
    assert int((tt < model.t_rec_max).sum()) not in (0, 4)
    eps = syn.synthetic_normal(eshape, "obj.eps").to(DEV)
    model.t_sample = lambda n: tt
    def run(fused):
        D.FUSED_OBJECTIVE = fused
        orig = torch.randn_like
        torch.randn_like = lambda z: eps
        try:
            for p in model.parameters():
                p.grad = None
            obj, extra = model(x)
            obj.backward()
            torch.cuda.synchronize()
            return [obj.detach(), extra["latent"].detach(), extra["recon"].detach()], [p.grad.clone() for p in model.parameters()]
        finally:
            torch.randn_like = orig
            D.FUSED_OBJECTIVE = True
    v0, g0 = run(False)
    v1, g1 = run(True)
    for a, b in zip(v0, v1):
        assert abs(float(a) / float(b) - 1) < 1e-6
    for a, b in zip(g0, g1):
        assert rel_err(b.cpu(), a.cpu()) < 1e-6


def test_unet_grads_large_resolution_vs_oracle():
    """Full-resolution-style training (cfg5 shape class): 64x64 maps at 64 channels put 32768 elements in a GroupNorm
    group, i.e. the streamed large-slab kernels; 4096 pixels per sample in the linear attention.  Gradients of a
    weighted-sum objective against torch-CPU autograd through the oracle's functional UNet."""
    from models import Unet
    cfg = dict(unet_chan=64, unet_in=3, unet_dims=(1, 2), unet_dropout=0.0)
    model = Unet(cfg)
    sd = syn.fill_state_dict(model.state_dict(), 77)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    x = syn.synthetic_input((2, 3, 64, 64), "bigres.x")
    t = torch.tensor([3, 700])
    wgt = syn.synthetic_normal((2, 3, 64, 64), "bigres.w")

    probe = ["downs.0.0.block1.block.0.weight", "downs.0.0.block1.block.1.weight", "downs.0.2.fn.fn.to_qkv.weight",
             "ups.0.1.block2.block.0.weight", "final_conv.1.weight", "time_mlp.1.weight", "downs.0.1.mlp.1.bias"]
    ref_sd = {k: v.clone() for k, v in sd.items()}
    for k in probe:
        ref_sd[k].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    out_ref = U.unet_forward(ref_sd, cfg, xr, t)
    (out_ref * wgt).sum().backward()

    xd = x.to(DEV).requires_grad_(True)
    out = model(xd, t.to(DEV))
    assert rel_err(out.detach().cpu(), out_ref.detach()) < 5e-5
    (out * wgt.to(DEV)).sum().backward()
    params = dict(model.named_parameters())
    assert rel_err(xd.grad.cpu(), xr.grad) < 2e-4
    for k in probe:
        assert rel_err(params[k].grad.cpu(), ref_sd[k].grad) < 2e-4, k


def test_unet_grads_cfg5_full_resolution_vs_oracle():
    """cfg5 itself (CelebA-HQ 256x256 DDPM, unet_chan 128, dims (1,2,2,2), C_in 3): ONE micro-batch of B = 1 at the full
    256 x 256 resolution -- 65 536-pixel linear attention backward, streamed GroupNorm over 1 M-element groups, the Winograd
    forward / input-gradient convs and the weight-gradient GEMMs at their real sizes.  Output and gradients of the input and of six
    probe tensors against torch-CPU autograd through the oracle's functional UNet (reference: trainers/trainer_ddpm.py:118-144,
    models/unet/unet.py:74-104).  Tens of seconds of oracle time on the GPU box's host cores."""
    from models import Unet
    cfg = dict(unet_chan=128, unet_in=3, unet_dims=(1, 2, 2, 2), unet_dropout=0.0)
    model = Unet(cfg)
    sd = syn.fill_state_dict(model.state_dict(), 55)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    x = syn.synthetic_input((1, 3, 256, 256), "cfg5.train.x")
    t = torch.tensor([437])
    wgt = syn.synthetic_normal((1, 3, 256, 256), "cfg5.train.w")

    probe = ["downs.0.0.block1.block.0.weight", "downs.0.2.fn.fn.to_qkv.weight", "downs.1.0.res_conv.weight",
             "mid_block1.block2.block.1.weight", "ups.2.0.block1.block.0.weight", "final_conv.1.weight"]
    ref_sd = {k: v.clone() for k, v in sd.items()}
    for k in probe:
        ref_sd[k].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    out_ref = U.unet_forward(ref_sd, cfg, xr, t)
    (out_ref * wgt).sum().backward()

    xd = x.to(DEV).requires_grad_(True)
    out = model(xd, t.to(DEV))
    assert rel_err(out.detach().cpu(), out_ref.detach()) < 5e-5
    (out * wgt.to(DEV)).sum().backward()
    params = dict(model.named_parameters())
    assert rel_err(xd.grad.cpu(), xr.grad) < 3e-4
    for k in probe:
        assert rel_err(params[k].grad.cpu(), ref_sd[k].grad) < 3e-4, k



def test_deferred_weight_gradient_reduces_are_bit_identical():
    """ops.deferred_wgrad(): the slab reduces of many weight gradients in a few launches (ddk_wgrad_reduce_jobs: the records travel as
    kernel arguments, 48 to a launch) == one reduce launch per call, bit for bit -- 60 calls (two launches) over the shapes of the
    training path: 3x3 / 1x1 / stride-2 / transpose, dual-source windows, with and without the bias gradient."""
    from ddk import ops
    g = torch.Generator().manual_seed(11)
    cases = []
    for rep in range(10):
        for kind, B, H, C, N, k in ((ops.CONV3X3_S1, 4, 16, 64, 64, 3), (ops.CONV1X1, 4, 16, 128, 32, 1), (ops.CONV3X3_S1, 8, 32, 32, 32, 3),
                                   (ops.CONV3X3_S2, 4, 16, 32, 64, 3), (ops.CONV3X3_S1, 2, 8, 96, 128, 3), (ops.CONV1X1, 2, 8, 64, 96, 1)):
            Ho = H // 2 if kind == ops.CONV3X3_S2 else H
            x = torch.randn(B, H, H, C, generator=g).to(DEV)
            dy = torch.randn(B, Ho, Ho, N, generator=g).to(DEV)
            cw = C + (32 if rep % 2 else 0)                       # odd repetitions: the source is the second window of a wider weight
            cases.append((kind, x, dy, (N, cw, k, k), C, cw, cw - C, rep % 3 == 0))
    def run(defer):
        outs = []
        ctx = ops.deferred_wgrad() if defer else contextlib.nullcontext()
        with ctx:
            for kind, x, dy, wshape, c_real, cw, c_off, with_b in cases:
                gw = torch.full(wshape, 0.25, device=DEV)
                gb = torch.full((wshape[0],), -0.5, device=DEV) if with_b else None
                ops.conv_wgrad_(kind, x, dy, gw, c_real=c_real, cw=cw, c_off=c_off, grad_b=gb, persistent=True)
                outs.append((gw, gb))
        torch.cuda.synchronize()
        return outs
    a, b = run(False), run(True)
    for (gw0, gb0), (gw1, gb1) in zip(a, b):
        assert torch.equal(gw0, gw1)
        assert (gb0 is None) == (gb1 is None) and (gb0 is None or torch.equal(gb0, gb1))
    # two reduces into the same gradient elements are kept in separate launches, in call order
    kind, x, dy, wshape, c_real, cw, c_off, _ = cases[0]
    gw_a, gw_b = torch.zeros(wshape, device=DEV), torch.zeros(wshape, device=DEV)
    ops.conv_wgrad_(kind, x, dy, gw_a, c_real=c_real, cw=cw, c_off=c_off)
    ops.conv_wgrad_(kind, x, dy, gw_a, c_real=c_real, cw=cw, c_off=c_off)
    with ops.deferred_wgrad():
        ops.conv_wgrad_(kind, x, dy, gw_b, c_real=c_real, cw=cw, c_off=c_off, persistent=True)
        ops.conv_wgrad_(kind, x, dy, gw_b, c_real=c_real, cw=cw, c_off=c_off, persistent=True)
    assert torch.equal(gw_a, gw_b)
    # a gradient buffer that does NOT outlive the pass (no `.grad` slot: autograd takes the tensor at once) is reduced at once
    gw_c = torch.zeros(wshape, device=DEV)
    with ops.deferred_wgrad():
        ops.conv_wgrad_(kind, x, dy, gw_c, c_real=c_real, cw=cw, c_off=c_off)
        inside = gw_c.clone()
    assert torch.equal(inside, gw_c) and float(gw_c.abs().max()) > 0


def test_deferred_norm_parameter_gradients_are_bit_identical():
    """inside ops.deferred_wgrad() the GroupNorm / LayerNorm backward kernels only record their parameter-gradient row sums
    (ddk_rows_sum_jobs runs them 48 to a launch when the block ends) == one rows_sum_targets launch per call, bit for bit"""
    from ddk import ops
    g = torch.Generator().manual_seed(5)
    cases = []
    for k in range(30):
        c = (32, 64, 128, 256)[k % 4]
        b, h = (3, 8) if k % 2 else (5, 4)
        cases.append((torch.randn(b, h, h, c, generator=g).to(DEV), torch.randn(b, h, h, c, generator=g).to(DEV),
                      (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)))
    def run(defer):
        res = []
        ctx = ops.deferred_wgrad() if defer else contextlib.nullcontext()
        with ctx:
            for x, dy, ga, be in cases:
                c = x.shape[-1]
                acc = tuple(torch.full((c,), 0.5, device=DEV) for _ in range(3))
                dx, dtemb, sums = ops.groupnorm_mish_bwd(x, ga, be, dy, acc=acc)
                assert sums == [None, None, None]
                lacc = (torch.full((c,), -1.0, device=DEV), torch.full((c,), 2.0, device=DEV))
                ldx, _, _ = ops.chan_layernorm_bwd(x, ga.view(1, c, 1, 1), dy, acc=lacc)
                res.append((dx, dtemb) + acc + (ldx,) + lacc)
        torch.cuda.synchronize()
        return res
    a, b = run(False), run(True)
    for ra, rb in zip(a, b):
        for ta, tb in zip(ra, rb):
            assert torch.equal(ta, tb)


@pytest.mark.parametrize("B,H,W,C,S", [(4, 16, 16, 128, 2), (64, 2, 2, 256, 8), (3, 8, 8, 256, 4), (2, 32, 32, 128, 3)])
def test_groupnorm_train_slab_forms_are_bit_identical(B, H, W, C, S):
    """ddk_groupnorm_mish_train_fwd_slabs / ddk_groupnorm_mish_bwd_slabs: the tensor still in S split-K slabs, summed while loading
    == the slabs reduced first (slab order, then the conv bias), bit for bit; the forward also writes the reduced tensor."""
    from ddk import ops
    g = torch.Generator().manual_seed(5)
    slabs = torch.randn(S, B, H, W, C, generator=g).to(DEV)
    bias, gamma, beta = (torch.randn(C, generator=g).to(DEV) for _ in range(3))
    temb = torch.randn(B, C, generator=g).to(DEV)
    x = slabs[0].clone()
    for s_ in range(1, S):
        x += slabs[s_]
    x += bias
    for drop in (0.0, 0.1):
        y0 = ops.groupnorm_mish_train(x, gamma, beta, temb=temb, drop_p=drop, seed=7, layer=3)
        raw = torch.empty_like(x)
        y1 = ops.groupnorm_mish_train(raw, gamma, beta, temb=temb, drop_p=drop, seed=7, layer=3, slabs=slabs, conv_bias=bias)
        assert torch.equal(raw, x) and torch.equal(y0, y1)
        dy = slabs[0].clone()
        for s_ in range(1, S):
            dy += slabs[s_]
        dx0, dt0, sums0 = ops.groupnorm_mish_bwd(x, gamma, beta, dy, drop, 7, 3)
        dx1, dt1, sums1 = ops.groupnorm_mish_bwd(x, gamma, beta, slabs[0], drop, 7, 3, dy_slabs=slabs)
        assert torch.equal(dx0, dx1) and torch.equal(dt0, dt1)
        assert all(torch.equal(a, b) for a, b in zip(sums0, sums1))
    with pytest.raises(Exception):          # a group slab too large for the register-resident kernel has no slab form
        big = torch.zeros(2, 1, 256, 256, 128, device=DEV)
        ops.groupnorm_mish_train(torch.empty_like(big[0]), gamma[:128], beta[:128], slabs=big, conv_bias=bias[:128])


@pytest.mark.parametrize("B,H,W,C,N", [(2, 5, 5, 256, 256), (3, 7, 6, 128, 64), (2, 8, 8, 256, 128)])
def test_block_with_slab_fold_on_the_im2col_kernels_is_bit_identical(B, H, W, C, N):
    """ConvGNMishFn where the conv is NOT Winograd-eligible (odd maps) or is, with a k split: the slabs of the im2col / Winograd launch
    summed by the GroupNorm == reduced first; output and all gradients bit for bit."""
    from ddk import autograd as AG
    from ddk import ops
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(B, H, W, C, generator=g).to(DEV)
    w0 = (torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(DEV)
    b0, ga0, be0 = (torch.randn(N, generator=g).to(DEV) * 0.1 for _ in range(3))
    te = torch.randn(B, N, generator=g).to(DEV)
    dy = torch.randn(B, H, W, N, generator=g).to(DEV)
    def run(fold):
        AG.FOLD_SLABS = fold
        try:
            x, w, b, ga, be = (t.clone().requires_grad_(True) for t in (x0, w0, b0, ga0 + 1, be0))
            y = AG.conv_groupnorm_mish(x, w, b, ga, be, temb=te)
            y.backward(dy)
            torch.cuda.synchronize()
            return [y.detach()] + [t.grad for t in (x, w, b, ga, be)]
        finally:
            AG.FOLD_SLABS = True
    for a, bb in zip(run(False), run(True)):
        assert torch.equal(a, bb)
    lib = __import__("ddk.lib", fromlist=["load"]).load()
    wino = H % 2 == 0 and W % 2 == 0 and lib.ddk_conv_wino_splits(B, H, W, C, N) > 0
    splits = lib.ddk_conv_wino_splits(B, H, W, C, N) if wino else lib.ddk_conv_splits(ops.CONV3X3_S1, B, H, W, C, N)
    assert splits > 1, "the case is meant to split k"


def test_unet_training_with_slab_folds_is_bit_identical():
    """One forward + backward of the UNet (cfg3 shape: 16x16 latent, dims (1,2,2,2), batch 16, dropout on) with the split-K slabs
    summed inside the GroupNorm kernels == with one reduce launch per conv: loss and every gradient bit for bit.  Also: the
    placeholder check of the backward link."""
    from ddk import autograd as AG
    from models import Unet
    from trainers.autograd_unet import unet_forward_autograd
    cfg = unet_cfg(128, 8)
    cfg["unet_dims"] = (1, 2, 2, 2)
    cfg["unet_dropout"] = 0.1
    u = det_load(Unet(cfg), "latent_model.").to(DEV).train()
    x = syn.synthetic_normal((16, 16, 16, 8), "fold.x").to(DEV)
    t = (torch.arange(16, device=DEV) * 61) % 1000
    from ddk import ops
    def run(fold, side=False, skip=True):
        AG.FOLD_SLABS = fold
        AG.SKIP_HANDOFF = skip
        ops.WGRAD_SIDE_STREAM = side
        try:
            for p in u.parameters():
                p.grad = torch.zeros_like(p) if side else None     # `.grad` slots: the weight gradients are deferred (and forked)
            torch.manual_seed(3)
            with (ops.deferred_wgrad() if side else contextlib.nullcontext()):
                out = unet_forward_autograd(u, x, t)
                loss = (out * out).mean()
                loss.backward()
            torch.cuda.synchronize()
            return loss.detach().clone(), [p.grad.clone() for p in u.parameters()]
        finally:
            AG.FOLD_SLABS = True
            AG.SKIP_HANDOFF = True
            ops.WGRAD_SIDE_STREAM = True
    l0, g0 = run(False)
    l1, g1 = run(True)
    assert torch.equal(l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # the deferred weight-gradient launches on their side stream (fork after their operands, join before the reduce): same bits
    l2, g2 = run(True, side=True)
    assert torch.equal(l0, l2)
    for a, b in zip(g0, g2):
        assert torch.equal(a, b)
    # the up path's gradient of a skip tensor added by the Downsample conv's input-gradient launch == added by autograd
    l3, g3 = run(True, skip=False)
    assert torch.equal(l0, l3)
    for a, b in zip(g0, g3):
        assert torch.equal(a, b)
    link = AG.SlabLink()
    ph = link.put(torch.zeros(2, 1, 4, 4, 32, device=DEV))
    with pytest.raises(RuntimeError):
        link.take(ph + 1)                      # not the placeholder: some other gradient was mixed in
