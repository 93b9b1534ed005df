"""Seeded shape fuzz of the conv family (forward, input gradient, weight gradient) against torch-CPU: odd spatial sizes,
batch sizes that leave ragged tiles, channel counts on both sides of every kernel-selection threshold (halo kernel,
split-K, halo weight gradient, small-N tiles).  One process, ~70 cases, each a few MFLOP to a few GFLOP."""
import random

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, to_nchw, to_nhwc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rnd(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        kind = rng.choice(["s1", "s1", "s1", "s2", "1x1", "T"])
        c0 = rng.choice([32, 64, 96, 128, 256])
        c1 = rng.choice([0, 0, 0, 32, 128]) if kind in ("s1", "1x1") else 0
        n_out = rng.choice([32, 64, 128, 160, 256, 384])
        hw = rng.choice([(4, 4), (8, 8), (16, 16), (32, 32), (5, 7), (12, 20), (8, 32), (16, 8), (64, 64), (3, 3), (24, 24)])
        b = rng.choice([1, 2, 3, 5, 8, 17, 32])
        if kind == "T":
            n_out = c0 = rng.choice([32, 64, 128])
        # keep the CPU reference cheap
        while b * hw[0] * hw[1] * (c0 + c1) * n_out * (9 if kind in ("s1", "s2") else 16 if kind == "T" else 1) > 6e9 and b > 1:
            b = max(1, b // 2)
        out.append((kind, b, hw[0], hw[1], c0, c1, n_out, 1000 + i))
    return out


@pytest.mark.parametrize("kind,B,H,W,c0,c1,N,seed", _cases(48, 20260901))
def test_conv_forward_fuzz(kind, B, H, W, c0, c1, N, seed):
    from ddk import ops
    cin = c0 + c1
    x = _rnd(B, cin, H, W, seed=seed)
    k = {"s1": 3, "s2": 3, "1x1": 1, "T": 4}[kind]
    bias = _rnd(N, seed=seed + 1) * 0.1
    if kind == "T":
        w = _rnd(cin, N, 4, 4, seed=seed + 2) * (cin * 4) ** -0.5
        ref = F.conv_transpose2d(x, w, bias, stride=2, padding=1)
        wp, code = ops.pack_convT_weight(w.to(DEV)), ops.CONVT4X4_S2
    else:
        w = _rnd(N, cin, k, k, seed=seed + 2) * (cin * k * k) ** -0.5
        ref = F.conv2d(x, w, bias, stride=2 if kind == "s2" else 1, padding=k // 2)
        wp = ops.pack_conv_weight(w.to(DEV))
        code = {"s1": ops.CONV3X3_S1, "s2": ops.CONV3X3_S2, "1x1": ops.CONV1X1}[kind]
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    resid = _rnd(*ref.shape, seed=seed + 3)
    out = ops.conv(code, x0, wp, bias.to(DEV), x2=x1, resid=to_nhwc(resid).to(DEV), post_mish=bool(seed & 1))
    want = ref + resid
    if seed & 1:
        want = want * torch.tanh(F.softplus(want))
    assert rel_err(to_nchw(out.cpu()), want) < 3e-5


@pytest.mark.parametrize("kind,B,H,W,c0,c1,N,seed", [c for c in _cases(40, 777) if c[0] in ("s1", "s2", "1x1")][:22])
def test_conv_backward_fuzz(kind, B, H, W, c0, c1, N, seed):
    from ddk import autograd as AG, ops
    cin = c0 + c1
    k = 1 if kind == "1x1" else 3
    x = _rnd(B, cin, H, W, seed=seed).requires_grad_(True)
    w = (_rnd(N, cin, k, k, seed=seed + 2) * (cin * k * k) ** -0.5).requires_grad_(True)
    b = (_rnd(N, seed=seed + 1) * 0.1).requires_grad_(True)
    ref = F.conv2d(x, w, b, stride=2 if kind == "s2" else 1, padding=k // 2)
    go = _rnd(*ref.shape, seed=seed + 5)
    gx, gw, gb = torch.autograd.grad(ref, (x, w, b), go)
    code = {"s1": ops.CONV3X3_S1, "s2": ops.CONV3X3_S2, "1x1": ops.CONV1X1}[kind]
    xh = to_nhwc(x.detach()).to(DEV)
    x0 = xh[..., :c0].contiguous().requires_grad_(True)
    x1 = xh[..., c0:].contiguous().requires_grad_(True) if c1 else None
    wd, bd = w.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    out = AG.conv(code, x0, wd, bd, x2=x1)
    out.backward(to_nhwc(go).to(DEV))
    got_gx = x0.grad.cpu() if x1 is None else torch.cat([x0.grad.cpu(), x1.grad.cpu()], dim=-1)
    assert rel_err(to_nchw(out.detach().cpu()), ref.detach()) < 3e-5
    assert rel_err(got_gx, to_nhwc(gx)) < 5e-5
    assert rel_err(wd.grad.cpu(), gw) < 5e-5 and rel_err(bd.grad.cpu(), gb) < 5e-5


def _norm_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        c = rng.choice([32, 64, 128, 256])
        hw = rng.choice([(4, 4), (8, 8), (16, 16), (32, 32), (5, 7), (48, 40), (64, 64), (96, 80), (3, 11)])
        b = rng.choice([1, 2, 3, 6])
        out.append((b, hw[0], hw[1], c, 3000 + i))
    return out


@pytest.mark.parametrize("B,H,W,C,seed", _norm_cases(14, 4242))
def test_groupnorm_layernorm_attention_fuzz(B, H, W, C, seed):
    """GroupNorm+Mish (register-resident and multi-workgroup statistics), its train-mode forward / backward, channel
    LayerNorm and the linear-attention core (single workgroup, pixel-range splits, one-launch small maps) on the same shape."""
    from ddk import autograd as AG, ops
    from oracle import unet_ref as U
    x = _rnd(B, C, H, W, seed=seed) * 1.7 + 0.4
    g, b = 1 + 0.1 * _rnd(C, seed=seed + 1), 0.1 * _rnd(C, seed=seed + 2)
    temb, add = _rnd(B, C, seed=seed + 3), _rnd(B, C, H, W, seed=seed + 4)
    f = lambda xx, gg, bb, tt, aa: U.mish(F.group_norm(xx, 8, gg, bb, 1e-5)) + tt[:, :, None, None] + aa
    leaves = [t.clone().requires_grad_(True) for t in (x, g, b, temb, add)]
    ref = f(*leaves)
    go = _rnd(*ref.shape, seed=seed + 5)
    grads = torch.autograd.grad(ref, leaves, go)
    xh = to_nhwc(x).to(DEV)
    out = ops.groupnorm_mish(xh, g.to(DEV), b.to(DEV), temb=temb.to(DEV), addend=to_nhwc(add).to(DEV))
    assert rel_err(to_nchw(out.cpu()), ref.detach()) < 1e-5
    xd = xh.clone().requires_grad_(True)
    gd, bd = g.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    td, ad = temb.to(DEV).requires_grad_(True), to_nhwc(add).to(DEV).requires_grad_(True)
    out_t = AG.groupnorm_mish(xd, gd, bd, temb=td, addend=ad)
    assert rel_err(to_nchw(out_t.detach().cpu()), ref.detach()) < 1e-5
    out_t.backward(to_nhwc(go).to(DEV))
    assert rel_err(xd.grad.cpu(), to_nhwc(grads[0])) < 5e-5
    assert rel_err(gd.grad.cpu(), grads[1]) < 5e-5 and rel_err(bd.grad.cpu(), grads[2]) < 5e-5
    assert rel_err(td.grad.cpu(), grads[3]) < 5e-5
    # channel LayerNorm
    lg, lb = (1 + 0.1 * _rnd(1, C, 1, 1, seed=seed + 6)), 0.1 * _rnd(1, C, 1, 1, seed=seed + 7)
    ln = ops.chan_layernorm(xh, lg.to(DEV), lb.to(DEV))
    assert rel_err(to_nchw(ln.cpu()), U.chan_layernorm(x, lg, lb)) < 1e-5
    # linear attention core on a qkv tensor of this spatial size
    qkv = _rnd(B, 384, H, W, seed=seed + 8)
    want = U.linear_attention_core(qkv, 4) if hasattr(U, "linear_attention_core") else None
    got, _ = ops.linattn(to_nhwc(qkv).to(DEV), 4)
    if want is None:
        q, k, v = qkv.reshape(B, 3, 4, 32, H * W).unbind(1)
        k = k.softmax(dim=-1)
        ctx = torch.einsum("bhdn,bhen->bhde", k, v)
        want = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, 128, H, W)
    assert rel_err(to_nchw(got.cpu()), want) < 2e-5


def _wino_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        c0 = rng.choice([32, 64, 96, 128, 256])
        c1 = rng.choice([0, 0, 32, 128, 256])
        n_out = rng.choice([64, 128, 192, 256])
        hw = rng.choice([(2, 2), (4, 4), (8, 8), (16, 16), (32, 32), (6, 10), (12, 20), (8, 32), (2, 14), (24, 24), (64, 64)])
        b = rng.choice([1, 2, 3, 5, 8, 17, 32])
        while b * hw[0] * hw[1] * (c0 + c1) * n_out * 9 > 6e9 and b > 1:
            b = max(1, b // 2)
        out.append((b, hw[0], hw[1], c0, c1, n_out, 5000 + i))
    return out


@pytest.mark.parametrize("B,H,W,c0,c1,N,seed", _wino_cases(24, 20261004))
def test_conv_winograd_fuzz(B, H, W, c0, c1, N, seed):
    """conv3x3_wino_kernel on random eligible shapes (even H, W; cin % 32 == 0; N % 64 == 0): ragged tile blocks, every split
    count, concat sources, epilogue options -- against F.conv2d and bit-stable run to run."""
    from ddk import ops
    cin = c0 + c1
    x = _rnd(B, cin, H, W, seed=seed)
    w = _rnd(N, cin, 3, 3, seed=seed + 2) * (cin * 9) ** -0.5
    bias = _rnd(N, seed=seed + 1) * 0.1
    ref = F.conv2d(x, w, bias, padding=1)
    assert ops.L.load().ddk_conv_wino_splits(B, H, W, cin, N) >= 1
    wp, wu = ops.pack_conv_weight(w.to(DEV)), ops.pack_conv_weight_wino(w.to(DEV))
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    resid = _rnd(*ref.shape, seed=seed + 3)
    mish = bool(seed & 1)
    out = ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1, resid=to_nhwc(resid).to(DEV), post_mish=mish, w_wino=wu)
    want = ref + resid
    if mish:
        want = want * torch.tanh(F.softplus(want))
    assert rel_err(to_nchw(out.cpu()), want) < 2e-5
    assert torch.equal(out, ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1, resid=to_nhwc(resid).to(DEV), post_mish=mish, w_wino=wu))


def _local_cases(n, seed0):
    import random
    rng = random.Random(seed0)
    out = []
    for i in range(n):
        px = rng.choice([16, 16, 64, 64])
        h = rng.choice([2, 4, 8] if px == 16 else [2, 4, 8, 16, 32])
        w = px // h
        n_out = rng.choice([64, 128, 256])          # 8 groups: 8 / 16 / 32 channels per group
        c0 = rng.choice([32, 64, 96, 128, 160, 256])
        c1 = rng.choice([0, 0, 32, 64, 128])
        if px == 64 and c0 + c1 > 320:              # LDS budget of the Winograd form
            c1 = 0
        b = rng.choice([1, 2, 3, 7, 32])
        out.append((b, h, w, c0, c1, n_out, 7000 + i))
    return out


@pytest.mark.parametrize("B,H,W,c0,c1,N,seed", _local_cases(20, 20261005))
def test_conv_groupnorm_one_launch_fuzz(B, H, W, c0, c1, N, seed):
    """The image-local one-launch Block kernels (direct on 16-pixel maps, Winograd on 64-pixel maps) on random eligible shapes --
    non-square maps, every channels-per-group count, concat sources, shift / residual options -- against torch's
    conv2d -> group_norm -> mish (blocks.py:75-84, 110-115), and bit-stable run to run."""
    from ddk import ops
    cin = c0 + c1
    lib = ops.L.load()
    x = _rnd(B, cin, H, W, seed=seed)
    w = _rnd(N, cin, 3, 3, seed=seed + 2) * (cin * 9) ** -0.5
    bias = _rnd(N, seed=seed + 1) * 0.1
    gamma, beta = 1 + 0.2 * _rnd(N, seed=seed + 4), 0.2 * _rnd(N, seed=seed + 5)
    temb = _rnd(B, N, seed=seed + 6) if seed & 1 else None
    resid = _rnd(B, N, H, W, seed=seed + 3) if seed & 2 else None
    h = F.group_norm(F.conv2d(x, w, bias, padding=1), 8, gamma, beta, eps=1e-5)
    want = h * torch.tanh(F.softplus(h))
    if temb is not None:
        want = want + temb[:, :, None, None]
    if resid is not None:
        want = want + resid
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    kw = dict(temb=temb.to(DEV) if temb is not None else None, addend=to_nhwc(resid).to(DEV) if resid is not None else None, x2=x1)
    if H * W == 16:
        assert lib.ddk_conv3x3_gn_mish_ok(H, W, cin, c0, N, 8)
        run = lambda: ops.conv3x3_gn_mish(x0, ops.pack_conv_weight_local(w.to(DEV)), bias.to(DEV), gamma.to(DEV), beta.to(DEV), **kw)
    else:
        assert lib.ddk_conv3x3_gn_mish_wino_ok(H, W, cin, c0, N, 8)
        run = lambda: ops.conv3x3_gn_mish_wino(x0, ops.pack_conv_weight_wino_local(w.to(DEV)), bias.to(DEV), gamma.to(DEV), beta.to(DEV), **kw)
    out = run()
    assert rel_err(to_nchw(out.cpu()), want) < 2e-5
    assert torch.equal(out, run())
