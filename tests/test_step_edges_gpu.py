"""The first and last kernels of a reverse step (round 3): the small-C_in first conv on the unpadded input (conv_first.hip), the
first ResnetBlock's 1x1 res_conv evaluated inside the GroupNorm launch, and GroupNorm + Mish + final 1x1 projection + reverse-step
update in one launch (final_tail_kernel), each through the C ABI against plain torch-CPU fp32 / the oracle on the same seeded inputs.
Reference: models/unet/unet.py:43-50,69-72, models/unet/blocks.py:74-84,103-115, models/diffusion/ddpm.py:203-227."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, to_nchw, to_nhwc
from oracle import diffusion_ref as D
from oracle import philox_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(scope="module")
def ops():
    from ddk import ops as o
    from ddk import lib
    assert lib.load().ddk_device_ok() == 1, lib.last_error()
    return o


def gn_mish(x, gamma, beta, groups=8):
    return F.mish(F.group_norm(x, groups, gamma, beta, 1e-5))


FIRST_CASES = [
    # B, H, W, cin, N
    (2, 16, 16, 1, 32), (2, 16, 16, 3, 64), (3, 16, 8, 8, 128), (2, 32, 32, 2, 96), (1, 16, 24, 5, 160), (2, 16, 16, 7, 256),
    (32, 32, 32, 8, 128),     # cfg4
    (64, 32, 32, 3, 128),     # cfg2
    (16, 32, 32, 1, 128),     # cfg1
    (64, 16, 16, 8, 128),     # cfg3
    (1, 8, 16, 4, 64), (2, 32, 32, 6, 128),
]


@pytest.mark.parametrize("B,H,W,cin,N", FIRST_CASES)
def test_conv_first_and_partials(ops, B, H, W, cin, N):
    """conv_first == F.conv2d(padding=1) on the unpadded input; its partials drive GroupNorm+Mish(+shift) == torch"""
    x = rnd(B, cin, H, W, seed=1 + cin) + 0.3
    w = rnd(N, cin, 3, 3, seed=2 + cin, scale=(cin * 9) ** -0.5)
    b = rnd(N, seed=3)
    ref = F.conv2d(x, w, b, padding=1)
    raw, part, tiles = ops.conv_first(to_nhwc(x).to(DEV), ops.pack_conv_weight_first(w.to(DEV)), b.to(DEV), N)
    assert tiles == H * W // 128
    assert rel_err(to_nchw(raw.cpu()), ref) < 2e-5
    # run-to-run bit stability
    raw2, part2, _ = ops.conv_first(to_nhwc(x).to(DEV), ops.pack_conv_weight_first(w.to(DEV)), b.to(DEV), N)
    assert torch.equal(raw, raw2) and torch.equal(part, part2)
    # partials: {mean, M2} per (128-pixel tile, group) of the output
    cpg = N // 8
    t = to_nhwc(ref).reshape(B * tiles, 128, 8, cpg).permute(0, 2, 1, 3).reshape(B * tiles, 8, 128 * cpg).double()
    mean = t.mean(-1)
    m2 = ((t - mean[..., None]) ** 2).sum(-1)
    assert (part[..., 0].cpu().double() - mean).abs().max() < 2e-5 * ref.abs().max()
    assert rel_err(part[..., 1].cpu(), m2) < 1e-4
    gamma, beta, temb = 1 + 0.1 * rnd(N, seed=5), 0.1 * rnd(N, seed=6), rnd(B, N, seed=7)
    out = ops.groupnorm_mish_from_partials(raw, part, tiles, gamma.to(DEV), beta.to(DEV), temb=temb.to(DEV))
    want = gn_mish(ref, gamma, beta) + temb[:, :, None, None]
    assert rel_err(to_nchw(out.cpu()), want) < 2e-5


def test_conv_first_offset_mean(ops):
    """|mean| >> std: the pairwise {mean, M2} merge does not cancel"""
    B, H, W, cin, N = 2, 16, 16, 8, 128
    x = rnd(B, cin, H, W, seed=11) * 0.01
    w = rnd(N, cin, 3, 3, seed=12, scale=0.1)
    b = torch.full((N,), 100.0)
    ref = F.conv2d(x, w, b, padding=1)
    raw, part, tiles = ops.conv_first(to_nhwc(x).to(DEV), ops.pack_conv_weight_first(w.to(DEV)), b.to(DEV), N)
    gamma, beta = torch.ones(N), torch.zeros(N)
    out = ops.groupnorm_mish_from_partials(raw, part, tiles, gamma.to(DEV), beta.to(DEV))
    want = gn_mish(ref.double(), gamma.double(), beta.double()).float()
    assert rel_err(to_nchw(out.cpu()), want) < 2e-3      # the fp32 input itself carries ~1e-7 * 100 / 0.003 of relative noise


@pytest.mark.parametrize("B,H,W,cin,N", [(2, 16, 16, 3, 64), (32, 32, 32, 8, 128), (4, 16, 16, 1, 128), (2, 16, 16, 8, 256), (2, 16, 32, 5, 32)])
def test_groupnorm_res1x1_addend(ops, B, H, W, cin, N):
    """out = Mish(GN(h)) + res_conv(x): the 1x1 conv of the <= 8-channel input evaluated inside the GroupNorm launch"""
    x = rnd(B, cin, H, W, seed=21)
    h_in = rnd(B, 4, H, W, seed=22)
    w3 = rnd(N, 4, 3, 3, seed=23, scale=(4 * 9) ** -0.5)
    b3 = rnd(N, seed=24)
    wr, br = rnd(N, cin, 1, 1, seed=25, scale=cin ** -0.5), rnd(N, seed=26)
    gamma, beta = 1 + 0.1 * rnd(N, seed=27), 0.1 * rnd(N, seed=28)
    conv = F.conv2d(h_in, w3, b3, padding=1)
    want = gn_mish(conv, gamma, beta) + F.conv2d(x, wr, br)
    raw, part, tiles = ops.conv_first(to_nhwc(h_in).to(DEV), ops.pack_conv_weight_first(w3.to(DEV)), b3.to(DEV), N)
    out = ops.groupnorm_mish_from_partials_res1x1(raw, part, tiles, gamma.to(DEV), beta.to(DEV), to_nhwc(x).to(DEV), wr.to(DEV), br.to(DEV))
    assert rel_err(to_nchw(out.cpu()), want) < 2e-5
    temb = rnd(B, N, seed=29)
    out = ops.groupnorm_mish_from_partials_res1x1(raw, part, tiles, gamma.to(DEV), beta.to(DEV), to_nhwc(x).to(DEV), wr.to(DEV), None,
                                                  temb=temb.to(DEV))
    assert rel_err(to_nchw(out.cpu()), gn_mish(conv, gamma, beta) + temb[:, :, None, None] + F.conv2d(x, wr)) < 2e-5


def _tables():
    buf = D.schedule_buffers("linear", 1000)
    sigma = torch.exp(0.5 * buf["posterior_log_variance_clipped"])
    tb = {k: buf[v].to(DEV) for k, v in (("c_recip", "sqrt_recip_alphas_cumprod"), ("c_recipm1", "sqrt_recipm1_alphas_cumprod"),
                                         ("c1", "posterior_mean_coef1"), ("c2", "posterior_mean_coef2"))}
    tb["sigma"] = sigma.to(DEV)
    return buf, tb


@pytest.mark.parametrize("B,H,W,C,n_out", [(2, 16, 16, 32, 3), (2, 16, 16, 64, 1), (32, 32, 32, 128, 8), (3, 16, 8, 128, 3),
                                           (2, 16, 16, 256, 8), (2, 32, 32, 128, 5)])
def test_final_tail(ops, B, H, W, C, n_out):
    """GroupNorm + Mish + 1x1 projection (+ reverse-step update) in one launch == the separate kernels / torch"""
    h_in = rnd(B, 8, H, W, seed=31)
    w3 = rnd(C, 8, 3, 3, seed=32, scale=(8 * 9) ** -0.5)
    b3 = rnd(C, seed=33)
    gamma, beta = 1 + 0.1 * rnd(C, seed=34), 0.1 * rnd(C, seed=35)
    wf, bf = rnd(n_out, C, 1, 1, seed=36, scale=C ** -0.5), rnd(n_out, seed=37)
    raw, part, tiles = ops.conv_first(to_nhwc(h_in).to(DEV), ops.pack_conv_weight_first(w3.to(DEV)), b3.to(DEV), C)
    eps_ref = F.conv2d(gn_mish(F.conv2d(h_in, w3, b3, padding=1), gamma, beta), wf, bf)
    eps = ops.final_tail(raw, part, tiles, gamma.to(DEV), beta.to(DEV), wf.to(DEV), bf.to(DEV))
    assert rel_err(to_nchw(eps.cpu()), eps_ref) < 2e-5
    # the unfused kernels give the same eps_hat up to summation order
    a1 = ops.groupnorm_mish_from_partials(raw, part, tiles, gamma.to(DEV), beta.to(DEV))
    eps2 = ops.conv1x1_small_n(a1, wf.to(DEV), bf.to(DEV))
    assert rel_err(eps.cpu(), eps2.cpu()) < 5e-6
    # with the update: bit-identical to p_sample_update on the eps_hat the same launch reports (injected noise and Philox)
    buf, tb = _tables()
    x0 = rnd(B, H, W, n_out, seed=38, scale=1.5)
    t = torch.tensor([0, 1, 500, 999] * 8)[:B]
    z = rnd(B, H, W, n_out, seed=39)
    xa = x0.to(DEV).clone()
    eps3 = ops.final_tail(raw, part, tiles, gamma.to(DEV), beta.to(DEV), wf.to(DEV), bf.to(DEV), x=xa, t=t.to(DEV), tables=tb,
                          noise=z.to(DEV))
    assert torch.equal(eps3, eps)
    xb = ops.p_sample_update_(x0.to(DEV).clone(), eps, t.to(DEV), noise=z.to(DEV), **tb)
    assert torch.equal(xa, xb)
    if (H * W * n_out) % 4 == 0:
        seed, stream = 0x1234567887654321, 3
        xa = x0.to(DEV).clone()
        ops.final_tail(raw, part, tiles, gamma.to(DEV), beta.to(DEV), wf.to(DEV), bf.to(DEV), x=xa, t=t.to(DEV), tables=tb, seed=seed,
                       stream_id=stream, want_eps=False)
        xb = ops.p_sample_update_(x0.to(DEV).clone(), eps, t.to(DEV), seed=seed, stream_id=stream, **tb)
        assert torch.equal(xa, xb)


def test_unet_forward_takes_the_fast_edges():
    """The plan's eager forward (first-layer kernel + res1x1 addend + fused tail) against the oracle at full width, C_in = 8 and 3"""
    from helpers import det_state, unet_cfg
    from models import Unet
    from oracle import unet_ref as U
    from utils import synthetic as syn
    for cin in (8, 3):
        cfg = unet_cfg(128, cin)
        net = Unet(cfg)
        sd = det_state({k: v.shape for k, v in net.state_dict().items()})
        net.load_state_dict(sd)
        net = net.to(DEV).eval()
        x = syn.synthetic_normal((2, cin, 32, 32), f"edges.x{cin}")
        t = torch.tensor([3, 977])
        with torch.no_grad():
            y = net(x.to(DEV), t.to(DEV)).cpu()
        ref = U.unet_forward(sd, cfg, x, t)
        assert rel_err(y, ref) < 5e-5


# ---------------------------------------------------------------- GroupNorm finished inside the Winograd conv launch
CLUSTER_CASES = [
    # B, H, W, c0, c1, N: cfg4's eligible Block shapes (32x32: 8 workgroups per image cluster, 2 rounds; 16x16: 2 per cluster)
    (32, 32, 32, 128, 0, 128), (32, 16, 16, 128, 0, 256), (32, 16, 16, 256, 0, 256), (64, 32, 32, 128, 0, 128), (16, 32, 32, 128, 0, 128),
    (64, 16, 16, 128, 0, 128), (64, 16, 16, 64, 64, 128),
]


@pytest.mark.parametrize("B,H,W,c0,c1,N", CLUSTER_CASES)
def test_conv3x3_gn_mish_cluster(ops, B, H, W, c0, c1, N):
    """conv3x3 + GroupNorm + Mish + shift + residual in ONE Winograd launch (tile statistics exchanged between the workgroups of an
    image) == the two-launch path (same statistics, same merge order: <= 1e-6, FMA contraction of the last adds may differ),
    == torch within 2e-5; bit-stable run to run; no cluster ever timed out"""
    cin = c0 + c1
    if ops.L.load().ddk_conv3x3_gn_mish_cluster_ok(B, H, W, cin, N, 8) <= 0:
        pytest.skip("shape not eligible")
    x = rnd(B, cin, H, W, seed=41)
    w = rnd(N, cin, 3, 3, seed=42, scale=(cin * 9) ** -0.5)
    b = rnd(N, seed=43)
    gamma, beta, temb = 1 + 0.1 * rnd(N, seed=44), 0.1 * rnd(N, seed=45), rnd(B, N, seed=46)
    add = rnd(B, N, H, W, seed=47)
    xh = to_nhwc(x).to(DEV)
    x0, x1 = (xh[..., :c0].contiguous(), xh[..., c0:].contiguous()) if c1 else (xh, None)
    wu = ops.pack_conv_weight_wino(w.to(DEV))
    before = ops.cluster_timeouts()
    out = ops.conv3x3_gn_mish_cluster(x0, wu, b.to(DEV), gamma.to(DEV), beta.to(DEV), x2=x1, temb=temb.to(DEV), addend=to_nhwc(add).to(DEV))
    out2 = ops.conv3x3_gn_mish_cluster(x0, wu, b.to(DEV), gamma.to(DEV), beta.to(DEV), x2=x1, temb=temb.to(DEV), addend=to_nhwc(add).to(DEV))
    torch.cuda.synchronize()
    assert ops.cluster_timeouts() == before
    assert torch.equal(out, out2)
    raw, part, tiles = ops.conv_with_gn_partials(x0, ops.pack_conv_weight(w.to(DEV)), b.to(DEV), wu, x2=x1)
    two = ops.groupnorm_mish_from_partials(raw, part, tiles, gamma.to(DEV), beta.to(DEV), temb=temb.to(DEV), addend=to_nhwc(add).to(DEV))
    assert rel_err(out.cpu(), two.cpu()) < 1e-6
    if B <= 32:
        want = gn_mish(F.conv2d(x, w, b, padding=1), gamma, beta) + temb[:, :, None, None] + add
        assert rel_err(to_nchw(out.cpu()), want) < 2e-5
    # no shift, no residual
    out3 = ops.conv3x3_gn_mish_cluster(x0, wu, b.to(DEV), gamma.to(DEV), beta.to(DEV), x2=x1)
    assert rel_err(out3.cpu(), ops.groupnorm_mish_from_partials(raw, part, tiles, gamma.to(DEV), beta.to(DEV)).cpu()) < 1e-6
    # consecutive launches on DIFFERENT inputs through the same records / counters: a stale record would show
    for k in range(3):
        xk = (xh * (k + 2.0)).contiguous()
        xk0, xk1 = (xk[..., :c0].contiguous(), xk[..., c0:].contiguous()) if c1 else (xk, None)
        ok = ops.conv3x3_gn_mish_cluster(xk0, wu, b.to(DEV), gamma.to(DEV), beta.to(DEV), x2=xk1)
        rk, pk, _ = ops.conv_with_gn_partials(xk0, ops.pack_conv_weight(w.to(DEV)), b.to(DEV), wu, x2=xk1)
        assert rel_err(ok.cpu(), ops.groupnorm_mish_from_partials(rk, pk, tiles, gamma.to(DEV), beta.to(DEV)).cpu()) < 1e-6
    assert ops.cluster_timeouts() == before


@pytest.mark.parametrize("B,H,W,c0,c1,N", [(32, 16, 16, 128, 0, 128), (32, 16, 16, 256, 256, 128), (16, 16, 16, 128, 0, 128),
                                           (8, 32, 32, 64, 0, 64)])
def test_conv3x3_gn_mish_cluster_on_k_split_shapes(ops, B, H, W, c0, c1, N):
    """the same ONE launch on shapes whose channel chunks the Winograd conv splits over 2-4 workgroups (cfg4's 16x16 up level): a tile's
    first workgroup sums its partners' partial tiles inside the launch, then the image's tiles exchange statistics as above.
    == conv (slabs + reduce launch) followed by the GroupNorm launch within 2e-6 (the k sum's order differs), == torch within 2e-5;
    bit-stable run to run; counters re-arm; nothing timed out"""
    cin = c0 + c1
    lib = ops.L.load()
    if lib.ddk_conv3x3_gn_mish_cluster_ok(B, H, W, cin, N, 8) <= 0:
        pytest.skip("shape / device not eligible")
    assert lib.ddk_conv_wino_splits(B, H, W, cin, N) > 1 and lib.ddk_conv3x3_gn_mish_cluster_split_workspace_bytes(B, H, W, cin, N, 8) > 0
    x = rnd(B, cin, H, W, seed=61)
    w = rnd(N, cin, 3, 3, seed=62, scale=(cin * 9) ** -0.5)
    b = rnd(N, seed=63)
    gamma, beta, temb = 1 + 0.1 * rnd(N, seed=64), 0.1 * rnd(N, seed=65), rnd(B, N, seed=66)
    add = rnd(B, N, H, W, seed=67)
    xh = to_nhwc(x).to(DEV)
    x0, x1 = (xh[..., :c0].contiguous(), xh[..., c0:].contiguous()) if c1 else (xh, None)
    wu, wp = ops.pack_conv_weight_wino(w.to(DEV)), ops.pack_conv_weight(w.to(DEV))
    g, be, te, ad = gamma.to(DEV), beta.to(DEV), temb.to(DEV), to_nhwc(add).to(DEV)
    before = ops.cluster_timeouts()
    out = ops.conv3x3_gn_mish_cluster(x0, wu, b.to(DEV), g, be, x2=x1, temb=te, addend=ad)
    out2 = ops.conv3x3_gn_mish_cluster(x0, wu, b.to(DEV), g, be, x2=x1, temb=te, addend=ad)
    torch.cuda.synchronize()
    assert ops.cluster_timeouts() == before
    assert torch.equal(out, out2)
    raw = ops.conv(ops.CONV3X3_S1, x0, wp, b.to(DEV), x2=x1, w_wino=wu)
    two = ops.groupnorm_mish(raw, g, be, temb=te, addend=ad)
    assert rel_err(out.cpu(), two.cpu()) < 2e-6
    want = gn_mish(F.conv2d(x, w, b, padding=1), gamma, beta) + temb[:, :, None, None] + add
    assert rel_err(to_nchw(out.cpu()), want) < 2e-5
    for k in range(3):        # different inputs through the same pair counters / slabs / records: a stale partial tile would show
        xk = (xh * (k + 2.0)).contiguous()
        xk0, xk1 = (xk[..., :c0].contiguous(), xk[..., c0:].contiguous()) if c1 else (xk, None)
        ok = ops.conv3x3_gn_mish_cluster(xk0, wu, b.to(DEV), g, be, x2=xk1)
        rk = ops.conv(ops.CONV3X3_S1, xk0, wp, b.to(DEV), x2=xk1, w_wino=wu)
        assert rel_err(ok.cpu(), ops.groupnorm_mish(rk, g, be).cpu()) < 2e-6
    assert ops.cluster_timeouts() == before


def test_cluster_groupnorm_option_gives_identical_unet():
    """The plan with GroupNorm finished inside the conv launches == the plan with the conv + GroupNorm-apply pairs (<= 2e-5 of the
    output's max: the two differ in FMA contraction of the residual add only), and is bit-stable run to run"""
    from helpers import det_state, unet_cfg
    from models import Unet
    from utils import synthetic as syn
    cfg = unet_cfg(128, 8)
    net = Unet(cfg)
    net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
    net = net.to(DEV).eval()
    x = syn.synthetic_normal((32, 8, 32, 32), "cluster.x").to(DEV)
    t = torch.arange(32, device=DEV) * 31
    with torch.no_grad():
        plan = net.plan()
        y_default = net(x, t)                                 # option value 1: single forwards keep the conv + apply pairs
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 2)        # 2: single forwards take the in-launch path too (and check it)
        y_on = net(x, t)
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 0)
        y_off = net(x, t)
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 2)
        y_on2 = net(x, t)
    assert plan.cluster_timeouts() == 0
    assert plan._cluster == 2, "a give-up would have switched the option off"
    assert torch.equal(y_on, y_on2)
    assert torch.equal(y_default, y_off)
    # (same statistics, same merge order, same finishing arithmetic: with the round-4 kernels the two paths agree bit for bit)
    assert rel_err(y_on.cpu(), y_off.cpu()) < 2e-5


@pytest.mark.parametrize("C,N", [(256, 256), (128, 128)])
def test_cluster_groupnorm_next_to_a_foreign_kernel_is_never_silently_wrong(ops, C, N):
    """A long-running kernel on ANOTHER stream holds most CUs (through its LDS footprint) while cluster launches run: the
    co-residency the exchange assumes does not hold.  Allowed outcomes: the correct tensor, or a raised error (the bounded
    wait gave up: NaN tiles + sticky counter) -- never a finite wrong tensor.  And the counters re-arm: the next launch on an
    idle GPU is correct again.  (128 -> 128: the k-split form, whose first workgroups also wait for their partners' partial tiles.)"""
    B, H, W = 32, 16, 16
    if ops.L.load().ddk_conv3x3_gn_mish_cluster_ok(B, H, W, C, N, 8) <= 0:
        pytest.skip("device / shape not eligible")
    x = to_nhwc(rnd(B, C, H, W, seed=51)).to(DEV)
    w = rnd(N, C, 3, 3, seed=52, scale=(C * 9) ** -0.5).to(DEV)
    b, gamma, beta = rnd(N, seed=53).to(DEV), (1 + 0.1 * rnd(N, seed=54)).to(DEV), (0.1 * rnd(N, seed=55)).to(DEV)
    wu = ops.pack_conv_weight_wino(w)
    if ops.L.load().ddk_conv_wino_splits(B, H, W, C, N) > 1:
        want = ops.groupnorm_mish(ops.conv(ops.CONV3X3_S1, x, ops.pack_conv_weight(w), b, w_wino=wu), gamma, beta)
    else:
        raw, part, tiles = ops.conv_with_gn_partials(x, ops.pack_conv_weight(w), b, wu)
        want = ops.groupnorm_mish_from_partials(raw, part, tiles, gamma, beta)
    torch.cuda.synchronize()
    lib = ops.L.load()
    side = torch.cuda.Stream(device=DEV)
    outcomes = []
    for filler_wgs in (224, 252):
        # 100 KB of LDS per filler workgroup: a CU that hosts one cannot take a 144 KB conv workgroup; 60 ms > the 20 ms wait bound
        ops.L.check(lib.ddk_debug_occupy(filler_wgs, 100 * 1024, 60000, side.cuda_stream), "debug_occupy")
        try:
            out = ops.conv3x3_gn_mish_cluster(x, wu, b, gamma, beta)           # check=True: waits, raises on a give-up
            assert rel_err(out.cpu(), want.cpu()) < 2e-6
            outcomes.append("correct")
        except ops.L.DDKError as e:
            assert "gave up" in str(e)
            outcomes.append("raised")
        side.synchronize()
    # idle again: the same records / counters serve a correct launch (they re-armed, or the wrapper zeroed them)
    out = ops.conv3x3_gn_mish_cluster(x, wu, b, gamma, beta)
    assert rel_err(out.cpu(), want.cpu()) < 2e-6
    print("outcomes next to the foreign kernel:", outcomes)


def test_sampler_reruns_without_cluster_groupnorm_after_a_give_up(ops):
    """plan.sample_nhwc next to a foreign kernel: whatever happened inside, the chain it returns equals the chain of a plan
    that never used the in-launch GroupNorm (<= 1e-5: FMA contraction of the last adds), and a give-up switched the option off
    with a RuntimeWarning instead of returning NaN / wrong images."""
    import warnings
    from helpers import det_state, unet_cfg
    from models import DDPM, Unet
    cfg = dict(unet_cfg(128, 8), image_size=32, T=1000, loss_type="simple", beta_schedule="linear", loss_flat="sum")
    model = DDPM(cfg, Unet(cfg), "cuda", 8)
    model.load_state_dict(det_state({k: v.shape for k, v in model.state_dict().items()}), strict=False)
    model = model.to(DEV).eval()
    plan = model.latent_model.plan()
    tables = model._tables()
    x0 = ops.randn((32, 32, 32, 8), DEV, seed=7, step=1000, stream_id=0)
    with torch.no_grad():
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 0)
        ref = plan.sample_nhwc(x0.clone(), tables, 999, 994, seed=7).clone()
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 1)
        side = torch.cuda.Stream(device=DEV)
        ops.L.check(ops.L.load().ddk_debug_occupy(240, 100 * 1024, 150000, side.cuda_stream), "debug_occupy")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = plan.sample_nhwc(x0.clone(), tables, 999, 994, seed=7).clone()
        side.synchronize()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 1e-4
    gave_up = any("gave up" in str(w.message) for w in caught)
    assert (plan._cluster == 0) == gave_up
    print("sampler next to the foreign kernel:", "gave up and reran" if gave_up else "finished with the in-launch GroupNorm")


# ---- weights-stationary 1x1 conv (conv1x1_ws.hip): blocks.py:103 (res_conv), :123 (to_qkv behind the channel LayerNorm), :124 + :13-14
WS_CASES = [
    # B, H, W, N, bias, resid, ln
    (32, 32, 32, 384, False, False, True),    # to_qkv at 32x32 (cfg4): 512 tiles over 3 x 85 workgroups
    (32, 32, 32, 128, True, True, False),     # to_out + Residual at 32x32
    (32, 16, 16, 256, True, False, False),    # res_conv 128 -> 256 at 16x16: one tile per workgroup
    (32, 16, 16, 384, False, False, True),    # to_qkv at 16x16 (up path): ragged tile counts per workgroup (128 tiles, 85 workgroups)
    (2, 32, 32, 128, False, True, True),      # the minimum size: 32 tiles, every option at once
    (3, 32, 32, 256, True, True, True),
    (5, 16, 32, 128, False, False, False),    # 40 tiles
]


@pytest.mark.parametrize("B,H,W,N,use_bias,use_resid,use_ln", WS_CASES)
def test_conv1x1_ws(ops, B, H, W, N, use_bias, use_resid, use_ln):
    """ddk_conv1x1_ws == F.conv2d(k=1) (+ bias, + residual), with the channel LayerNorm folded in == conv(LayerNorm(x)); and the
    generic conv entry takes the same kernel on full-chip shapes (bit-identical)"""
    K = 128
    x = rnd(B, K, H, W, seed=11) * 1.5 + 0.4
    w = rnd(N, K, 1, 1, seed=12, scale=K ** -0.5)
    b = rnd(N, seed=13) if use_bias else None
    r = rnd(B, N, H, W, seed=14) if use_resid else None
    g, be = 1 + 0.2 * rnd(K, seed=15), 0.1 * rnd(K, seed=16)
    xin = x
    if use_ln:
        std = x.var(dim=1, unbiased=False, keepdim=True).sqrt()
        xin = (x - x.mean(dim=1, keepdim=True)) / (std + 1e-5) * g.view(1, K, 1, 1) + be.view(1, K, 1, 1)
    ref = F.conv2d(xin.double(), w.double(), b.double() if use_bias else None)
    if use_resid:
        ref = ref + r.double()
    w2 = w.reshape(N, K)
    ln = None
    if use_ln:
        ln = ((w2 @ g).to(DEV).contiguous(), (w2 @ be).to(DEV).contiguous())
        w2 = w2 * g.view(1, K)
    xd = to_nhwc(x).to(DEV)
    rd = to_nhwc(r).to(DEV) if use_resid else None
    bd = b.to(DEV) if use_bias else None
    out = ops.conv1x1_ws(xd, w2.contiguous().to(DEV), bd, rd, ln)
    assert rel_err(to_nchw(out.cpu()).double(), ref) < 2e-5
    out2 = ops.conv1x1_ws(xd, w2.contiguous().to(DEV), bd, rd, ln)
    assert torch.equal(out, out2)
    if not use_ln:
        # the generic entry takes this kernel from 256 (tile, slice) pairs on (a full chip), the tile kernel below that
        y = ops.conv(ops.CONV1X1, xd, ops.pack_conv_weight(w.to(DEV)), bd, resid=rd)
        if (B * H * W // 64) * (N // 128) >= 256:
            assert torch.equal(y, out)
        else:
            assert rel_err(y.cpu(), out.cpu()) < 2e-6


def test_conv1x1_ws_eligibility(ops):
    from ddk import lib
    ok = lib.load().ddk_conv1x1_ws_ok
    assert ok(32768, 128, 384) and ok(2048, 128, 128)
    assert not ok(32768, 256, 384) and not ok(32768, 128, 96) and not ok(1024, 128, 128) and not ok(2048 + 32, 128, 128)
    with pytest.raises(lib.DDKError):
        ops.conv1x1_ws(torch.zeros(1, 8, 8, 128, device=DEV), torch.zeros(128, 128, device=DEV))


# ---- folded attention (attention.hip attn_fold_kernel + conv1x1_ws per-image weights): blocks.py:8-14,57-71,116-134
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (32, 32, 32), (4, 16, 32), (1, 64, 64)])
def test_attention_folded_vs_torch(ops, B, H, W):
    """Residual(PreNorm(LinearAttention)) with q projection + apply + to_out folded into one per-image 128x128 conv == the
    reference formulation in torch (LayerNorm -> to_qkv -> softmax(k) -> two einsums -> to_out -> + x)"""
    C, heads = 128, 4
    x = rnd(B, C, H, W, seed=81) * 1.3 + 0.2
    wq = rnd(3 * C, C, 1, 1, seed=82, scale=C ** -0.5)
    wo, bo = rnd(C, C, 1, 1, seed=83, scale=C ** -0.5), rnd(C, seed=84, scale=0.1)
    g, be = 1 + 0.2 * rnd(C, seed=85), 0.1 * rnd(C, seed=86)
    xd = x.double()
    std = xd.var(dim=1, unbiased=False, keepdim=True).sqrt()
    xn = (xd - xd.mean(dim=1, keepdim=True)) / (std + 1e-5) * g.double().view(1, C, 1, 1) + be.double().view(1, C, 1, 1)
    q, k, v = F.conv2d(xn, wq.double()).reshape(B, 3, heads, 32, H * W).unbind(1)
    ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
    o = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, C, H, W)
    ref = F.conv2d(o, wo.double(), bo.double()) + xd
    out = ops.attention_folded(to_nhwc(x).to(DEV), wq.to(DEV), g.to(DEV), be.to(DEV), wo.to(DEV), bo.to(DEV))
    assert rel_err(to_nchw(out.cpu()).double(), ref) < 2e-5
    out2 = ops.attention_folded(to_nhwc(x).to(DEV), wq.to(DEV), g.to(DEV), be.to(DEV), wo.to(DEV), bo.to(DEV))
    assert torch.equal(out, out2)


def test_attention_fold_option_gives_the_same_unet():
    """plan option DDK_OPT_ATTENTION_FOLD: the folded attention block at the 32x32 level == to_qkv / context / apply / to_out
    (<= 2e-5 of the output's max), bit-stable"""
    from helpers import det_state, unet_cfg
    from models import Unet
    from utils import synthetic as syn
    cfg = unet_cfg(128, 8)
    net = Unet(cfg)
    net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
    net = net.to(DEV).eval()
    x = syn.synthetic_normal((32, 8, 32, 32), "fold.x").to(DEV)
    t = torch.arange(32, device=DEV) * 29
    with torch.no_grad():
        y_on = net(x, t)
        plan = net.plan()
        plan.set_option(plan.OPT_ATTENTION_FOLD, 0)
        y_off = net(x, t)
        plan.set_option(plan.OPT_ATTENTION_FOLD, 1)
        y_on2 = net(x, t)
    assert torch.equal(y_on, y_on2)
    assert not torch.equal(y_on, y_off)          # the two formulations round differently: the option really switches paths
    assert rel_err(y_on.cpu(), y_off.cpu()) < 2e-5


def lib_ok(ops, M, K, N):
    return bool(ops.L.load().ddk_conv1x1_ws_ok(M, K, N))


@pytest.mark.parametrize("B,H,W,shift", [(2, 32, 32, 0.0), (32, 32, 32, 0.0), (3, 16, 32, 0.0), (1, 64, 64, 0.0), (5, 16, 16, 0.0), (2, 32, 32, 30.0)])
def test_attention_kv_projection_and_context_in_one_launch(ops, B, H, W, shift):
    """ddk_attention_kv_context (attn_kvctx_kernel + the split merge): ctx = softmax_pixels(k) v^T with k, v = the k / v thirds of
    to_qkv(LayerNorm(x)) (blocks.py:57-60,123,129-131) against torch in fp64, against the two-launch path (1x1 conv that writes kv +
    linattn_context_kv), and bit-stable.  shift != 0: k columns whose maximum moves from tile to tile by tens (the online rescale of
    the accumulated context is exercised: a per-pixel ramp is added to x along one channel direction)."""
    C, heads = 128, 4
    x = rnd(B, C, H, W, seed=181) * 1.3 + 0.2
    if shift:
        ramp = torch.linspace(0, 1, H * W).reshape(1, 1, H, W)
        x = x + shift * ramp * rnd(1, C, 1, 1, seed=187)
    wq = rnd(3 * C, C, 1, 1, seed=182, scale=C ** -0.5 * (3.0 if shift else 1.0))
    g, be = 1 + 0.2 * rnd(C, seed=185), 0.1 * rnd(C, seed=186)
    xd = x.double()
    std = xd.var(dim=1, unbiased=False, keepdim=True).sqrt()
    xn = (xd - xd.mean(dim=1, keepdim=True)) / (std + 1e-5) * g.double().view(1, C, 1, 1) + be.double().view(1, C, 1, 1)
    q, k, v = F.conv2d(xn, wq.double()).reshape(B, 3, heads, 32, H * W).unbind(1)
    ref = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
    xh = to_nhwc(x).to(DEV)
    ctx = ops.attention_kv_context(xh, wq.to(DEV), g.to(DEV), be.to(DEV))
    assert rel_err(ctx.cpu().double(), ref) < 2e-5
    assert torch.equal(ctx, ops.attention_kv_context(xh, wq.to(DEV), g.to(DEV), be.to(DEV)))
    # the two-launch path it replaces in the plan (the streaming 1x1 kernel takes M >= 2048 pixels)
    hc = heads * 32
    if not lib_ok(ops, B * H * W, C, 2 * hc):
        return
    wqf = wq.reshape(3 * hc, C)
    wg = (wqf * g.view(1, C)).to(DEV)
    c1, c2 = (wqf @ g).to(DEV), (wqf @ be).to(DEV)
    lib = ops.L.load()
    kv = torch.empty((B, H, W, 2 * hc), device=DEV)
    ops.L.check(lib.ddk_conv1x1_ws(ops.L.ptr(xh), ops.L.ptr(wg[hc:].contiguous()), None, None, ops.L.ptr(kv), B * H * W, 2 * hc,
                                   ops.L.ptr(c1[hc:].contiguous()), ops.L.ptr(c2[hc:].contiguous()), 1e-5, ops.L.stream()), "conv1x1_ws(kv)")
    two = torch.empty((B, heads, 32, 32), device=DEV)
    nbytes = lib.ddk_linattn_context_workspace_bytes(B, H * W, heads)
    ws = torch.empty(max(nbytes, 16) // 4, device=DEV)
    ops.L.check(lib.ddk_linattn_context_kv(ops.L.ptr(kv), ops.L.ptr(two), B, H * W, heads, ops.L.ptr(ws), nbytes, ops.L.stream()), "context_kv")
    assert rel_err(ctx.cpu(), two.cpu()) < 2e-5


def test_attention_kv_context_option_gives_the_same_unet():
    """plan option DDK_OPT_ATTENTION_KV_CONTEXT: the folded attention block of the 32x32 level with the k, v projection + context as one
    launch == as a 1x1 conv + the context kernel (<= 2e-5 of the output's max; the two round differently), bit-stable"""
    from helpers import det_state, unet_cfg
    from models import Unet
    from utils import synthetic as syn
    cfg = unet_cfg(128, 8)
    net = Unet(cfg)
    net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
    net = net.to(DEV).eval()
    x = syn.synthetic_normal((32, 8, 32, 32), "kvctx.x").to(DEV)
    t = torch.arange(32, device=DEV) * 29
    with torch.no_grad():
        y_on = net(x, t)
        plan = net.plan()
        plan.set_option(plan.OPT_ATTENTION_KV_CONTEXT, 0)
        y_off = net(x, t)
        plan.set_option(plan.OPT_ATTENTION_KV_CONTEXT, 1)
        y_on2 = net(x, t)
    assert torch.equal(y_on, y_on2)
    assert not torch.equal(y_on, y_off)
    assert rel_err(y_on.cpu(), y_off.cpu()) < 2e-5


def test_downsample_reduce_fold_option_gives_the_same_unet_bits():
    """plan option DDK_OPT_FOLD_DOWNSAMPLE_REDUCE: at batch 32 the Downsample convs of the 16x16 -> 8x8 and 8x8 -> 4x4 transitions split k;
    with the option on their slabs are summed by the image-local ResnetBlock behind them (two launches less per forward), with it off by
    the reduce launch -- the same sums in the same order, so the network output is bit-identical (blocks.py:41-47, unet.py:83-90)."""
    from helpers import det_state, unet_cfg
    from models import Unet
    from utils import synthetic as syn
    cfg = unet_cfg(128, 8)
    net = Unet(cfg)
    net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
    net = net.to(DEV).eval()
    x = syn.synthetic_normal((32, 8, 32, 32), "downfold.x").to(DEV)
    t = torch.arange(32, device=DEV) * 31
    from ddk import lib as L
    lib = L.load()
    assert lib.ddk_conv_splits(1, 32, 16, 16, 256, 256) > 1 and lib.ddk_conv_splits(1, 32, 8, 8, 256, 256) > 1   # kind 1 = 3x3 stride 2
    with torch.no_grad():
        plan = net.plan()
        plan.set_option(plan.OPT_FOLD_DOWNSAMPLE_REDUCE, 1)
        y_on = net(x, t)
        plan.set_option(plan.OPT_FOLD_DOWNSAMPLE_REDUCE, 0)        # the default: measured faster at batch 32 (tools/fold_ab.py)
        y_off = net(x, t)
        plan.set_option(plan.OPT_FOLD_DOWNSAMPLE_REDUCE, 1)
        y_on2 = net(x, t)
    assert torch.equal(y_on, y_off) and torch.equal(y_on, y_on2)


def _chain_net(in_ch=8):
    from helpers import det_state, unet_cfg
    from models import Unet
    cfg = unet_cfg(128, in_ch)
    net = Unet(cfg)
    net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
    return net.to(DEV).eval()


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("batch", [32, 5, 40])
def test_level_chain_option_gives_the_same_unet(batch, mode):
    """plan option DDK_OPT_LEVEL_CHAIN (csrc/level_chain.hip): the 4x4 level of the full-width UNet -- 2 + 2 + 2 ResnetBlocks and three
    attention blocks, reference unet.py:83-101 -- as ONE persistent launch whose workgroups hand the images to each other, against the
    same level as 19 launches (mode 2; mode 1, the default: with the Downsample conv in front and the Upsample transpose conv behind
    inside the launch as well, 23 launches); the 8x8 levels -- downs[2], ups[1] -- as one launch each against 7 + 9 (mode 4;
    8 / 16: one of them); all three (mode 3).  Same arithmetic up to summation order (<= 2e-5 of the output's max), not the same bits (the option
    really switches paths), bit-stable from launch to launch (a stale hand-off would show), no wait timed out.  Batch 5: fewer
    workgroups than CUs; batch 40: images walked in two rounds by the same workgroups."""
    from ddk import ops
    from utils import synthetic as syn
    net = _chain_net()
    x = syn.synthetic_normal((batch, 8, 32, 32), f"chain.x{batch}").to(DEV)
    x2 = syn.synthetic_normal((batch, 8, 32, 32), f"chain.y{batch}").to(DEV)
    t = (torch.arange(batch, device=DEV) * 23) % 1000
    with torch.no_grad():
        plan = net.plan()
        y_plain = net(x, t)                                        # single forwards take neither in-launch path by default
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 2)             # ... with 2 they take both (and wait + check behind the call)
        plan.set_option(plan.OPT_LEVEL_CHAIN, mode)
        plan.set_option(10, 1 << 20)                               # diagnostic: the chain at any batch (by default only up to one round of 32 images)
        before = ops.cluster_timeouts()
        y_on = net(x, t)
        y_other = net(x2, t)                                       # different data through the same hand-off buffers
        y_on2 = net(x, t)
        plan.set_option(plan.OPT_LEVEL_CHAIN, 0)
        y_off = net(x, t)
        plan.set_option(plan.OPT_LEVEL_CHAIN, mode)
        y_on3 = net(x, t)
    assert plan._cluster == 2, "the in-launch paths were switched off by a failed check"
    assert ops.cluster_timeouts() == before
    assert torch.isfinite(y_on).all()
    assert torch.equal(y_on, y_on2) and torch.equal(y_on, y_on3)
    assert not torch.equal(y_on, y_other)
    assert not torch.equal(y_on, y_off)
    assert rel_err(y_on.cpu(), y_off.cpu()) < 2e-5
    assert rel_err(y_on.cpu(), y_plain.cpu()) < 2e-5


def test_level_chain_vs_oracle():
    """the chained level inside the whole UNet against the CPU oracle (oracle/unet_ref.py, pinned to the reference by tests/golden):
    in_ch 3 and 8 (cfg2 / cfg4 shapes), 1e-3 relative as BASELINE.json's north_star states it, measured ~5e-6"""
    from oracle import unet_ref as U
    from helpers import unet_cfg
    from utils import synthetic as syn
    for in_ch in (3, 8):
        net = _chain_net(in_ch)
        cfg = unet_cfg(128, in_ch)
        x = syn.synthetic_normal((4, in_ch, 32, 32), f"chain.oracle{in_ch}")
        t = torch.tensor([0, 17, 500, 999])
        with torch.no_grad():
            plan = net.plan()
            plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 2)
            y = net(x.to(DEV), t.to(DEV)).cpu()
            ref = U.unet_forward({k: v.cpu() for k, v in net.state_dict().items()}, cfg, x, t)
        assert plan._cluster == 2
        assert rel_err(y, ref) < 5e-5


def test_level_chain_in_the_sampler_matches_the_unchained_chain():
    """ddk_sampler_run (hipGraph replay, 16 steps per graph) with the level chain on (the default) vs off: 40 reverse steps from the
    same x_T with the same Philox stream end within 1e-4 of each other, and the counters re-arm across replays (no timeout)."""
    from ddk import ops
    from models import DDPM
    from helpers import ddpm_cfg
    from utils import synthetic as syn
    cfg = ddpm_cfg(128, 3, 32, T=1000)
    from models import Unet
    model = DDPM(cfg, Unet(cfg), "cuda", 3)
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    model = model.to(DEV).eval()
    plan = model.latent_model.plan()
    tables = model._tables()
    outs = {}
    before = ops.cluster_timeouts()
    for chain in (1, 0):
        plan.set_option(plan.OPT_LEVEL_CHAIN, chain)
        x = ops.randn((8, 32, 32, 3), DEV, seed=77, step=1000, stream_id=0)
        with torch.no_grad():
            plan.sample_nhwc(x, tables, 999, 960, seed=77, stream_id=0, use_graph=True)
        outs[chain] = x.clone()
    plan.set_option(plan.OPT_LEVEL_CHAIN, 1)
    assert plan._cluster >= 1 and ops.cluster_timeouts() == before
    assert torch.isfinite(outs[1]).all()
    assert not torch.equal(outs[0], outs[1])
    assert float((outs[0] - outs[1]).abs().max()) < 1e-4 * float(outs[0].abs().max())


@pytest.mark.parametrize("in_ch,batch", [(8, 32), (3, 5), (1, 40)])
def test_first_block_groupnorm_in_launch_option_gives_the_same_unet(in_ch, batch):
    """plan option DDK_OPT_FIRST_GROUPNORM (conv_first_kernel<C, true>): the first Block's GroupNorm + Mish + time shift finished inside
    the conv's launch (the eight 128-pixel tiles of an image exchange their statistics) against conv + GroupNorm-apply: <= 2e-5 of
    the output's max, not the same bits, bit-stable across launches with other data in between, no wait timed out (blocks.py:74-84)"""
    from ddk import ops
    from utils import synthetic as syn
    net = _chain_net(in_ch)
    x = syn.synthetic_normal((batch, in_ch, 32, 32), f"firstgn.x{batch}").to(DEV)
    x2 = syn.synthetic_normal((batch, in_ch, 32, 32), f"firstgn.y{batch}").to(DEV)
    t = (torch.arange(batch, device=DEV) * 37) % 1000
    with torch.no_grad():
        plan = net.plan()
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 2)
        before = ops.cluster_timeouts()
        y_on = net(x, t)
        y_other = net(x2, t)
        y_on2 = net(x, t)
        plan.set_option(plan.OPT_FIRST_GROUPNORM, 0)
        y_off = net(x, t)
        plan.set_option(plan.OPT_FIRST_GROUPNORM, 1)
    assert plan._cluster == 2 and ops.cluster_timeouts() == before
    assert torch.isfinite(y_on).all() and torch.equal(y_on, y_on2) and not torch.equal(y_on, y_other)
    assert rel_err(y_on.cpu(), y_off.cpu()) < 2e-5
