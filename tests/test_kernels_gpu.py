"""Per-kernel parity: every HIP kernel family (called through the C ABI via ddk.ops) against the CPU oracle /
plain torch-CPU fp32 on the same seeded inputs.  fp32 tolerance: max|diff| / max|ref| <= 2e-5 for contractions
(summation order differs), tighter for elementwise; bit-exact where the kernel pins the evaluation order."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, to_nchw, to_nhwc
from oracle import diffusion_ref as D
from oracle import philox_ref
from oracle import unet_ref as U

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


# ---------------------------------------------------------------- conv family
CONV_CASES = [
    # kind, B, H, W, c0, c1, N
    ("s1", 2, 8, 8, 32, 0, 64),          # small-M: split-K path, 64x64 tile
    ("s1", 2, 32, 32, 128, 0, 128),      # split-K on 128-wide
    ("s1", 32, 32, 32, 128, 0, 128),     # cfg4 level-0 conv: 128x128 tile, direct epilogue
    ("s1", 32, 16, 16, 128, 0, 256),     # 128x64 tile
    ("s1", 4, 8, 8, 256, 256, 256),      # dual source (concat-free skip), K = 4608
    ("s1", 3, 5, 7, 32, 0, 32),          # ragged M, N = 32 tile
    ("s1", 2, 16, 16, 64, 0, 64),        # N = 64 tile
    # halo-tile kernel (>= 208 workgroups after splitting the channel chunks, >= 4 chunks each)
    ("s1", 32, 16, 16, 256, 256, 256),   # dual source, channel-chunk split
    ("s1", 69, 8, 8, 256, 0, 384),       # two images per tile, odd batch: last tile half empty
    ("s1", 64, 16, 16, 128, 0, 160),     # ragged N tile
    ("s1", 52, 8, 32, 256, 0, 128),      # 4-row tiles, two tiles per image
    ("s1", 32, 32, 32, 160, 0, 128),     # odd chunk count (5): halo double-buffer parity
    # 32 -> 32 channel resampler convs (128x32 im2col tile) at the decoder's resolutions
    ("s1", 32, 16, 16, 32, 0, 32), ("s1", 9, 32, 32, 32, 0, 32), ("s1", 3, 64, 64, 32, 0, 32), ("s1", 1, 128, 128, 32, 0, 32),
    ("s1", 1, 40, 256, 32, 0, 32),
    ("s2", 4, 16, 16, 64, 0, 64),
    ("s2", 32, 32, 32, 128, 0, 128),
    ("s2", 2, 7, 9, 32, 0, 32),          # odd spatial size
    ("1x1", 8, 16, 16, 128, 0, 384),     # to_qkv shape
    ("1x1", 2, 4, 4, 256, 256, 256),     # res_conv over a dual source
    ("T", 4, 4, 4, 64, 0, 64),
    ("T", 32, 16, 16, 128, 0, 128),      # ups.2.3 at cfg4
]


@pytest.mark.parametrize("kind,B,H,W,c0,c1,N", CONV_CASES)
def test_conv(ops, kind, B, H, W, c0, c1, N):
    cin = c0 + c1
    x = rnd(B, cin, H, W, seed=1)
    k = {"s1": 3, "s2": 3, "1x1": 1, "T": 4}[kind]
    bias = rnd(N, seed=3, scale=0.1)
    if kind == "T":
        w = rnd(cin, N, 4, 4, seed=2, scale=(cin * 4) ** -0.5)
        ref = F.conv_transpose2d(x, w, bias, stride=2, padding=1)
        wp = ops.pack_convT_weight(w.to(DEV))
        code = ops.CONVT4X4_S2
    else:
        w = rnd(N, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
        ref = F.conv2d(x, w, bias, stride=2 if kind == "s2" else 1, padding=k // 2)
        wp = ops.pack_conv_weight(w.to(DEV))
        code = {"s1": ops.CONV3X3_S1, "s2": ops.CONV3X3_S2, "1x1": ops.CONV1X1}[kind]
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    resid = rnd(*ref.shape, seed=4)
    out = ops.conv(code, x0, wp, bias.to(DEV), x2=x1)
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5
    out_r = ops.conv(code, x0, wp, bias.to(DEV), x2=x1, resid=to_nhwc(resid).to(DEV))
    assert rel_err(to_nchw(out_r.cpu()), ref + resid) < 2e-5
    # run-to-run determinism (fixed-order split-K reduction, no float atomics)
    assert torch.equal(out, ops.conv(code, x0, wp, bias.to(DEV), x2=x1))


CONVT_WINO_CASES = [  # B, H, W, C, N  (ConvTranspose2d 4x4 s2 p1 as Winograd F(2x2,2x2) per phase: even H, W; C % 32 == 0; N % 128 == 0)
    (32, 16, 16, 128, 128),        # ups.2.3 at cfg4: 256 workgroups, no split
    (32, 8, 8, 256, 256),          # ups.1.3: two channel-chunk slabs + reduce
    (32, 4, 4, 256, 256),          # ups.0.3: eight slabs of one chunk
    (3, 6, 10, 96, 128),           # ragged tile block (45 tiles), odd chunk count
    (1, 2, 2, 32, 128),            # one tile: every patch pixel but the centre is padding
    (2, 34, 18, 64, 256),          # borders on non-square maps, two n tiles
    (8, 64, 64, 128, 128),         # many rounds (1024 tile blocks x 4 phases)
]


@pytest.mark.parametrize("B,H,W,C,N", CONVT_WINO_CASES)
def test_conv_transpose_winograd(ops, B, H, W, C, N):
    """reference models/unet/blocks.py:32-39 (Upsample = ConvTranspose2d(dim, dim, 4, 2, 1)): the Winograd F(2x2, 2x2)-per-phase kernel
    == F.conv_transpose2d (2e-5 of the tensor's max, the bar of the direct kernels), == the direct im2col kernel to the same bar,
    bit-stable run to run"""
    x = rnd(B, C, H, W, seed=11)
    w = rnd(C, N, 4, 4, seed=12, scale=(C * 4) ** -0.5)
    bias = rnd(N, seed=13, scale=0.1)
    ref = F.conv_transpose2d(x, w, bias, stride=2, padding=1)
    wd = w.to(DEV)
    wp, wu = ops.pack_convT_weight(wd), ops.pack_convT_weight_wino(wd)
    assert ops.L.load().ddk_convT_wino_splits(B, H, W, C, N) > 0
    xh = to_nhwc(x).to(DEV)
    out = ops.conv(ops.CONVT4X4_S2, xh, wp, bias.to(DEV), w_wino=wu)
    assert out.shape == (B, 2 * H, 2 * W, N)
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5
    direct = ops.conv(ops.CONVT4X4_S2, xh, wp, bias.to(DEV))
    assert rel_err(out.cpu(), direct.cpu()) < 2e-5
    assert not torch.equal(out, direct), "the Winograd form really ran (it sums in another order)"
    assert torch.equal(out, ops.conv(ops.CONVT4X4_S2, xh, wp, bias.to(DEV), w_wino=wu))
    out_nb = ops.conv(ops.CONVT4X4_S2, xh, wp, None, w_wino=wu)
    assert rel_err(to_nchw(out_nb.cpu()), ref - bias[None, :, None, None]) < 2e-5


WINO_CASES = [  # B, H, W, c0, c1, N  (Winograd F(2x2,3x3): even H, W; cin % 32 == 0; N % 64 == 0)
    (2, 8, 8, 32, 0, 64),          # one chunk, 32 tiles exactly
    (2, 4, 4, 64, 0, 64),          # 8 tiles < 32: ragged tile block, channel-chunk splits
    (3, 6, 10, 96, 0, 128),        # odd tile counts (45 tiles), 3 chunks
    (2, 16, 16, 128, 128, 128),    # concat of two sources (unet.py:97), 8 chunks
    (32, 4, 4, 256, 0, 256),       # cfg4 4x4 level: splits > 1
    (32, 4, 4, 256, 256, 256),     # cfg4 ups.0.0 conv1
    (4, 32, 32, 128, 0, 128),      # cfg4 32x32 level
    (32, 32, 32, 128, 0, 128),     # ... at the full batch: the headline shape of bench.py's roofline (512 workgroups, two rounds)
    (4, 16, 16, 256, 0, 256),
    (2, 2, 2, 32, 0, 64),          # a single tile per image
]


@pytest.mark.parametrize("B,H,W,c0,c1,N", WINO_CASES)
def test_conv_winograd(ops, B, H, W, c0, c1, N):
    """conv3x3_wino_kernel (F(2x2,3x3) on the fp32 MFMA) against F.conv2d: same bar as the direct kernels; bias, residual,
    Mish epilogue, concat, split reduction, run-to-run bit stability, and agreement with the direct kernel."""
    cin = c0 + c1
    x = rnd(B, cin, H, W, seed=61)
    w = rnd(N, cin, 3, 3, seed=62, scale=(cin * 9) ** -0.5)
    bias = rnd(N, seed=63, scale=0.1)
    ref = F.conv2d(x, w, bias, padding=1)
    wp, wu = ops.pack_conv_weight(w.to(DEV)), ops.pack_conv_weight_wino(w.to(DEV))
    assert ops.L.load().ddk_conv_wino_splits(B, H, W, cin, N) >= 1
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    out = ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1, w_wino=wu)
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5
    direct = ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1)
    assert rel_err(out.cpu(), direct.cpu()) < 2e-5
    resid = rnd(*ref.shape, seed=64)
    out_r = ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1, resid=to_nhwc(resid).to(DEV), post_mish=True, w_wino=wu)
    assert rel_err(to_nchw(out_r.cpu()), U.mish(ref + resid)) < 2e-5
    assert torch.equal(out, ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1, w_wino=wu))


def test_conv_winograd_not_eligible_falls_back_to_direct(ops):
    """odd W / N % 64 != 0: the Winograd weights are ignored, the direct kernel runs (same entry point, same result)"""
    x = rnd(2, 32, 6, 7, seed=65)
    w = rnd(32, 32, 3, 3, seed=66, scale=(32 * 9) ** -0.5)
    assert ops.L.load().ddk_conv_wino_splits(2, 6, 7, 32, 32) == 0
    out = ops.conv(ops.CONV3X3_S1, to_nhwc(x).to(DEV), ops.pack_conv_weight(w.to(DEV)), None, w_wino=ops.pack_conv_weight_wino(w.to(DEV)))
    assert rel_err(to_nchw(out.cpu()), F.conv2d(x, w, None, padding=1)) < 2e-5


@pytest.mark.parametrize("B,H,W,c0,c1,N", [(32, 16, 16, 64, 0, 256), (16, 32, 32, 64, 0, 128), (32, 16, 16, 32, 32, 256), (1, 32, 16, 32, 0, 64),
                                          (40, 16, 16, 32, 0, 64)])
def test_conv_winograd_groupnorm_partials(ops, B, H, W, c0, c1, N):
    """One-pass Winograd conv that also emits per-tile {mean, M2} + the GroupNorm kernel that merges them (one read, one write)
    against torch's conv2d -> group_norm -> mish (blocks.py:75-84); bit-stable; agrees with the register-resident GroupNorm."""
    cin = c0 + c1
    lib = ops.L.load()
    assert lib.ddk_conv_gn_partials(B, H, W, cin, N, 8) == H * W // 128
    x = rnd(B, cin, H, W, seed=81)
    w = rnd(N, cin, 3, 3, seed=82, scale=(cin * 9) ** -0.5)
    bias = rnd(N, seed=83, scale=0.3)
    gamma, beta = 1 + rnd(N, seed=84, scale=0.2), rnd(N, seed=85, scale=0.2)
    temb = rnd(B, N, seed=86)
    resid = rnd(B, N, H, W, seed=87)
    h = F.group_norm(F.conv2d(x, w, bias, padding=1), 8, gamma, beta, eps=1e-5)
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    wp, wu = ops.pack_conv_weight(w.to(DEV)), ops.pack_conv_weight_wino(w.to(DEV))
    args = (x0, wp, bias.to(DEV), gamma.to(DEV), beta.to(DEV))
    out = ops.conv3x3_groupnorm_mish(*args, x2=x1, temb=temb.to(DEV), addend=to_nhwc(resid).to(DEV), w_wino=wu)
    assert rel_err(to_nchw(out.cpu()), U.mish(h) + temb[:, :, None, None] + resid) < 2e-5
    assert torch.equal(out, ops.conv3x3_groupnorm_mish(*args, x2=x1, temb=temb.to(DEV), addend=to_nhwc(resid).to(DEV), w_wino=wu))
    two = ops.groupnorm_mish(ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1, w_wino=wu), gamma.to(DEV), beta.to(DEV), temb=temb.to(DEV),
                             addend=to_nhwc(resid).to(DEV))
    assert rel_err(out.cpu(), two.cpu()) < 2e-6


def test_conv_groupnorm_partials_offset_mean(ops):
    """statistics as {mean, M2} per tile, merged by Chan's formula: a conv output with |mean| >> std (large bias) keeps its precision"""
    B, H, W, cin, N = 2, 16, 16, 32, 64
    x = rnd(B, cin, H, W, seed=88)
    w = rnd(N, cin, 3, 3, seed=89, scale=0.01 * (cin * 9) ** -0.5)
    bias = 50.0 + rnd(N, seed=90)
    gamma, beta = torch.ones(N), torch.zeros(N)
    ref = U.mish(F.group_norm(F.conv2d(x.double(), w.double(), bias.double(), padding=1), 8, eps=1e-5)).float()
    assert ops.L.load().ddk_conv_gn_partials(B, H, W, cin, N, 8) == 2      # one channel chunk: no split
    out = ops.conv3x3_groupnorm_mish(to_nhwc(x).to(DEV), ops.pack_conv_weight(w.to(DEV)), bias.to(DEV), gamma.to(DEV), beta.to(DEV),
                                     w_wino=ops.pack_conv_weight_wino(w.to(DEV)))
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-4      # the input of the normalisation itself carries ~50 * 2^-24 of rounding


LOCAL_CASES = [    # B, H, W, c0, c1, N   (GroupNorm groups = 8)
    (32, 4, 4, 256, 0, 256),       # cfg4 4x4 level (blocks.py:75-84 at downs.3 / mid / ups.0)
    (32, 4, 4, 256, 256, 256),     # ups.0.0 conv1: concat of two sources (unet.py:97)
    (3, 4, 4, 64, 0, 128),         # 16 channels per group: two groups per 32-channel tile
    (2, 4, 4, 32, 0, 64),          # 8 channels per group; a single (tap, chunk) unit for most waves
    (5, 8, 8, 256, 0, 256),        # 8x8 map: four M blocks
    (2, 8, 8, 128, 384, 256),      # 8x8 concat, c0 != c1, 134 KB of LDS
    (2, 2, 8, 96, 0, 96),          # non-square 16-pixel map, 3 chunks, 12 channels/group is not eligible -> see below
    (3, 2, 8, 64, 32, 64),         # non-square 16-pixel map, concat, 8 channels per group
    (1, 8, 2, 32, 0, 64),          # batch 1
    (64, 2, 2, 256, 0, 256),       # cfg3 2x2 level (16x16 latents, 3 downsamples): four 4-pixel images share a 16-row block
    (8, 2, 2, 256, 256, 256),      # ... with the concat source of the up path
    (4, 2, 2, 64, 0, 128),         # one block, 16 channels per group
    (12, 2, 2, 32, 0, 64),         # three blocks, 8 channels per group
]


@pytest.mark.parametrize("B,H,W,c0,c1,N", LOCAL_CASES)
def test_conv3x3_groupnorm_mish_one_launch(ops, B, H, W, c0, c1, N):
    """conv3x3_gn_local_kernel: Conv2d(3, padding=1) -> GroupNorm(8) -> Mish (+ time shift) (+ residual) in one launch
    (blocks.py:75-84, 110-115) against torch; same 2e-5-of-max bar as the two-launch path, and bit-stable run to run."""
    cin = c0 + c1
    lib = ops.L.load()
    if not lib.ddk_conv3x3_gn_mish_ok(H, W, cin, c0, N, 8):
        assert (N // 8) not in (8, 16, 32)
        pytest.skip("shape not eligible (channels per group)")
    x = rnd(B, cin, H, W, seed=71)
    w = rnd(N, cin, 3, 3, seed=72, scale=(cin * 9) ** -0.5)
    bias = rnd(N, seed=73, scale=0.1)
    gamma, beta = 1 + rnd(N, seed=74, scale=0.2), rnd(N, seed=75, scale=0.2)
    temb = rnd(B, N, seed=76)
    resid = rnd(B, N, H, W, seed=77)
    h = F.group_norm(F.conv2d(x, w, bias, padding=1), 8, gamma, beta, eps=1e-5)
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    wp = ops.pack_conv_weight(w.to(DEV))
    args = (x0, ops.pack_conv_weight_local(w.to(DEV)), bias.to(DEV), gamma.to(DEV), beta.to(DEV))
    out = ops.conv3x3_gn_mish(*args, x2=x1)
    assert rel_err(to_nchw(out.cpu()), U.mish(h)) < 2e-5
    out_t = ops.conv3x3_gn_mish(*args, temb=temb.to(DEV), x2=x1)
    assert rel_err(to_nchw(out_t.cpu()), U.mish(h) + temb[:, :, None, None]) < 2e-5
    out_r = ops.conv3x3_gn_mish(*args, addend=to_nhwc(resid).to(DEV), x2=x1)
    assert rel_err(to_nchw(out_r.cpu()), U.mish(h) + resid) < 2e-5
    assert torch.equal(out, ops.conv3x3_gn_mish(*args, x2=x1))
    # the two-launch path (conv, then GroupNorm+Mish) gives the same tensor up to summation order
    two = ops.groupnorm_mish(ops.conv(ops.CONV3X3_S1, x0, wp, bias.to(DEV), x2=x1), gamma.to(DEV), beta.to(DEV))
    assert rel_err(out.cpu(), two.cpu()) < 2e-5


WLOCAL_CASES = [   # B, H, W, c0, c1, N
    (32, 8, 8, 256, 0, 256),       # cfg4 8x8 level
    (3, 8, 8, 64, 0, 128),         # 16 channels per group
    (2, 8, 8, 32, 0, 64),          # one chunk, 8 channels per group
    (2, 4, 16, 96, 32, 96),        # non-square 64-pixel map, concat; 12 channels per group is not eligible
    (2, 16, 4, 128, 192, 256),     # concat filling the 320-channel LDS budget
    (5, 8, 8, 160, 0, 64),         # odd chunk count (5)
    (1, 4, 16, 64, 0, 64),         # batch 1, non-square
    (3, 32, 2, 32, 32, 64),        # 32x2 map (tiles 16x1), concat
]


@pytest.mark.parametrize("B,H,W,c0,c1,N", WLOCAL_CASES)
def test_conv3x3_groupnorm_mish_one_launch_winograd(ops, B, H, W, c0, c1, N):
    """conv3x3_gn_wlocal_kernel (64-pixel maps, Winograd F(2x2,3x3) inside the image-local tiling) against torch's
    conv2d -> group_norm -> mish (+ shift, + residual), blocks.py:75-84, 110-115; bit-stable; agrees with the two-launch path."""
    cin = c0 + c1
    lib = ops.L.load()
    if not lib.ddk_conv3x3_gn_mish_wino_ok(H, W, cin, c0, N, 8):
        assert (N // 8) not in (8, 16, 32)
        pytest.skip("shape not eligible (channels per group)")
    x = rnd(B, cin, H, W, seed=91)
    w = rnd(N, cin, 3, 3, seed=92, scale=(cin * 9) ** -0.5)
    bias = rnd(N, seed=93, scale=0.1)
    gamma, beta = 1 + rnd(N, seed=94, scale=0.2), rnd(N, seed=95, scale=0.2)
    temb = rnd(B, N, seed=96)
    resid = rnd(B, N, H, W, seed=97)
    h = F.group_norm(F.conv2d(x, w, bias, padding=1), 8, gamma, beta, eps=1e-5)
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    args = (x0, ops.pack_conv_weight_wino_local(w.to(DEV)), bias.to(DEV), gamma.to(DEV), beta.to(DEV))
    out = ops.conv3x3_gn_mish_wino(*args, x2=x1)
    assert rel_err(to_nchw(out.cpu()), U.mish(h)) < 2e-5
    out_t = ops.conv3x3_gn_mish_wino(*args, temb=temb.to(DEV), addend=to_nhwc(resid).to(DEV), x2=x1)
    assert rel_err(to_nchw(out_t.cpu()), U.mish(h) + temb[:, :, None, None] + resid) < 2e-5
    assert torch.equal(out, ops.conv3x3_gn_mish_wino(*args, x2=x1))
    two = ops.groupnorm_mish(ops.conv(ops.CONV3X3_S1, x0, ops.pack_conv_weight(w.to(DEV)), bias.to(DEV), x2=x1), gamma.to(DEV), beta.to(DEV))
    assert rel_err(out.cpu(), two.cpu()) < 2e-5


def test_conv3x3_groupnorm_mish_rejects_other_maps(ops):
    x = torch.zeros(1, 16, 16, 32, device=DEV)
    wp = torch.zeros(1, 9, 1, 1024, device=DEV)
    z = torch.zeros(32, device=DEV)
    with pytest.raises(ops.L.DDKError):
        ops.conv3x3_gn_mish(x, wp, z, z, z)


def test_conv_pre_post_mish(ops):
    x = rnd(2, 64, 16, 16, seed=5)
    w = rnd(32, 64, 1, 1, seed=6, scale=0.125)
    b = rnd(32, seed=7, scale=0.1)
    ref = U.mish(F.conv2d(U.mish(x), w, b))
    out = ops.conv(ops.CONV1X1, to_nhwc(x).to(DEV), ops.pack_conv_weight(w.to(DEV)), b.to(DEV), pre_mish=True, post_mish=True)
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5
    w3 = rnd(32, 32, 3, 3, seed=8, scale=(32 * 9) ** -0.5)
    x3 = rnd(2, 32, 8, 8, seed=9)          # split-K path: post_mish applied by the reduce kernel
    ref3 = U.mish(F.conv2d(x3, w3, b, padding=1))
    out3 = ops.conv(ops.CONV3X3_S1, to_nhwc(x3).to(DEV), ops.pack_conv_weight(w3.to(DEV)), b.to(DEV), post_mish=True)
    assert rel_err(to_nchw(out3.cpu()), ref3) < 2e-5


C32_CASES = [  # B, H, W: conv3x3 32 -> 32 on >= 128 whole 128-pixel tiles -> the register-resident kernel (conv_c32_kernel.inc)
    (8, 64, 64),      # cfg3 encoder / decoder level 0: 2-row tiles, 4 tiles per workgroup
    (33, 32, 32),     # 4-row tiles, a tile count (264) that is not a multiple of 8: plain tile order
    (128, 16, 16),    # 8-row tiles, two per image
    (512, 8, 8),      # two images per tile
    (5, 64, 128),     # one row per tile, non-square
    (64, 16, 16),     # 128 tiles: the smallest grid the kernel takes (half the chip)
]


@pytest.mark.parametrize("B,H,W", C32_CASES)
def test_conv3x3_32_to_32_register_resident(ops, B, H, W):
    """the dDDPM encoder / decoder's 3x3 convs (convblocks.py:112-130, d_chans 64) with every epilogue the training path uses:
    bias, second output Mish(out) (ddk_conv_args.mish_out), Mish' of the pre-activation (dmish_src), residual, Mish"""
    x = rnd(B, 32, H, W, seed=31)
    w = rnd(32, 32, 3, 3, seed=32, scale=(32 * 9) ** -0.5)
    b = rnd(32, seed=33, scale=0.1)
    ref = F.conv2d(x, w, b, padding=1)
    xd, wp, bd = to_nhwc(x).to(DEV), ops.pack_conv_weight(w.to(DEV)), b.to(DEV)
    out = ops.conv(ops.CONV3X3_S1, xd, wp, bd)
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5
    assert torch.equal(out, ops.conv(ops.CONV3X3_S1, xd, wp, bd))
    a_out = torch.empty_like(out)
    out2 = ops.conv(ops.CONV3X3_S1, xd, wp, bd, mish_out=a_out)
    assert torch.equal(out2, out) and rel_err(to_nchw(a_out.cpu()), U.mish(ref)) < 2e-5
    res = rnd(B, 32, H, W, seed=34)
    out3 = ops.conv(ops.CONV3X3_S1, xd, wp, bd, resid=to_nhwc(res).to(DEV), post_mish=True)
    assert rel_err(to_nchw(out3.cpu()), U.mish(ref + res)) < 2e-5
    # the input-gradient form: dX = conv(dY, flipped / transposed filter) * Mish'(h), against autograd of conv(Mish(h))
    h = rnd(B, 32, H, W, seed=35).requires_grad_(True)
    dy = rnd(B, 32, H, W, seed=36)
    F.conv2d(U.mish(h), w, None, padding=1).backward(dy)
    wd = ops.pack_conv_weight_dgrad(w.to(DEV), i_pad=32)
    dx = ops.conv(ops.CONV3X3_S1, to_nhwc(dy).to(DEV), wd, None, n_out=32, dmish_src=to_nhwc(h.detach()).to(DEV))
    assert rel_err(to_nchw(dx.cpu()), h.grad) < 3e-5


SM_CASES = [  # B, H, W, c0, c1, N: 1x1 convs on small maps -> conv1x1_sm_kernel (32x32 tiles, the four waves split K)
    (32, 4, 4, 128, 0, 256),      # to_out at 4x4 (cfg4)
    (32, 8, 8, 128, 0, 256),      # to_out at 8x8
    (32, 4, 4, 256, 256, 256),    # res_conv over the concat at 4x4: K = 512 from two sources
    (32, 8, 8, 256, 256, 256),    # ... at 8x8
    (8, 8, 8, 256, 0, 128),       # K = 256, 64 tiles (the smallest grid the kernel takes)
    (3, 16, 32, 128, 0, 96),      # ragged sizes that still tile by 32
]


@pytest.mark.parametrize("B,H,W,c0,c1,N", SM_CASES)
def test_conv1x1_small_maps(ops, B, H, W, c0, c1, N):
    cin = c0 + c1
    x = rnd(B, cin, H, W, seed=41)
    w = rnd(N, cin, 1, 1, seed=42, scale=cin ** -0.5)
    b = rnd(N, seed=43, scale=0.1)
    res = rnd(B, N, H, W, seed=44)
    ref = F.conv2d(x, w, b)
    xh = to_nhwc(x).to(DEV)
    x0 = xh[..., :c0].contiguous()
    x1 = xh[..., c0:].contiguous() if c1 else None
    wp = ops.pack_conv_weight(w.to(DEV))
    out = ops.conv(ops.CONV1X1, x0, wp, b.to(DEV), x2=x1)
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5
    out_r = ops.conv(ops.CONV1X1, x0, wp, b.to(DEV), x2=x1, resid=to_nhwc(res).to(DEV))
    assert rel_err(to_nchw(out_r.cpu()), ref + res) < 2e-5
    out_n = ops.conv(ops.CONV1X1, x0, wp, None, x2=x1)
    assert rel_err(to_nchw(out_n.cpu()), F.conv2d(x, w)) < 2e-5
    assert torch.equal(out, ops.conv(ops.CONV1X1, x0, wp, b.to(DEV), x2=x1))


STREAM_CASES = [  # B, H, W, cin, N: 1x1 convs between 32 / 64-channel tensors on >= 16384 pixels -> conv1x1_stream_kernel
    (16, 32, 32, 64, 32),     # c1 of an encoder block at 32x32
    (16, 32, 32, 32, 64),     # c4
    (4, 64, 64, 64, 64),
    (8, 64, 64, 32, 32),
    (1, 128, 130, 64, 32),    # 1040 tiles: the last round of the grid-stride walk is ragged
    (64, 16, 16, 3, 64),      # first 1x1 of the encoder: 3 input channels padded to 32
]


@pytest.mark.parametrize("B,H,W,cin,N", STREAM_CASES)
def test_conv1x1_stream(ops, B, H, W, cin, N):
    """conv1x1_stream.hip (weights in registers, no LDS, float4 epilogue): every epilogue of ddk_conv_args against torch"""
    x = rnd(B, cin, H, W, seed=51)
    w = rnd(N, cin, 1, 1, seed=52, scale=cin ** -0.5)
    b = rnd(N, seed=53, scale=0.1)
    res = rnd(B, N, H, W, seed=54)
    src = rnd(B, N, H, W, seed=55)
    cp = ops.pad32(cin)
    xd = ops.nchw_to_nhwc(x.to(DEV), cp)
    wp = ops.pack_conv_weight(w.to(DEV))
    bd, rd, sd = b.to(DEV), to_nhwc(res).to(DEV), to_nhwc(src).to(DEV)
    ref = F.conv2d(x, w, b)
    assert rel_err(to_nchw(ops.conv(ops.CONV1X1, xd, wp, bd).cpu()), ref) < 2e-5
    assert rel_err(to_nchw(ops.conv(ops.CONV1X1, xd, wp, None, resid=rd).cpu()), F.conv2d(x, w) + res) < 2e-5
    pm = ops.conv(ops.CONV1X1, xd, wp, bd, pre_mish=True, post_mish=True)
    assert rel_err(to_nchw(pm.cpu()), F.mish(F.conv2d(F.mish(x), w, b))) < 2e-5
    a_out = torch.empty(B, H, W, N, device=DEV)
    h = ops.conv(ops.CONV1X1, xd, wp, bd, mish_out=a_out)
    assert rel_err(to_nchw(h.cpu()), ref) < 2e-5 and rel_err(to_nchw(a_out.cpu()), F.mish(ref)) < 2e-5
    sg = src.clone().requires_grad_(True)
    F.mish(sg).sum().backward()                     # sg.grad = Mish'(src)
    dg = ops.conv(ops.CONV1X1, xd, wp, bd, dmish_src=sd, resid=rd)
    assert rel_err(to_nchw(dg.cpu()), ref * sg.grad + res) < 2e-5
    assert torch.equal(dg, ops.conv(ops.CONV1X1, xd, wp, bd, dmish_src=sd, resid=rd))


def test_conv_small_cin_padding(ops):
    """C_in in {1,3,8}: channels zero-padded to 32 on both operands (first UNet conv / res_conv)."""
    for cin in (1, 3, 8):
        x = rnd(2, cin, 16, 16, seed=10 + cin)
        w = rnd(64, cin, 3, 3, seed=20 + cin, scale=(cin * 9) ** -0.5)
        ref = F.conv2d(x, w, None, padding=1)
        out = ops.conv(ops.CONV3X3_S1, ops.nchw_to_nhwc(x.to(DEV), 32), ops.pack_conv_weight(w.to(DEV)))
        assert rel_err(to_nchw(out.cpu()), ref) < 2e-5


def test_conv_argument_errors(ops):
    from ddk.lib import DDKError
    x = torch.zeros(1, 4, 4, 48, device=DEV)
    with pytest.raises(DDKError):
        ops.conv(ops.CONV1X1, x, torch.zeros(32, 1, 64, device=DEV))      # c0 not a multiple of 32
    # the im2col copy of a filter may be omitted only where the Winograd copy is given AND the launch takes that path
    x = torch.randn(2, 8, 8, 64, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV) * 0.05
    wu = ops.pack_conv_weight_wino(w)
    ref = ops.conv(ops.CONV3X3_S1, x, ops.pack_conv_weight(w), None, n_out=64, w_wino=wu)
    assert torch.equal(ops.conv(ops.CONV3X3_S1, x, None, None, n_out=64, w_wino=wu), ref)
    with pytest.raises(DDKError):
        ops.conv(ops.CONV3X3_S1, x, None, None, n_out=64, w_wino=wu, pre_mish=True)       # pre_mish lives on the im2col kernels
    with pytest.raises(DDKError):
        ops.conv(ops.CONV3X3_S1, torch.randn(2, 7, 7, 64, device=DEV), None, None, n_out=64, w_wino=wu)   # odd map: not Winograd-eligible


# ---------------------------------------------------------------- norms
@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 32), (32, 32, 32, 128), (32, 16, 16, 256), (4, 4, 4, 256), (3, 5, 7, 64),
                                     (2, 128, 128, 32), (1, 256, 256, 128)])
def test_groupnorm_mish(ops, B, H, W, C):
    x = rnd(B, C, H, W, seed=30, scale=2.0) + 0.5
    g, b = 1 + 0.1 * rnd(C, seed=31), 0.1 * rnd(C, seed=32)
    temb = rnd(B, C + 8, seed=33)[:, 4:4 + C]            # strided view, like a slice of the [B][3584] table
    add = rnd(B, C, H, W, seed=34)
    ref0 = U.mish(F.group_norm(x, 8, g, b, 1e-5))
    xh = to_nhwc(x).to(DEV)
    out0 = ops.groupnorm_mish(xh, g.to(DEV), b.to(DEV))
    assert rel_err(to_nchw(out0.cpu()), ref0) < 5e-6
    tb = rnd(B, C + 8, seed=33).to(DEV)
    out1 = ops.groupnorm_mish(xh, g.to(DEV), b.to(DEV), temb=tb[:, 4:4 + C], addend=to_nhwc(add).to(DEV))
    assert rel_err(to_nchw(out1.cpu()), ref0 + temb[:, :, None, None] + add) < 5e-6


@pytest.mark.parametrize("B,H,W,cin,N", [(2, 8, 8, 64, 64), (32, 4, 4, 256, 256), (4, 16, 16, 128, 128), (32, 32, 32, 128, 128)])
def test_conv_groupnorm_fused_reduce(ops, B, H, W, cin, N):
    """conv3x3 -> GN+Mish with the split-K reduction folded into the GroupNorm load == the two-kernel form"""
    x = rnd(B, cin, H, W, seed=35)
    w = rnd(N, cin, 3, 3, seed=36, scale=(cin * 9) ** -0.5)
    bias, g, b = rnd(N, seed=37, scale=0.1), 1 + 0.1 * rnd(N, seed=38), 0.1 * rnd(N, seed=39)
    temb, add = rnd(B, N, seed=40), rnd(B, N, H, W, seed=41)
    ref = U.mish(F.group_norm(F.conv2d(x, w, bias, padding=1), 8, g, b, 1e-5)) + temb[:, :, None, None] + add
    out = ops.conv3x3_groupnorm_mish(to_nhwc(x).to(DEV), ops.pack_conv_weight(w.to(DEV)), bias.to(DEV), g.to(DEV), b.to(DEV),
                                     temb=temb.to(DEV), addend=to_nhwc(add).to(DEV))
    assert rel_err(to_nchw(out.cpu()), ref) < 2e-5


@pytest.mark.parametrize("C", [32, 64, 128, 256, 512])
def test_chan_layernorm(ops, C):
    x = rnd(3, C, 6, 5, seed=40, scale=3.0) + 1.0
    g, b = 1 + 0.1 * rnd(1, C, 1, 1, seed=41), 0.1 * rnd(1, C, 1, 1, seed=42)
    out = ops.chan_layernorm(to_nhwc(x).to(DEV), g.to(DEV), b.to(DEV))
    assert rel_err(to_nchw(out.cpu()), U.chan_layernorm(x, g, b)) < 5e-6


def test_mish_matches_torch(ops):
    x = torch.cat([torch.linspace(-30, 30, 4097), torch.tensor([-100.0, 100.0, 0.0, 20.0, 20.000002, -20.0])])
    out = ops.mish(x.to(DEV)).cpu()
    ref = F.mish(x)
    assert (out - ref).abs().max() < 2e-6 and rel_err(out, ref) < 1e-6
    assert rel_err(ops.tanh(x.to(DEV)).cpu(), torch.tanh(x)) < 1e-6


def test_pool_upsample_add(ops):
    x = rnd(2, 64, 8, 12, seed=50)
    xh = to_nhwc(x).to(DEV)
    assert torch.equal(to_nchw(ops.upsample_nearest2(xh).cpu()), F.interpolate(x, scale_factor=2))
    assert rel_err(to_nchw(ops.avgpool2(xh).cpu()), F.avg_pool2d(x, 2, 2)) < 1e-6
    y = rnd(2, 64, 8, 12, seed=51)
    assert torch.equal(ops.add(xh, to_nhwc(y).to(DEV)).cpu(), to_nhwc(x + y))


# ---------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,W", [(2, 4, 4), (3, 8, 8), (2, 32, 32), (1, 10, 13), (32, 16, 16), (2, 15, 17), (1, 16, 17)])
def test_linattn(ops, B, H, W):
    qkv = rnd(B, 384, H, W, seed=60, scale=1.5)
    q, k, v = qkv.reshape(B, 3, 4, 32, H * W).unbind(1)
    ks = k.softmax(dim=-1)
    ctx_ref = torch.einsum("bhdn,bhen->bhde", ks, v)
    out_ref = torch.einsum("bhde,bhdn->bhen", ctx_ref, q).reshape(B, 128, H, W)
    out, ctx = ops.linattn(to_nhwc(qkv).to(DEV), 4)
    assert rel_err(ctx.cpu(), ctx_ref) < 1e-5
    assert rel_err(to_nchw(out.cpu()), out_ref) < 1e-5
    if 64 < H * W <= 256:        # ... which was the one-launch kernel with 256 rows in LDS; the context / merge / apply path:
        out1, ctx1 = ops.linattn(to_nhwc(qkv).to(DEV), 4, fused_up_to=64)
        assert rel_err(ctx1.cpu(), ctx_ref) < 1e-5 and rel_err(to_nchw(out1.cpu()), out_ref) < 1e-5


@pytest.mark.parametrize("B,H,W,C", [(32, 4, 4, 256), (5, 8, 8, 256), (3, 8, 8, 128), (2, 2, 2, 64), (2, 6, 6, 96), (1, 8, 8, 384)])
def test_linattn_small_maps_projection_inside(ops, B, H, W, C):
    """linattn_small_qkv_kernel: channel LayerNorm -> to_qkv -> softmax_n(k) -> ctx -> out for one (image, head) per workgroup
    (blocks.py:57-60, 123-134), against torch; and against the two-launch path (LN-folded to_qkv conv, then the core)."""
    x = rnd(B, C, H, W, seed=62, scale=1.3) + 0.4
    g, b = 1 + rnd(C, seed=63, scale=0.2), rnd(C, seed=64, scale=0.2)
    wq = rnd(384, C, seed=65, scale=C ** -0.5)
    mean = x.mean(dim=1, keepdim=True)
    std = x.var(dim=1, unbiased=False, keepdim=True).sqrt()
    xn = (x - mean) / (std + 1e-5) * g[None, :, None, None] + b[None, :, None, None]
    qkv = F.conv2d(xn, wq[:, :, None, None])
    q, k, v = qkv.reshape(B, 3, 4, 32, H * W).unbind(1)
    ctx_ref = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
    out_ref = torch.einsum("bhde,bhdn->bhen", ctx_ref, q).reshape(B, 128, H, W)
    out, ctx = ops.linattn_small_from_x(to_nhwc(x).to(DEV), wq.to(DEV), g.to(DEV), b.to(DEV))
    assert rel_err(ctx.cpu(), ctx_ref) < 2e-5
    assert rel_err(to_nchw(out.cpu()), out_ref) < 2e-5
    out2, _ = ops.linattn_small_from_x(to_nhwc(x).to(DEV), wq.to(DEV), g.to(DEV), b.to(DEV))
    assert torch.equal(out, out2)
    two, _ = ops.linattn(to_nhwc(qkv).to(DEV), 4)
    assert rel_err(out.cpu(), two.cpu()) < 2e-5


def test_linattn_small_maps_offset_input(ops):
    """|mean| >> std per pixel (x + 50): the in-kernel LayerNorm variance is two-pass like torch.var, not E[x^2] - mean^2"""
    B, H, W, C = 4, 4, 4, 256
    x = rnd(B, C, H, W, seed=66, scale=0.5) + 50.0
    g, b = 1 + rnd(C, seed=67, scale=0.2), rnd(C, seed=68, scale=0.2)
    wq = rnd(384, C, seed=69, scale=C ** -0.5)
    xd = x.double()
    mean = xd.mean(dim=1, keepdim=True)
    std = xd.var(dim=1, unbiased=False, keepdim=True).sqrt()
    xn = ((xd - mean) / (std + 1e-5) * g.double()[None, :, None, None] + b.double()[None, :, None, None])
    qkv = F.conv2d(xn, wq.double()[:, :, None, None])
    q, k, v = qkv.reshape(B, 3, 4, 32, H * W).unbind(1)
    ctx_ref = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
    out_ref = torch.einsum("bhde,bhdn->bhen", ctx_ref, q).reshape(B, 128, H, W)
    out, ctx = ops.linattn_small_from_x(to_nhwc(x).to(DEV), wq.to(DEV), g.to(DEV), b.to(DEV))
    # the folded form r (W o g) x - r mean (W g) still subtracts two numbers ~ |mean| / std apart: ~1e-7 * 100 of relative noise
    assert rel_err(to_nchw(out.cpu()), out_ref.float()) < 3e-4
    assert rel_err(ctx.cpu(), ctx_ref.float()) < 3e-4


def test_linattn_softmax_extremes(ops):
    """a dominant key (softmax ~ one-hot) and large negative logits must not overflow / lose the max"""
    B, H, W = 1, 8, 8
    qkv = rnd(B, 384, H, W, seed=61)
    qkv[:, 128:256, 3, 5] += 60.0
    qkv[:, 128:256, 0, 0] -= 80.0
    q, k, v = qkv.reshape(B, 3, 4, 32, H * W).unbind(1)
    ctx_ref = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
    _, ctx = ops.linattn(to_nhwc(qkv).to(DEV), 4)
    assert torch.isfinite(ctx).all() and rel_err(ctx.cpu(), ctx_ref) < 1e-5


# ---------------------------------------------------------------- time embedding
@pytest.mark.parametrize("dim", [32, 128])
def test_time_embedding(ops, dim):
    from ddk.plan import sinusoidal_freqs
    t = torch.tensor([0, 1, 17, 500, 999, 3, 3, 250, 731])
    w1, b1 = rnd(4 * dim, dim, seed=70, scale=dim ** -0.5), rnd(4 * dim, seed=71, scale=0.1)
    w2, b2 = rnd(dim, 4 * dim, seed=72, scale=(4 * dim) ** -0.5), rnd(dim, seed=73, scale=0.1)
    wp, bp = rnd(200, dim, seed=74, scale=dim ** -0.5), rnd(200, seed=75, scale=0.1)
    e = U.sinusoidal_embedding(t, dim)
    raw_ref = F.linear(U.mish(F.linear(e, w1, b1)), w2, b2)
    act, raw = ops.time_mlp(t.to(DEV), sinusoidal_freqs(dim).to(DEV), w1.t().contiguous().to(DEV), b1.to(DEV),
                            w2.t().contiguous().to(DEV), b2.to(DEV))
    assert rel_err(raw.cpu(), raw_ref) < 2e-5
    assert rel_err(act.cpu(), U.mish(raw_ref)) < 2e-5
    out = ops.time_proj(act, wp.t().contiguous().to(DEV), bp.to(DEV))
    assert rel_err(out.cpu(), F.linear(U.mish(raw_ref), wp, bp)) < 2e-5


def test_conv1x1_small_n(ops):
    for C, n_out in ((128, 8), (128, 3), (32, 1), (64, 3), (64, 8), (256, 8), (256, 5), (128, 1)):
        x = rnd(2, C, 9, 7, seed=80)
        w, b = rnd(n_out, C, 1, 1, seed=81, scale=C ** -0.5), rnd(n_out, seed=82)
        out = ops.conv1x1_small_n(to_nhwc(x).to(DEV), w.to(DEV), b.to(DEV))
        assert rel_err(to_nchw(out.cpu()), F.conv2d(x, w, b)) < 1e-5


def test_layout_roundtrip(ops):
    x = rnd(3, 5, 6, 7, seed=90)
    xh = ops.nchw_to_nhwc(x.to(DEV), 32)
    assert xh.shape == (3, 6, 7, 32) and torch.equal(xh[..., :5].cpu(), to_nhwc(x)) and (xh[..., 5:] == 0).all()
    assert torch.equal(ops.nhwc_to_nchw(xh, 5).cpu(), x)
    assert torch.equal(ops.pad_channels(to_nhwc(x).to(DEV), 32).cpu(), xh.cpu())


# ---------------------------------------------------------------- noise-schedule arithmetic
def _tables():
    buf = D.schedule_buffers("linear", 1000)
    sigma = (0.5 * buf["posterior_log_variance_clipped"]).exp()
    return buf, sigma


def test_q_sample_bit_exact(ops):
    buf, _ = _tables()
    x, eps = rnd(5, 3, 16, 16, seed=100), rnd(5, 3, 16, 16, seed=101)
    t = torch.tensor([0, 1, 499, 998, 999])
    out = ops.q_sample(x.to(DEV), eps.to(DEV), t.to(DEV), buf["sqrt_alphas_cumprod"].to(DEV),
                       buf["sqrt_one_minus_alphas_cumprod"].to(DEV))
    assert torch.equal(out.cpu(), D.q_sample(buf, x, t, eps))


def test_p_sample_update_bit_exact(ops):
    buf, sigma = _tables()
    x, eh, z = rnd(6, 8, 8, 8, seed=110, scale=1.5), rnd(6, 8, 8, 8, seed=111), rnd(6, 8, 8, 8, seed=112)
    t = torch.tensor([0, 1, 2, 500, 998, 999])
    ref = D.p_sample_update(buf, x, t, eh, z)
    tb = {k: buf[v].to(DEV) for k, v in (("c_recip", "sqrt_recip_alphas_cumprod"), ("c_recipm1", "sqrt_recipm1_alphas_cumprod"),
                                         ("c1", "posterior_mean_coef1"), ("c2", "posterior_mean_coef2"))}
    out = ops.p_sample_update_(x.to(DEV).clone(), eh.to(DEV), t.to(DEV), sigma=sigma.to(DEV), noise=z.to(DEV), **tb)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(out[0].cpu(), ref[0]) and (ref[0] - D.p_sample_update(buf, x, t, eh, 0 * z)[0]).abs().max() == 0  # t==0: no noise
    # in-kernel Philox noise == oracle Philox + the same update
    seed, stream = 0x1234567887654321, 5
    out_p = ops.p_sample_update_(x.to(DEV).clone(), eh.to(DEV), torch.full((6,), 321, dtype=torch.long, device=DEV),
                                 sigma=sigma.to(DEV), seed=seed, stream_id=stream, **tb)
    zp = torch.from_numpy(philox_ref.philox_normal(x.numel(), seed, 321, stream)).reshape(x.shape)
    ref_p = D.p_sample_update(buf, x, torch.full((6,), 321), eh, zp)
    assert (out_p.cpu() - ref_p).abs().max() < 2e-6


def test_randn_matches_philox_oracle(ops):
    n = 4096 * 3 + 2
    z = ops.randn((n,), DEV, seed=42, step=7, stream_id=1).cpu().numpy()
    ref = philox_ref.philox_normal(n, 42, 7, 1)
    assert np.abs(z - ref).max() < 2e-6
    big = ops.randn((1 << 22,), DEV, seed=1, step=0).cpu().double()
    assert abs(big.mean()) < 3e-3 and abs(big.std() - 1) < 3e-3


def test_sq_err_sum(ops):
    a, b = rnd(4, 8, 32, 32, seed=120), rnd(4, 8, 32, 32, seed=121)
    out = ops.sq_err_sum(a.to(DEV), b.to(DEV)).cpu()
    ref = ((a - b).double() ** 2).sum(dim=(1, 2, 3))
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("B,H,S,SA", [(32, 4, 8, 1), (32, 8, 4, 1), (5, 4, 3, 2), (3, 8, 2, 4)])
def test_one_launch_block_reads_split_k_slabs(ops, B, H, S, SA):
    """ddk_conv3x3_gn_mish_slabs: the image-local Block kernels (4x4 direct, 8x8 Winograd) summing the split-K slabs a Downsample conv left
    for their input (and for the residual) while they load -- equal, bit for bit, to the same Block on the reduced tensors (the
    fixed-order sum + bias of splitk_reduce_kernel); blocks.py:41-47 followed by blocks.py:75-84,105-115."""
    C = N = 256
    slabs = torch.stack([to_nhwc(rnd(B, C, H, H, seed=300 + k)) for k in range(S)]).to(DEV).contiguous()
    sbias = rnd(C, seed=320, scale=0.3).to(DEV)
    w = (rnd(N, C, 3, 3, seed=321, scale=(C * 9) ** -0.5)).to(DEV)
    bias, gamma, beta = rnd(N, seed=322, scale=0.1).to(DEV), (1 + rnd(N, seed=323, scale=0.2)).to(DEV), rnd(N, seed=324, scale=0.2).to(DEV)
    temb = rnd(B, N, seed=325).to(DEV)
    wp = ops.pack_conv_weight_local(w) if H == 4 else ops.pack_conv_weight_wino_local(w)
    plain = ops.conv3x3_gn_mish if H == 4 else ops.conv3x3_gn_mish_wino
    red = slabs[0].clone()
    for k in range(1, S):
        red += slabs[k]
    red += sbias
    a_sl = torch.stack([to_nhwc(rnd(B, N, H, H, seed=340 + k)) for k in range(SA)]).to(DEV).contiguous()
    abias = rnd(N, seed=350, scale=0.3).to(DEV)
    ared = a_sl[0].clone()
    for k in range(1, SA):
        ared += a_sl[k]
    if SA > 1:
        ared += abias
    want = plain(red, wp, bias, gamma, beta, temb=temb, addend=ared)
    got = ops.conv3x3_gn_mish_slabs(slabs, sbias, wp, bias, gamma, beta, temb=temb, addend_slabs=a_sl, addend_bias=abias)
    assert torch.equal(got, want)
    # the residual a ResnetBlock without a skip conv adds is its own input: the same slabs on both ports
    want2 = plain(red, wp, bias, gamma, beta, addend=red)
    got2 = ops.conv3x3_gn_mish_slabs(slabs, sbias, wp, bias, gamma, beta, addend_slabs=slabs, addend_bias=sbias)
    assert torch.equal(got2, want2)
