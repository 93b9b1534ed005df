#!/usr/bin/env python3
"""Launch ONE kernel of the reverse step, at its cfg4 shape, 20 times -- the program rocprofv3 counter passes run
(`rocprofv3 --kernel-trace --pmc ... -- python3 tools/pmc_one.py <case>`; counters and --stats in separate passes).
cases: cls16 (the k-split in-launch GroupNorm conv of the 16x16 up level), wino (conv3x3 128->128 @32x32 B=32, Winograd), gn (GroupNorm-apply from partials, 32x32x32x128), local4 (conv+GroupNorm+Mish
one launch, 256->256 @4x4), wlocal8 (the same @8x8, Winograd form), first (conv_first 8->128 @32x32), tail (final_tail_kernel),
cluster16 (conv3x3 256->256 @16x16 with GroupNorm finished in the launch), ws (to_out 1x1 128->128 + bias + residual @32x32,
weights-stationary kernel), fold (attn_fold_kernel), halo32 (wgrad3x3_halo32_kernel), gnbig (gn_apply_kernel on the 256x256 tensor),
kvctx (attn_kvctx_kernel: k, v projection + context @32x32), c32 (conv3x3 32->32 @64x64 B=64 with the filter in registers), stream (conv1x1 32->64 @64x64 B=64 with Mish' and residual: conv1x1_stream.hip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch  # noqa: E402
from ddk import ops  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "wino"
dev, B = "cuda", 32


def rw(n, c, k=3):
    return torch.randn(n, c, k, k, device=dev) * (c * k * k) ** -0.5


if case == "wino":
    x, w = torch.randn(B, 32, 32, 128, device=dev), rw(128, 128)
    wp, wu, b = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w), torch.zeros(128, device=dev)
    fn = lambda: ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu)
elif case == "gn":
    x, w = torch.randn(B, 32, 32, 128, device=dev), rw(128, 128)
    raw, part, tiles = ops.conv_with_gn_partials(x, ops.pack_conv_weight(w), torch.zeros(128, device=dev), ops.pack_conv_weight_wino(w))
    gam, bet, temb = torch.ones(128, device=dev), torch.zeros(128, device=dev), torch.randn(B, 128, device=dev)
    fn = lambda: ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet, temb=temb)
elif case in ("local4", "wlocal8"):
    H = 4 if case == "local4" else 8
    x, w = torch.randn(B, H, H, 256, device=dev), rw(256, 256)
    gam, bet, b = torch.ones(256, device=dev), torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    temb = torch.randn(B, 256, device=dev)
    if case == "local4":
        wl = ops.pack_conv_weight_local(w)
        fn = lambda: ops.conv3x3_gn_mish(x, wl, b, gam, bet, temb=temb)
    else:
        wl = ops.pack_conv_weight_wino_local(w)
        fn = lambda: ops.conv3x3_gn_mish_wino(x, wl, b, gam, bet, temb=temb)
elif case == "first":
    x, w = torch.randn(B, 32, 32, 8, device=dev), rw(128, 8)
    wf, b = ops.pack_conv_weight_first(w), torch.zeros(128, device=dev)
    fn = lambda: ops.conv_first(x, wf, b, 128)
elif case == "tail":
    x, w = torch.randn(B, 32, 32, 8, device=dev), rw(128, 8)
    raw, part, tiles = ops.conv_first(x, ops.pack_conv_weight_first(w), torch.zeros(128, device=dev), 128)
    gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    wo, bo = torch.randn(8, 128, device=dev) * 128 ** -0.5, torch.zeros(8, device=dev)
    xs = torch.randn(B, 32, 32, 8, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.long)
    tab = {k: torch.rand(1000, device=dev) for k in ("c_recip", "c_recipm1", "c1", "c2", "sigma")}
    fn = lambda: ops.final_tail(raw, part, tiles, gam, bet, wo, bo, x=xs, t=t, tables=tab, seed=3, want_eps=False)
elif case == "ws":
    x, w = torch.randn(B, 32, 32, 128, device=dev), torch.randn(128, 128, device=dev) * 128 ** -0.5
    b, r = torch.zeros(128, device=dev), torch.randn(B, 32, 32, 128, device=dev)
    fn = lambda: ops.conv1x1_ws(x, w, b, r)
elif case == "kvctx":          # k, v projection + context of the folded attention block in one launch (32 images, 32x32, C = 128)
    x = torch.randn(B, 32, 32, 128, device=dev)
    wq, g, be = torch.randn(384, 128, device=dev) * 128 ** -0.5, torch.ones(128, device=dev), torch.zeros(128, device=dev)
    fn = lambda: ops.attention_kv_context(x, wq, g, be)
elif case == "fold":           # the folded attention block's per-image matrix build at cfg4 (32 images, C = 128)
    ctx = torch.randn(B, 4, 32, 32, device=dev) * 0.1
    wq, c1, c2 = torch.randn(128, 128, device=dev) * 0.09, torch.randn(128, device=dev), torch.randn(128, device=dev)
    wo, bo = torch.randn(128, 128, device=dev) * 0.09, torch.zeros(128, device=dev)
    A, a1, a2 = torch.empty(B, 128, 128, device=dev), torch.empty(B, 128, device=dev), torch.empty(B, 128, device=dev)
    from ddk import lib as _L
    fn = lambda: _L.check(_L.load().ddk_attention_fold(_L.ptr(ctx), _L.ptr(wq), _L.ptr(c1), _L.ptr(c2), _L.ptr(wo), _L.ptr(bo), _L.ptr(A),
                                                       _L.ptr(a1), _L.ptr(a2), B, 128, 4, _L.stream()), "attention_fold")
elif case == "halo32":         # narrow weight gradient: 3x3 32->32 on 64 x 64 x 64 pixels (the dDDPM encoder's first block at cfg3)
    x, dy = torch.randn(64, 64, 64, 32, device=dev), torch.randn(64, 64, 64, 32, device=dev)
    gw = torch.zeros(32, 32, 3, 3, device=dev)
    fn = lambda: ops.conv_wgrad_(ops.CONV3X3_S1, x, dy, gw, c_real=32, cw=32, c_off=0)
elif case == "wgrad64":        # weight gradient of a 3x3 conv 256->256 on 128 x 4 x 4 pixels (cfg3 merged pass, 4x4 level): wgrad_kernel<64, 64, 2, 2>
    x, dy = torch.randn(128, 4, 4, 256, device=dev), torch.randn(128, 4, 4, 256, device=dev)
    gw = torch.zeros(256, 256, 3, 3, device=dev)
    fn = lambda: ops.conv_wgrad_(ops.CONV3X3_S1, x, dy, gw, c_real=256, cw=256, c_off=0)
elif case == "gnbig":          # large-slab GroupNorm apply pass: 8 x 256 x 256 x 128 (cfg5 top level)
    x = torch.randn(8, 256, 256, 128, device=dev)
    gam, bet, temb = torch.ones(128, device=dev), torch.zeros(128, device=dev), torch.randn(8, 128, device=dev)
    fn = lambda: ops.groupnorm_mish(x, gam, bet, temb=temb)
elif case == "cluster32":       # conv3x3 128->128 @32x32 with GroupNorm finished in the launch (8 workgroups per cluster)
    x, w = torch.randn(B, 32, 32, 128, device=dev), rw(128, 128)
    wu, b = ops.pack_conv_weight_wino(w), torch.zeros(128, device=dev)
    gam, bet, temb = torch.ones(128, device=dev), torch.zeros(128, device=dev), torch.randn(B, 128, device=dev)
    fn = lambda: ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet, temb=temb, check=False)
elif case == "wino16":          # conv3x3 256->256 @16x16 (the 64-channel tile of the 8-matrix-wave kernel)
    x, w = torch.randn(B, 16, 16, 256, device=dev), rw(256, 256)
    wp, wu, b = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w), torch.zeros(256, device=dev)
    fn = lambda: ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu)
elif case == "convT":          # ConvTranspose2d 4x4 s2 128 ch 16x16 -> 32x32 (ups.2.3 at cfg4) as Winograd F(2x2,2x2) per phase
    x, w = torch.randn(B, 16, 16, 128, device=dev), torch.randn(128, 128, 4, 4, device=dev) * (128 * 4) ** -0.5
    wp, wu, b = ops.pack_convT_weight(w), ops.pack_convT_weight_wino(w), torch.zeros(128, device=dev)
    fn = lambda: ops.conv(ops.CONVT4X4_S2, x, wp, b, w_wino=wu)
elif case == "c32":            # conv3x3 32->32 @64x64, 64 images + second output Mish(out): the dDDPM encoder / decoder conv of cfg3 training
    x, w = torch.randn(64, 64, 64, 32, device=dev), rw(32, 32)
    wp, b, ao = ops.pack_conv_weight(w), torch.zeros(32, device=dev), torch.empty(64, 64, 64, 32, device=dev)
    fn = lambda: ops.conv(ops.CONV3X3_S1, x, wp, b, mish_out=ao)
elif case == "stream":         # conv1x1 32->64 @64x64, 64 images, x Mish'(src) + residual: the c1 input-gradient conv of the dDDPM blocks (235 MB)
    x, w = torch.randn(64, 64, 64, 32, device=dev), rw(64, 32, 1)
    wp, b = ops.pack_conv_weight(w), torch.zeros(64, device=dev)
    src, res = torch.randn(64, 64, 64, 64, device=dev), torch.randn(64, 64, 64, 64, device=dev)
    fn = lambda: ops.conv(ops.CONV1X1, x, wp, b, dmish_src=src, resid=res)
elif case == "cluster16":
    x, w = torch.randn(B, 16, 16, 256, device=dev), rw(256, 256)
    wu, b = ops.pack_conv_weight_wino(w), torch.zeros(256, device=dev)
    gam, bet, temb = torch.ones(256, device=dev), torch.zeros(256, device=dev), torch.randn(B, 256, device=dev)
    fn = lambda: ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet, temb=temb, check=False)
elif case == "cls16":          # conv3x3 128->128 @16x16, k split over two workgroups per tile, partner tile summed + GroupNorm in the launch (round 6)
    x, w = torch.randn(B, 16, 16, 128, device=dev), rw(128, 128)
    wu, b = ops.pack_conv_weight_wino(w), torch.zeros(128, device=dev)
    gam, bet, temb = torch.ones(128, device=dev), torch.zeros(128, device=dev), torch.randn(B, 128, device=dev)
    fn = lambda: ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet, temb=temb, check=False)
elif case == "unet":           # the whole cfg4 UNet forward, in-launch paths on: counters of kernels that only exist inside it (level_chain_kernel)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import det_state, unet_cfg
    from models import Unet
    cfg = unet_cfg(128, 8)
    net = Unet(cfg)
    net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
    net = net.to(dev).eval()
    xin = torch.randn(B, 8, 32, 32, device=dev)
    tt = (torch.arange(B, device=dev) * 31) % 1000
    net.plan().set_option(net.plan().OPT_CLUSTER_GROUPNORM, 2)
    def fn():
        with torch.no_grad():
            return net(xin, tt)
else:
    raise SystemExit(f"unknown case {case}")
for _ in range(20):
    y = fn()
torch.cuda.synchronize()
print("ok", case)
