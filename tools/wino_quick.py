#!/usr/bin/env python3
"""Kernel-only (graph replay) timing of conv3x3_wino_kernel on the step's Winograd shapes (B = 32).  DDK_LIB selects the library
build, so variants of the kernel can be compared inside one GPU call:  DDK_LIB=.../libddk_x.so python tools/wino_quick.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
os.environ.setdefault("DDK_LIB", os.path.join(ROOT, "downsampled-diffusion_amd", "csrc", "libddk.so"))   # conv_sweep defaults to the tuning build
import torch  # noqa: E402
from ddk import ops  # noqa: E402
from conv_sweep import graph_time  # noqa: E402

SHAPES = [("128->128 @32", 32, 128, 0, 128, 5), ("128->256 @16", 16, 128, 0, 256, 1), ("256->256 @16", 16, 256, 0, 256, 3),
          ("512->128 @16", 16, 256, 256, 128, 1), ("128->128 @16", 16, 128, 0, 128, 3), ("512->256 @8", 8, 256, 256, 256, 1)]
B = 32
tot = 0.0
for name, H, c0, c1, N, count in SHAPES:
    cin = c0 + c1
    x0 = torch.randn(B, H, H, c0, device="cuda")
    x1 = torch.randn(B, H, H, c1, device="cuda") if c1 else None
    w = torch.randn(N, cin, 3, 3, device="cuda") * (cin * 9) ** -0.5
    wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w)
    bias = torch.zeros(N, device="cuda")
    ref = ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1)
    out = ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1, w_wino=wu)
    err = float((ref - out).abs().max() / ref.abs().max())
    tw = graph_time(lambda: ops.conv(ops.CONV3X3_S1, x0, wp, bias, x2=x1, w_wino=wu))
    ex = 2.0 * B * (H * H / 4) * 16 * cin * N
    tot += tw * count
    print(f"{name:14s} x{count} {tw:6.2f} us  {ex / tw / 1e6:5.1f} TF executed ({ex / tw / 1e6 / 157.3:.3f} of peak)  rel err vs direct {err:.1e}", flush=True)
print(f"sum over the step's {sum(s[5] for s in SHAPES)} launches: {tot:.1f} us   [{os.environ.get('DDK_LIB', 'libddk.so')}]")

# ---- the Block (conv + GroupNorm + Mish + shift): conv with tile statistics + GroupNorm-apply launch, against the in-launch GroupNorm
print("Block = conv3x3 + GroupNorm + Mish + time shift: two launches (conv with statistics, apply) vs one (cluster exchange)")
for name, H, C, N in [("128->128 @32", 32, 128, 128), ("256->256 @16", 16, 256, 256), ("128->256 @16", 16, 128, 256)]:
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(N, C, 3, 3, device="cuda") * (C * 9) ** -0.5
    wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w)
    bias, gam, bet, temb = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), torch.randn(B, N, device="cuda")

    def two():
        raw, part, tiles = ops.conv_with_gn_partials(x, wp, bias, wu)
        return ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet, temb=temb)
    t2 = graph_time(two)
    if ops.L.load().ddk_conv3x3_gn_mish_cluster_ok(B, H, H, C, N, 8) > 0:
        one = lambda: ops.conv3x3_gn_mish_cluster(x, wu, bias, gam, bet, temb=temb, check=False)
        t1 = graph_time(one)
        err = float((one() - two()).abs().max())
        ops.cluster_check(x.device, B)     # the unchecked eager call above used this stream's scratch
        print(f"{name:14s} two launches {t2:6.2f} us   one launch {t1:6.2f} us   max |diff| {err:.1e}", flush=True)
    else:
        print(f"{name:14s} two launches {t2:6.2f} us   (in-launch GroupNorm not eligible)", flush=True)

# ---- transpose conv 4x4 stride 2 (Upsample): Winograd F(2x2,2x2) per phase against the direct im2col kernel
print("ConvTranspose2d 4x4 s2 p1: direct im2col (+ slab reduce) vs Winograd F(2x2,2x2) per phase")
for name, H, C in [("128 @16->32", 16, 128), ("256 @8->16", 8, 256), ("256 @4->8", 4, 256)]:
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(C, C, 4, 4, device="cuda") * (C * 4) ** -0.5
    wp, wu, bias = ops.pack_convT_weight(w), ops.pack_convT_weight_wino(w), torch.zeros(C, device="cuda")
    ref = ops.conv(ops.CONVT4X4_S2, x, wp, bias)
    out = ops.conv(ops.CONVT4X4_S2, x, wp, bias, w_wino=wu)
    err = float((ref - out).abs().max() / ref.abs().max())
    td = graph_time(lambda: ops.conv(ops.CONVT4X4_S2, x, wp, bias))
    tw = graph_time(lambda: ops.conv(ops.CONVT4X4_S2, x, wp, bias, w_wino=wu))
    ex = 2.0 * B * (H * H / 4) * 36 * C * C
    print(f"{name:14s} direct {td:6.2f} us   Winograd {tw:6.2f} us  ({ex / tw / 1e6:5.1f} TF executed, {ex / tw / 1e6 / 157.3:.3f} of peak)  rel err {err:.1e}", flush=True)
