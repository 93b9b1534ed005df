#!/usr/bin/env python3
"""diagnostic: the k-split in-launch GroupNorm form against the two-launch path, which elements differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops
B, H, C, N = 32, 16, 128, 128
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, H, C, generator=g).cuda()
w = (torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda()
gam, bet = torch.ones(N).cuda(), torch.zeros(N).cuda()
wu, wp = ops.pack_conv_weight_wino(w), ops.pack_conv_weight(w)
raw = ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu)
two = ops.groupnorm_mish(raw, gam, bet)
for k in range(4):
    out = ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet)
    d = (out - two).abs()
    bad = d > 1e-4
    print("call", k, "max diff", d.max().item(), "bad elements", int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print("  images", sorted(set(idx[:, 0].tolist()))[:16], "rows", sorted(set(idx[:, 1].tolist())), "channel blocks", sorted(set((idx[:, 3] // 16).tolist())))
        big = d > 0.1
        bi = big.nonzero()
        print("  big", int(big.sum()), bi[:12].tolist())

lib = ops.L.load()
plain = lib.ddk_conv3x3_gn_mish_cluster_workspace_bytes(B, H, H, N) // 4
ws = ops._scratch[(str(x.device), "cluster", torch.cuda.current_stream().cuda_stream)]
pair_words = 64 * 2 * 16
slab = ws.view(torch.float32).view(-1)[plain + pair_words: plain + pair_words + B * H * H * N].view(B, H, H, N)
_, slabs = ops.conv(ops.CONV3X3_S1, x, wp, None, w_wino=wu, leave_slabs=True)
torch.cuda.synchronize()
d = (slab - slabs[1]).abs()
print("partner slab vs the split conv's slab 1: max diff", d.max().item(), "bad", int((d > 1e-5).sum()))
pairs = ws.view(torch.int32).view(-1)[plain: plain + pair_words]
print("pair counters nonzero after the launches:", int((pairs != 0).sum()))
