import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import ops
dev = "cuda"
B = 32
for (H, C, N) in ((32, 128, 128), (16, 128, 256), (16, 256, 256)):
    w = torch.randn(N, C, 3, 3, device=dev) * (C * 9) ** -0.5
    wu, wp = ops.pack_conv_weight_wino(w), ops.pack_conv_weight(w)
    b = torch.randn(N, device=dev)
    gam, bet = 1 + 0.1 * torch.randn(N, device=dev), 0.1 * torch.randn(N, device=dev)
    x = torch.randn(B, H, H, C, device=dev)
    temb = torch.randn(B, 3584, device=dev)[:, 128:128 + N]
    add = torch.randn(B, H, H, N, device=dev)
    for name, kw in (("none", {}), ("temb", dict(temb=temb)), ("add", dict(addend=add)), ("both", dict(temb=temb, addend=add))):
        out = ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet, **kw)
        raw, part, tiles = ops.conv_with_gn_partials(x, wp, b, wu)
        two = ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet, **kw)
        print(H, C, N, name, (out - two).abs().max().item())
