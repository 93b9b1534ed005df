#!/usr/bin/env python3
"""Diagnostic for the in-launch GroupNorm: consecutive launches on DIFFERENT inputs (a stale record of the previous launch would
show), compared with the two-launch path; and the UNet with the option on / off / on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from ddk import ops  # noqa: E402

dev = "cuda"
B, H, C, N = 32, 32, 128, 128
w = torch.randn(N, C, 3, 3, device=dev) * (C * 9) ** -0.5
wu, wp = ops.pack_conv_weight_wino(w), ops.pack_conv_weight(w)
b = torch.randn(N, device=dev)
gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
bad = 0
for it in range(20):
    x = torch.randn(B, H, H, C, device=dev) * (1 + it)
    out = ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet)
    raw, part, tiles = ops.conv_with_gn_partials(x, wp, b, wu)
    two = ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet)
    d = (out - two).abs().max().item()
    if d != 0:
        bad += 1
        print(f"iter {it}: max |cluster - two-launch| = {d:.3e}")
print(f"kernel level: {bad} of 20 launches differ; timeouts {ops.cluster_timeouts()}")

from helpers import det_state, unet_cfg  # noqa: E402
from models import Unet  # noqa: E402
from utils import synthetic as syn  # noqa: E402
cfg = unet_cfg(128, 8)
net = Unet(cfg)
net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
net = net.to(dev).eval()
x = syn.synthetic_normal((32, 8, 32, 32), "cluster.x").to(dev)
t = torch.arange(32, device=dev) * 31
with torch.no_grad():
    plan = net.plan()
    plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 0)
    y0 = net(x, t)
    plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 1)
    for lim in range(0, 8):
        plan.set_option(2, lim)
        y = net(x, t)
        print(f"first {lim} eligible launches clustered: max |y - y_off| = {(y - y0).abs().max().item():.3e}")
print("timeouts", plan.cluster_timeouts())
