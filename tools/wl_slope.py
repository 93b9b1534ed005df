import os, sys
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from ddk import ops
from conv_sweep import graph_time
B = 32
for cin in (32, 64, 128, 256, 320):
    x0 = torch.randn(B, 8, 8, cin, device="cuda")
    w = torch.randn(256, cin, 3, 3, device="cuda") * (cin * 9) ** -0.5
    wwl = ops.pack_conv_weight_wino_local(w)
    bias = torch.zeros(256, device="cuda"); gam = torch.ones(256, device="cuda"); bet = torch.zeros(256, device="cuda")
    t = graph_time(lambda: ops.conv3x3_gn_mish_wino(x0, wwl, bias, gam, bet))
    print(f"cin {cin:4d} ({cin // 32:2d} chunks): {t:6.2f} us", flush=True)
