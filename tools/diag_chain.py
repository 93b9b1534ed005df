import os, sys
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from helpers import det_state, unet_cfg
from models import Unet
from utils import synthetic as syn
from ddk import ops
DEV = "cuda"
cfg = unet_cfg(128, 8)
net = Unet(cfg)
net.load_state_dict(det_state({k: v.shape for k, v in net.state_dict().items()}))
net = net.to(DEV).eval()
for batch in (8, 16, 24, 31, 32):
    x = syn.synthetic_normal((batch, 8, 32, 32), f"chain.x{batch}").to(DEV)
    t = (torch.arange(batch, device=DEV) * 23) % 1000
    with torch.no_grad():
        plan = net.plan()
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 2)
        for mode in (8, 16):
            plan.set_option(plan.OPT_LEVEL_CHAIN, 0)
            y_off = net(x, t)
            plan.set_option(plan.OPT_LEVEL_CHAIN, mode)
            ys = [net(x, t) for _ in range(3)]
            d_off = [(y - y_off).abs().amax(dim=(1, 2, 3)).cpu() for y in ys]
            print(f"B={batch} mode={mode} timeouts={ops.cluster_timeouts()} cluster={plan._cluster}")
            for i, d in enumerate(d_off):
                bad = (d > 1e-4).nonzero().flatten().tolist()
                print(f"   run {i}: max diff vs off {float(d.max()):.3e}; images off by > 1e-4: {bad}")
