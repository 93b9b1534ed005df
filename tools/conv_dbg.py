#!/usr/bin/env python3
"""Ablation timing of the 128->128 @32 B=32 conv kernel under DDK_DEBUG masks (results are wrong by design)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
    import torch
    from ddk import ops
    B, H, W, C, N = 32, 32, 32, 128, 128
    x = torch.randn(B, H, W, C, device="cuda"); wp = ops.pack_conv_weight(torch.randn(N, C, 3, 3, device="cuda") * 0.03); b = torch.zeros(N, device="cuda")
    for _ in range(5): ops.conv(ops.CONV3X3_S1, x, wp, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.conv(ops.CONV3X3_S1, x, wp, b)
    e1.record(); torch.cuda.synchronize()
    print(f"DEBUG={os.environ.get('DDK_DEBUG','0'):>2s} TILE={os.environ.get('DDK_FORCE_TILE','auto'):>5s}: {e0.elapsed_time(e1)/30*1e3:7.1f} us")
else:
    for tile in ("0,1", "1,1"):
        for dbg in (0, 8, 12, 28):
            env = dict(os.environ, DDK_DEBUG=str(dbg), DDK_FORCE_TILE=tile)
            subprocess.run([sys.executable, __file__, "run"], env=env)
