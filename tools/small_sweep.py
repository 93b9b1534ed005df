import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from ddk import ops
from gn_bench import graph_time
SH = [("3x3 256->256 @4", ops.CONV3X3_S1, 32, 4, 4, 256, 256), ("3x3 256->256 @8", ops.CONV3X3_S1, 32, 8, 8, 256, 256),
      ("3x3 128->128 @16", ops.CONV3X3_S1, 32, 16, 16, 128, 128), ("1x1 256->384 @4", ops.CONV1X1, 32, 4, 4, 256, 384),
      ("1x1 256->384 @8", ops.CONV1X1, 32, 8, 8, 256, 384), ("1x1 128->256 @8", ops.CONV1X1, 32, 8, 8, 128, 256)]
for name, kind, B, H, W, C, N in SH:
    x = torch.randn(B, H, W, C, device="cuda")
    k = 1 if kind == ops.CONV1X1 else 3
    wp = ops.pack_conv_weight(torch.randn(N, C, k, k, device="cuda") * 0.02)
    bias = torch.zeros(N, device="cuda")
    fl = 2.0 * B * H * W * k * k * C * N
    os.environ.pop("DDK_FORCE_TILE", None)
    line = f"{name:18s} auto {graph_time(lambda: ops.conv(kind, x, wp, bias)):6.1f}us |"
    for t in (2, 4):
        for s_ in (1, 2, 4, 8, 16):
            os.environ["DDK_FORCE_TILE"] = f"{t},{s_}"
            try:
                line += f" t{t}/{s_}:{graph_time(lambda: ops.conv(kind, x, wp, bias), n=20, reps=3):5.1f}"
            except Exception as e:
                line += f" t{t}/{s_}: err"
    os.environ.pop("DDK_FORCE_TILE", None)
    print(line, flush=True)
