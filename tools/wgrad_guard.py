#!/usr/bin/env python3
"""Does any weight-gradient launch write past the workspace size ddk_conv_wgrad_workspace_bytes reports?  Every call gets a buffer of
exactly that size followed by a guard region of sentinels.   python tools/wgrad_guard.py [B]"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from ddk import lib as L, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = L.load()
GUARD = 1 << 18          # floats
bad = 0
kinds = {"s1": ops.CONV3X3_S1, "1x1": ops.CONV1X1, "s2": ops.CONV3X3_S2, "4x4s2": 4}
for H in (256, 128, 64, 32, 16, 8):
    for name, kind in kinds.items():
        for cx, N in itertools.product((32, 128, 256), (32, 128, 256, 384)):
            if (long := B * H * H * max(cx, N) * 4) > 3 << 30:
                continue
            Ho = H // 2 if name in ("s2", "4x4s2") else H
            k = {"s1": 3, "1x1": 1, "s2": 3, "4x4s2": 4}[name]
            nbytes = lib.ddk_conv_wgrad_workspace_bytes(kind, B, H, H, cx, N)
            if nbytes == 0:
                continue
            x = torch.randn(B, H, H, cx, device="cuda")
            dy = torch.randn(B, Ho, Ho, N, device="cuda")
            for with_b in (False, True):
                buf = torch.full((nbytes // 4 + GUARD,), 12345.0, device="cuda")
                gw = torch.zeros(N, cx, k, k, device="cuda")
                gb = torch.zeros(N, device="cuda") if with_b else None
                rc = lib.ddk_conv_wgrad_bias(kind, L.ptr(x), L.ptr(dy), L.ptr(gw), L.ptr(gb), B, H, H, cx, cx, cx, 0, N, L.ptr(buf), nbytes, L.stream())
                torch.cuda.synchronize()
                if rc != 0:
                    print(f"{name} {H}x{H} cx={cx} N={N} bias={with_b}: rc {rc} {L.last_error()}")
                    continue
                g = buf[nbytes // 4:]
                n_bad = int((g != 12345.0).sum())
                if n_bad:
                    bad += 1
                    first = int((g != 12345.0).nonzero()[0])
                    print(f"OVERRUN {name} {H}x{H} B={B} cx={cx} N={N} bias={with_b}: {n_bad} guard floats changed, first at +{first} (workspace {nbytes} B)", flush=True)
print("shapes with an overrun:", bad)
