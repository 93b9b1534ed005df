#!/usr/bin/env python3
"""Launch time of attn_kvctx_kernel (+ merge) at its cfg4 shape under one library (DDK_LIB): graph replay of 50 launches.
Used with the ablation builds of tools/kvctx_abl.sh to see where the kernel's time goes.  python tools/kvctx_abl.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT]
import torch
from bench import graph_kernel_seconds
from ddk import ops

dev = torch.device("cuda", 0)
B = 32
x = torch.randn(B, 32, 32, 128, device=dev)
wq, g, be = torch.randn(384, 128, device=dev) * 128 ** -0.5, torch.ones(128, device=dev), torch.zeros(128, device=dev)
hc = 128
wkv = (wq * g.view(1, 128))[hc:].contiguous()
c1, c2 = (wq @ g)[hc:].contiguous(), (wq @ be)[hc:].contiguous()
L = ops.L
lib = L.load()
ctx = torch.empty((B, 4, 32, 32), device=dev)
nbytes = lib.ddk_attention_kv_context_workspace_bytes(B, 1024)
ws = torch.empty(nbytes // 4 + 4, device=dev)


def fn():
    L.check(lib.ddk_attention_kv_context(L.ptr(x), L.ptr(wkv), L.ptr(c1), L.ptr(c2), 1e-5, L.ptr(ctx), B, 1024, L.ptr(ws), nbytes, L.stream()), "kvctx")


sec = graph_kernel_seconds(dev, fn)
print(f"{os.environ.get('DDK_LIB', 'libddk.so'):60s} kv+context + merge: {sec * 1e6:6.2f} us per call", flush=True)
