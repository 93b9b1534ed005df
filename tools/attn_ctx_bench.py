#!/usr/bin/env python3
"""Kernel-only time (graph replay) of the split linear-attention context (+ merge) at cfg4's 32x32 level for the folded block:
rows are [k | v] of 4 heads.  Tuning build: DDK_LINATTN_WGS caps the workgroup count (splits = cap / (B heads))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from ddk import ops, lib as L
from encdec_bench import graph_us
B, HW = 32, 1024
kv = torch.randn(B, HW, 256, device="cuda")
lib = L.load()
ctx = torch.empty(B, 4, 32, 32, device="cuda")
ws_bytes = lib.ddk_linattn_context_workspace_bytes(B, HW, 4)
ws = torch.empty(max(ws_bytes, 16) // 4, device="cuda")
fn = lambda: L.check(lib.ddk_linattn_context_kv(L.ptr(kv), L.ptr(ctx), B, HW, 4, L.ptr(ws), ws_bytes, L.stream()), "ctx")
print(os.environ.get("DDK_LINATTN_WGS", "default"), f"context (+ merge): {graph_us(fn):.1f} us, workspace {ws_bytes} B")
