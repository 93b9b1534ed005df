#!/usr/bin/env python3
"""Benchmark of the DDPM sampling hot path on MI355X (BASELINE.json metric: images/sec, T=1000 sampling).

Workload (N=1 and per GPU for N>1): cfg4 = CelebAMask-HQ 256x256 dDDPM-x3 -- batch 32 latents of 8x32x32,
full-width UNet (unet_chan 128, dims (1,2,2,2)), linear schedule T=1000, then the x3 ConvResNet decoder +
tanh to 3x256x256.  A "step" is one reverse step (UNet forward + fused x update) over the batch, run by the
native hipGraph sampler; images/sec = B*N / (T * t_step + t_decode), with t_decode measured in the same run.
Synthetic: closed-form weights (utils/synthetic.py), x_T and per-step noise from the in-kernel Philox stream.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            # no launcher: this file starts the N rank processes itself (launch_ranks)
    python bench.py --gpus N --train-dp     # data-parallel training line: cfg5 optimiser step with the C2 gradient all-reduce

Rank 0 prints ONE JSON line on stdout; diagnostics go to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "downsampled-diffusion_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

torch = None   # imported by main() AFTER the launcher decision: the parent of `--gpus N` never loads torch, HIP or libddk


def _import_torch():
    global torch
    import torch as _t
    torch = _t
    return _t


if __name__ != "__main__":     # imported for its helpers (tools/*.py): no launcher decision to wait for
    _import_torch()

FP32_PEAK_TFLOPS = 157.3      # MI355X fp32 matrix = vector peak (MI355X_MICROARCH.md, chip-level parameters)
T_STEPS = 1000
WINO_PMC = "r04_wino_pmc.json"      # committed counter summary of the timed Winograd kernel (tools/pmc_summary.py)
CLUSTER_PMC = "r04_wino_cluster32_pmc.json"   # ... of its in-launch-GroupNorm variant (three launches per step at 32x32)
WINO_SRC = "downsampled-diffusion_amd/csrc/conv_wino.hip+downsampled-diffusion_amd/csrc/conv_wino2_kernel.inc"   # what its hash covers
HBM_PMC = "r05_gn_pmc.json"         # ... of the GroupNorm-apply kernel (FETCH_SIZE / WRITE_SIZE passes)
LOCAL_PMC = "r06_wlocal8_pmc.json"  # ... of the image-local conv + GroupNorm kernel (8x8 maps)
STREAM_PMC = "r04_stream_pmc.json"  # ... of the streaming 1x1 conv of the dDDPM encoder / decoder blocks


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cfg4():
    return dict(unet_chan=128, unet_in=8, unet_dims=(1, 2, 2, 2), unet_dropout=0.1, image_size=256, T=T_STEPS,
                loss_type="simple", beta_schedule="linear", loss_flat="sum", d_mode="convolutional_res",
                u_mode="convolutional_res", d_dropout=0, d_chans=64, d_n_blocks=3, u_n_blocks=3, ae_loss=True,
                t_rec_max=100, force_latent=True, n_downsamples=3, dataset="celeba_hq", model="dddpm")


HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6300 GB/s achievable with a float4 copy)


def graph_kernel_seconds(device, fn, n=50, reps=4):
    """Average duration of ONE launch of `fn`'s kernel(s): n launches captured into a device graph and replayed, exactly how
    the sampler issues them (back to back, no host in between), timed with HIP events recorded on the launch stream.
    Launched one by one from the host each launch would carry a ~9 us dispatch gap that is not the kernel's."""
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        fn()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(n):
                fn()
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            graph.replay()
        e1.record(side)
        side.synchronize()
    return e0.elapsed_time(e1) / 1e3 / (n * reps)


def _sha16(rel):
    """sha256 (first 16 hex digits) of one source file, or of several '+'-separated ones in the order given"""
    import hashlib
    h = hashlib.sha256()
    try:
        for part in rel.split("+"):
            with open(os.path.join(ROOT, part), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def _profile_json(name, kernel_source=None):
    """A committed rocprofv3 --pmc summary (profiles/*.json).  Counters cannot be collected inside a timing loop, so they are
    taken in separate passes (tools/pmc_summary.py) and the summary records the sha256 of the kernel source they were measured
    on: when the kernel file has changed since, the figures are STALE and are not reported."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            pm = json.load(f)
    except Exception:   # noqa: BLE001 -- the profile file is optional evidence, never a reason to fail the bench
        return None
    if kernel_source and pm.get("kernel_source_sha16") != _sha16(kernel_source):
        log(f"profiles/{name}: measured on another version of {kernel_source}; counters not reported")
        return None
    return pm


def time_conv_roofline(device):
    """Dominant kernel of the step: the 3x3 conv 128->128 @32x32, batch 32, which the sampler runs as conv3x3_wino2_kernel
    (Winograd F(2x2,3x3) on the fp32 matrix pipe).  Launch duration is measured live (HIP events on the launch stream, graph
    replay).  `achieved` / `frac` price the MFMA FLOPs the kernel ISSUES -- 16 multiplies per 2x2 output tile and input channel,
    2 * (B*H*W/4) * 16 * Cin * Cout = 16/36 of the direct algorithm's -- against the fp32 MFMA peak, so frac <= 1 by
    construction.  The direct-conv figure of SURVEY.md section 8d (2*B*H*W*9*Cin*Cout) is reported separately as
    `algorithmic_equiv_tflops`: a statement about the algorithm, not the chip.  `mfma_busy` and `traffic` come from committed
    rocprofv3 --pmc passes on the same kernel and shape, dropped when the kernel source has changed since."""
    from ddk import ops
    B, H, W, C, N = 32, 32, 32, 128, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).to(device)
    w = (torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5).to(device)
    b = torch.zeros(N, device=device)
    wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w)
    for _ in range(5):
        ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu)
    torch.cuda.synchronize()
    sec = graph_kernel_seconds(device, lambda: ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu))
    sec_direct = graph_kernel_seconds(device, lambda: ops.conv(ops.CONV3X3_S1, x, wp, b), n=20, reps=2)
    flops = 2.0 * B * H * W * 9 * C * N
    executed = 2.0 * (B * H * W / 4) * 16 * C * N
    bytes_alg = 4.0 * (B * H * W * C + B * H * W * N + N * 9 * C)
    pm = _profile_json(WINO_PMC, WINO_SRC)
    src = f"profiles/{WINO_PMC} (rocprofv3 --pmc, separate passes: SQ_* | FETCH_SIZE x2 | WRITE_SIZE)" if pm else None
    return dict(kernel="conv3x3_wino2_kernel<2> conv3x3 128->128 @32x32 B=32 (Winograd F(2x2,3x3), fp32 MFMA, 8 matrix + 4 loader waves, "
                       "32 tiles x 128 channels per workgroup, one dispatch round)",
                bound="mfma", achieved=executed / sec / 1e12, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=executed / sec / 1e12 / FP32_PEAK_TFLOPS,
                traffic=pm["traffic_bytes_per_launch"] if pm else None, traffic_source=src,
                mfma_busy=pm["mfma_busy"] if pm else None, mfma_busy_source=src,
                launch_us=sec * 1e6, executed_gflop=executed / 1e9, algorithmic_gflop=flops / 1e9,
                algorithmic_equiv_tflops=flops / sec / 1e12, algorithmic_speedup_vs_direct_kernel=sec_direct / sec,
                algorithmic_mbytes=bytes_alg / 1e6, algorithmic_hbm_gbps=bytes_alg / sec / 1e9,
                note="achieved = MFMA FLOPs issued / launch time (<= peak); algorithmic_equiv_tflops prices the direct algorithm's "
                     "9.664 GFLOP on the same time and may exceed the peak",
                direct_kernel={"kernel": "conv3x3_halo_kernel<0> (direct implicit GEMM, same shape; used when a shape is not Winograd-eligible)",
                               "launch_us": sec_direct * 1e6, "achieved": flops / sec_direct / 1e12, "frac": flops / sec_direct / 1e12 / FP32_PEAK_TFLOPS})


def time_local_roofline(device):
    """Second-largest kernel family of the step (19 of its launches): conv3x3 + GroupNorm + Mish (+ shift, + residual) in ONE launch on
    the small maps -- here the 8x8 form (conv3x3_gn_wlocal_kernel, Winograd inside an image-local tiling) at its cfg4 shape
    256 -> 256, batch 32.  Priced like the dominant kernel: MFMA FLOPs issued (16 multiplies per 2x2 tile and channel pair) over the
    live launch time against the fp32 MFMA peak."""
    from ddk import ops
    B, H, C, N = 32, 8, 256, 256
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B, H, H, C, generator=g).to(device)
    w = (torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5).to(device)
    gam, bet, b = torch.ones(N, device=device), torch.zeros(N, device=device), torch.zeros(N, device=device)
    temb = torch.randn(B, N, generator=g).to(device)
    wl = ops.pack_conv_weight_wino_local(w)
    sec = graph_kernel_seconds(device, lambda: ops.conv3x3_gn_mish_wino(x, wl, b, gam, bet, temb=temb))
    executed = 2.0 * B * (H * H / 4) * 16 * C * N
    pm = _profile_json(LOCAL_PMC, "downsampled-diffusion_amd/csrc/conv_local.hip")
    src = f"profiles/{LOCAL_PMC} (rocprofv3 --pmc, separate passes)" if pm else None
    return dict(kernel="conv3x3_gn_wlocal_kernel: conv3x3 256->256 @8x8 B=32 + GroupNorm + Mish + time shift in one launch",
                bound="mfma", achieved=executed / sec / 1e12, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=executed / sec / 1e12 / FP32_PEAK_TFLOPS,
                traffic=pm["traffic_bytes_per_launch"] if pm else None, traffic_source=src,
                mfma_busy=pm["mfma_busy"] if pm else None, mfma_busy_source=src, launch_us=sec * 1e6, executed_gflop=executed / 1e9,
                algorithmic_gflop=2.0 * B * H * H * 9 * C * N / 1e9)


CHAIN_PMC = "r06_chain4_pmc.json"  # ... of the persistent 4x4-level kernel (csrc/level_chain.hip)


def time_chain_roofline(device, B=32):
    """The persistent 4x4-level kernel (csrc/level_chain.hip, `level_chain_kernel`): ONE launch walks the level's 23 ops -- the Downsample
    conv 8x8 -> 4x4, 2 + 2 + 2 ResnetBlocks (12 conv3x3 + GroupNorm + Mish, one on a 512-channel concat, and its 1x1 skip conv), three
    attention blocks (LayerNorm-folded to_qkv + attention core per head, then to_out), the Upsample transpose conv 4x4 -> 8x8 -- for
    `B` images on 8 x B workgroups that hand the 16-pixel images to each other through memory.  Timed live at the cfg4 shape (256
    channels) on synthetic weights through the library's op-list entry (ddk_debug_level_chain: the same kernel, the same op list as
    unet_plan.hip builds); priced on the MFMA FLOPs it ISSUES (direct products, padding taps included) against the fp32 MFMA peak."""
    import ctypes as C
    from ddk import lib, ops
    L = lib.load()

    class Op(C.Structure):
        _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p),
                    ("beta", C.c_void_p), ("out", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int), ("kind", C.c_int), ("flags", C.c_int),
                    ("temb_off", C.c_int), ("n_out", C.c_int)]
    CONV3, CONV1, ATTN, UPT = 0, 1, 2, 3
    WAIT, SIGNAL, ADDK, SAVEK, ADDK2, SAVEK2, NO_OUT, NO_GN, DOWN = 1, 2, 4, 8, 16, 32, 128, 256, 512
    g = torch.Generator(device="cpu").manual_seed(11)
    keep = []

    def t(*shape, scale=1.0):
        v = (torch.randn(*shape, generator=g) * scale).to(device)
        keep.append(v)
        return v

    def buf(hw=16, c=256):
        v = torch.empty((B, hw, c), device=device)
        keep.append(v)
        return v
    p = lambda v: None if v is None else v.data_ptr()   # noqa: E731
    temb = t(B, 6 * 256)
    opsl, flop = [], 0.0

    def conv3(src, src1, cin, temb_off, flags, out, down=False):
        nonlocal flop
        w = ops.pack_conv_weight_local(t(256, cin, 3, 3, scale=(cin * 9) ** -0.5))
        keep.append(w)
        opsl.append(Op(p(src), p(src1), p(w), p(t(256, scale=0.1)), p(1 + t(256, scale=0.1)), p(t(256, scale=0.1)), p(out), min(cin, 256),
                       cin - min(cin, 256), CONV3, flags | (DOWN | NO_GN if down else 0), temb_off, 256))
        flop += 2.0 * B * 16 * 9 * cin * 256

    def conv1(src, src1, cin, flags, out):
        nonlocal flop
        w = torch.empty(256 * cin, device=device)
        wc = t(256, cin, scale=cin ** -0.5)
        lib.check(L.ddk_pack_conv1x1_weight_local(lib.ptr(wc), lib.ptr(w), 256, cin, cin, lib.stream()), "pack1")
        keep.append(w)
        opsl.append(Op(p(src), p(src1), p(w), p(t(256, scale=0.1)), None, None, p(out), min(cin, 256), cin - min(cin, 256), CONV1, flags, -1, 256))
        flop += 2.0 * B * 16 * cin * 256

    def attn_block(src, out, signals):
        nonlocal flop
        lnw = t(384, 256, scale=256 ** -0.5)
        wop = torch.empty_like(lnw)
        lib.check(L.ddk_pack_qkv_operand(lib.ptr(lnw), lib.ptr(wop), 4, 256, lib.stream()), "pack_qkv_operand")
        keep.append(wop)
        heads = buf(16, 128)
        opsl.append(Op(p(src), None, p(wop), None, p(t(384)), p(t(384)), p(heads), 256, 0, ATTN, WAIT | SIGNAL, -1, 128))
        flop += 2.0 * B * 16 * 256 * 384 + 2 * 2.0 * B * 4 * 16 * 32 * 32
        conv1(heads, None, 128, (WAIT | SIGNAL if signals else WAIT) | ADDK | SAVEK, out)

    def res_plain(src, temb_off, out):
        h = buf()
        conv3(src, None, 256, temb_off, WAIT | SIGNAL, h)
        conv3(h, None, 256, -1, WAIT | SIGNAL | ADDK | SAVEK, out)
    x8 = t(B, 64, 256)
    dn, d0, d1, skip, m1, ma, m2, u0, u1, ua = (buf() for _ in range(10))
    up = buf(64, 256)
    conv3(x8, None, 256, -1, SIGNAL | SAVEK, dn, down=True)
    res_plain(dn, 0, d0)
    res_plain(d0, 256, d1)
    attn_block(d1, skip, True)
    res_plain(skip, 512, m1)
    attn_block(m1, ma, True)
    res_plain(ma, 768, m2)
    conv1(m2, skip, 512, WAIT | SAVEK2 | NO_OUT, None)
    h = buf()
    conv3(m2, skip, 512, 1024, SIGNAL, h)
    conv3(h, None, 256, -1, WAIT | SIGNAL | ADDK2 | SAVEK, u0)
    res_plain(u0, 1280, u1)
    attn_block(u1, ua, True)
    wt = t(256, 256, 4, 4, scale=(256 * 4) ** -0.5)
    wtp = torch.empty(256 * 16 * 256, device=device)
    lib.check(L.ddk_pack_convT_weight_local(lib.ptr(wt), lib.ptr(wtp), 256, 256, lib.stream()), "pack_convT_weight_local")
    opsl.append(Op(p(ua), None, p(wtp), p(t(256, scale=0.1)), None, None, p(up), 256, 0, UPT, WAIT, -1, 256))
    flop += 2.0 * B * 16 * 4 * 4 * 256 * 256
    arr = (Op * len(opsl))(*opsl)
    cnt = torch.zeros(64 * B + 16, device=device, dtype=torch.int32)
    L.ddk_debug_level_chain.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ddk_debug_level_chain.restype = C.c_int

    def fn():
        lib.check(L.ddk_debug_level_chain(C.cast(arr, C.c_void_p), len(opsl), 16, B, temb.data_ptr(), temb.shape[1], cnt.data_ptr(), lib.stream()),
                  "debug_level_chain")
    fn()
    torch.cuda.synchronize()
    sec = graph_kernel_seconds(device, fn, n=20, reps=3)
    torch.cuda.synchronize()
    if int(cnt[64 * B]) != 0 or not bool(torch.isfinite(up).all()):
        raise RuntimeError("level chain: a hand-off wait timed out during the timed launches (GPU shared?)")
    pm = _profile_json(CHAIN_PMC, "downsampled-diffusion_amd/csrc/level_chain.hip")
    src = f"profiles/{CHAIN_PMC} (rocprofv3 --pmc, separate passes, the kernel inside the whole UNet forward)" if pm else None
    return dict(kernel=f"level_chain_kernel: the whole 4x4 level of the cfg4 UNet ({len(opsl)} ops: Downsample conv, 12 conv3x3 + GroupNorm + Mish, "
                       "1x1 skip conv, 3 attention blocks, Upsample transpose conv) in ONE persistent launch, B=32 (replaces 23 launches)",
                bound="mfma", achieved=flop / sec / 1e12, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=flop / sec / 1e12 / FP32_PEAK_TFLOPS,
                traffic=pm["traffic_bytes_per_launch"] if pm else None, traffic_source=src,
                mfma_busy=pm["mfma_busy"] if pm else None, mfma_busy_source=src, launch_us=sec * 1e6, executed_gflop=flop / 1e9, ops=len(opsl))


def time_hbm_rooflines(device):
    """Secondary rooflines (HBM-bound kernels of the step), live: GroupNorm+Mish(+time shift) on the 32x32 128-channel
    tensor (8 B per element: one read, one write) -- the kernel on the step's path (statistics from the conv
    epilogue) and the register-resident one -- and the fused reverse-step update with in-kernel Philox noise (12 B per latent
    element: read x and eps_hat, write x)."""
    from ddk import ops
    out = []
    B, H, W, C = 32, 32, 32, 128
    x = torch.randn(B, H, W, C, device=device)
    gam, bet = torch.ones(C, device=device), torch.zeros(C, device=device)
    temb = torch.randn(B, C, device=device)
    w = torch.randn(C, C, 3, 3, device=device) * (C * 9) ** -0.5
    raw, part, tiles = ops.conv_with_gn_partials(x, ops.pack_conv_weight(w), torch.zeros(C, device=device), ops.pack_conv_weight_wino(w))
    sec = graph_kernel_seconds(device, lambda: ops.groupnorm_mish_from_partials(raw, part, tiles, gam, bet, temb=temb))
    nbytes = 8.0 * x.numel()
    pm = _profile_json(HBM_PMC, "downsampled-diffusion_amd/csrc/norm_act.hip")
    out.append(dict(kernel="gn_apply_parts_kernel GroupNorm(8)+Mish+time shift from the conv epilogue's per-tile statistics, 32x32x32x128",
                    bound="hbm", achieved=nbytes / sec / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s", frac=nbytes / sec / 1e9 / HBM_PEAK_GBPS,
                    traffic=pm["traffic_bytes_per_launch"] if pm else None,
                    traffic_source=f"profiles/{HBM_PMC} (rocprofv3 --pmc: FETCH_SIZE x2 | WRITE_SIZE)" if pm else None,
                    launch_us=sec * 1e6, algorithmic_mbytes=nbytes / 1e6))
    sec = graph_kernel_seconds(device, lambda: ops.groupnorm_mish(x, gam, bet, temb=temb))
    out.append(dict(kernel="gn_mish_resident_kernel<4,1024> (statistics + apply in one kernel; shapes whose conv splits channel chunks), "
                           "same tensor", bound="hbm", achieved=nbytes / sec / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=nbytes / sec / 1e9 / HBM_PEAK_GBPS, traffic=None, launch_us=sec * 1e6, algorithmic_mbytes=nbytes / 1e6))
    # the largest memory stream of the training path: the c1 input-gradient conv of a dDDPM encoder / decoder block at 64x64x64 pixels
    # (1x1 32 -> 64, x Mish'(src), + residual): reads dy [M][32], src and residual [M][64], writes [M][64]
    Bs, Ss = 64, 64
    xs1 = torch.randn(Bs, Ss, Ss, 32, device=device)
    ws1 = ops.pack_conv_weight(torch.randn(64, 32, 1, 1, device=device) * 32 ** -0.5)
    src1, res1, bs1 = torch.randn(Bs, Ss, Ss, 64, device=device), torch.randn(Bs, Ss, Ss, 64, device=device), torch.zeros(64, device=device)
    sec = graph_kernel_seconds(device, lambda: ops.conv(ops.CONV1X1, xs1, ws1, bs1, dmish_src=src1, resid=res1))
    nbytes = 4.0 * Bs * Ss * Ss * (32 + 3 * 64)
    pm = _profile_json(STREAM_PMC, "downsampled-diffusion_amd/csrc/conv1x1_stream.hip")
    out.append(dict(kernel="conv1x1_stream_kernel<32,64> 1x1 conv 32->64 x Mish'(src) + residual @64x64 B=64 (weights in registers, no LDS)",
                    bound="hbm", achieved=nbytes / sec / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s", frac=nbytes / sec / 1e9 / HBM_PEAK_GBPS,
                    traffic=pm["traffic_bytes_per_launch"] if pm else None,
                    traffic_source=f"profiles/{STREAM_PMC} (rocprofv3 --pmc: FETCH_SIZE x2 | WRITE_SIZE)" if pm else None,
                    launch_us=sec * 1e6, algorithmic_mbytes=nbytes / 1e6))
    del xs1, src1, res1
    Bz, S, Cz = 32, 32, 8
    xs = torch.randn(Bz, S, S, Cz, device=device)
    eps = torch.randn(Bz, S, S, Cz, device=device)
    t = torch.full((Bz,), 500, device=device, dtype=torch.long)
    tab = {k: torch.rand(1000, device=device) for k in ("c_recip", "c_recipm1", "c1", "c2", "sigma")}
    sec = graph_kernel_seconds(device, lambda: ops.p_sample_update_(xs, eps, t, noise=None, seed=1, **tab))
    nbytes = 12.0 * xs.numel()
    out.append(dict(kernel="p_sample_kernel fused reverse-step update + Philox noise, 32x32x32x8 latents", bound="hbm",
                    achieved=nbytes / sec / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s", frac=nbytes / sec / 1e9 / HBM_PEAK_GBPS,
                    traffic=None, launch_us=sec * 1e6, algorithmic_mbytes=nbytes / 1e6,
                    note="3 MB per launch: the launch is at the ~2 us kernel-boundary floor, not on the HBM roof"))
    return out


def time_conv_cluster_roofline(device):
    """The variant of the dominant kernel that runs THREE times per step at 32x32 (the plain one above runs once): the same conv
    with GroupNorm + Mish + time shift finished inside the launch (conv3x3_wino2_kernel<P, 0, true>: tile statistics exchanged
    between the workgroups of an image).  Priced the same way: issued MFMA FLOPs / live launch time / fp32 MFMA peak."""
    from ddk import ops
    B, H, W, C, N = 32, 32, 32, 128, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).to(device)
    w = (torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5).to(device)
    b, gam, bet = torch.zeros(N, device=device), torch.ones(N, device=device), torch.zeros(N, device=device)
    temb = torch.randn(B, N, generator=g).to(device)
    wu = ops.pack_conv_weight_wino(w)
    from ddk import lib
    if lib.load().ddk_conv3x3_gn_mish_cluster_ok(B, H, W, C, N, 8) <= 0:
        return None
    try:
        gave_up = lib.load().ddk_debug_cluster_timeouts()          # process-wide count of exchanges that timed out
        fn = lambda: ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet, temb=temb, check=False)   # noqa: E731
        fn()
        sec = graph_kernel_seconds(device, fn)
        torch.cuda.synchronize()
        if lib.load().ddk_debug_cluster_timeouts() != gave_up:
            raise RuntimeError("an in-launch GroupNorm exchange gave up during the timed launches (GPU shared?)")
    except Exception as e:   # noqa: BLE001 -- secondary figure
        log(f"roofline_cluster: {type(e).__name__}: {e}")
        return None
    executed = 2.0 * (B * H * W / 4) * 16 * C * N
    pm = _profile_json(CLUSTER_PMC, WINO_SRC)
    src = f"profiles/{CLUSTER_PMC} (rocprofv3 --pmc, separate passes)" if pm else None
    return dict(kernel="conv3x3_wino2_kernel<2,0,true> conv3x3 128->128 @32x32 B=32 + GroupNorm + Mish + time shift in the same launch "
                       "(3 launches per step)", bound="mfma", achieved=executed / sec / 1e12, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                frac=executed / sec / 1e12 / FP32_PEAK_TFLOPS, traffic=pm["traffic_bytes_per_launch"] if pm else None, traffic_source=src,
                mfma_busy=pm["mfma_busy"] if pm else None, mfma_busy_source=src, launch_us=sec * 1e6, executed_gflop=executed / 1e9)


WINO_STEP_SHAPES = (          # every conv3x3_wino2_kernel launch of one cfg4 reverse step (profiles/r0N_sampler_step_breakdown.txt)
    # (name, H, C_in, C_out, launches per step, GroupNorm finished inside the launch)
    ("128->128 @32x32 + GroupNorm in the launch", 32, 128, 128, 3, True),
    ("128->256 @16x16 + GroupNorm in the launch", 16, 128, 256, 1, True),
    ("256->256 @16x16 + GroupNorm in the launch", 16, 256, 256, 3, True),
    ("512->256 @8x8 (k split, slabs summed by the GroupNorm launch behind it)", 8, 512, 256, 1, False),
    ("512->128 @16x16 (k split over two workgroups, partner tile summed + GroupNorm in the launch)", 16, 512, 128, 1, True),
    ("128->128 @16x16 (k split over two workgroups, partner tile summed + GroupNorm in the launch)", 16, 128, 128, 3, True),
    ("128->128 @32x32", 32, 128, 128, 1, False),
)


def time_wino_variants(device, B=32):
    """`roofline.variants`: every shape the step runs on the Winograd kernel family, each timed live like the headline kernel
    (graph replay, HIP events on the launch stream) and priced on the MFMA FLOPs it ISSUES (16 multiplies per 2x2 output tile and
    channel pair, m tiles padded to 32) against the fp32 MFMA peak; `family_frac` weights them by their launches per step."""
    from ddk import lib, ops
    L = lib.load()
    out, tot_f, tot_t = [], 0.0, 0.0
    g = torch.Generator(device="cpu").manual_seed(7)
    for name, H, C, N, count, gn in WINO_STEP_SHAPES:
        try:
            x = torch.randn(B, H, H, C, generator=g).to(device)
            w = (torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5).to(device)
            b, gam, bet = torch.zeros(N, device=device), torch.ones(N, device=device), torch.zeros(N, device=device)
            temb = torch.randn(B, N, generator=g).to(device)
            wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w)
            in_launch = gn and L.ddk_conv3x3_gn_mish_cluster_ok(B, H, H, C, N, 8) > 0
            if in_launch:
                fn = lambda: ops.conv3x3_gn_mish_cluster(x, wu, b, gam, bet, temb=temb, check=False)   # noqa: E731
            else:
                if L.ddk_conv_wino_splits(B, H, H, C, N) <= 0:
                    continue
                fn = lambda: ops.conv(ops.CONV3X3_S1, x, wp, b, w_wino=wu, leave_slabs=True)           # noqa: E731
            fn()
            sec = graph_kernel_seconds(device, fn, n=40, reps=3)
            tiles = -(-(B * (H // 2) * (H // 2)) // 32) * 32
            executed = 2.0 * tiles * 16 * C * N
            out.append({"shape": name, "launches_per_step": count, "groupnorm_in_launch": bool(in_launch), "launch_us": sec * 1e6,
                        "executed_gflop": executed / 1e9, "achieved": executed / sec / 1e12,
                        "frac": executed / sec / 1e12 / FP32_PEAK_TFLOPS})
            tot_f += count * executed
            tot_t += count * sec
            del x, w, wp, wu
        except Exception as e:   # noqa: BLE001 -- secondary figures
            log(f"roofline.variants: {name}: {type(e).__name__}: {e}")
    fam = tot_f / tot_t / 1e12 / FP32_PEAK_TFLOPS if tot_t > 0 else None
    return out, fam, tot_t * 1e6


def time_b192(model, plan, tables, device, rank, steps=24):
    """Secondary figure at the batch the reference's sampler CLI defaults to (generate_model_samples.py:16: batch_size = 192):
    `steps` timed reverse steps + the x3 decoder at B = 192, extrapolated like the headline value.  Rank-local on purpose: no
    collective inside the guarded section, so a rank that fails here (out of memory, say) cannot leave the others in a barrier."""
    from ddk import ops
    B, C, S = 192, 8, 32
    try:
        with torch.no_grad():
            x = ops.randn((B, S, S, C), device, seed=99, step=T_STEPS, stream_id=rank)
            plan.sample_nhwc(x, tables, T_STEPS - 1, T_STEPS - 20, seed=99, stream_id=rank, use_graph=True)
            img = model.rescaled_upsample(ops.nhwc_to_nchw(x))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.sample_nhwc(x, tables, T_STEPS - 1, T_STEPS - steps, seed=99, stream_id=rank, use_graph=True)
            torch.cuda.synchronize()
            t_step = (time.perf_counter() - t0) / steps
            t1 = time.perf_counter()
            img = model.rescaled_upsample(ops.nhwc_to_nchw(x))
            torch.cuda.synchronize()
            t_dec = time.perf_counter() - t1
            ok = bool(torch.isfinite(img).all()) and tuple(img.shape) == (B, 3, 256, 256)
            del img, x
        torch.cuda.empty_cache()
        assert ok
        return {"batch_per_gpu": B, "ms_per_step": t_step * 1e3, "decode_ms": t_dec * 1e3, "steps": steps,
                "value": B / (T_STEPS * t_step + t_dec)}
    except Exception as e:   # noqa: BLE001 -- secondary figure: report the failure, keep the headline line
        log(f"B=192 secondary figure failed: {type(e).__name__}: {e}")
        return None


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (a 1-GPU box
    shows all 256 hardware threads in os.cpu_count() but grants a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("DDK_CPU_THREADS", "32"))))


def cpu_baseline(cfg, state_dict, batch):
    """The CPU oracle (torch-CPU restatement, pinned to the reference by tests/golden) timed on the host cores on the SAME
    work the GPU line measures: the cfg4 UNet step at the full batch (1 warm-up + 3 timed steps, extrapolated to T=1000)
    plus one pass of the x3 ConvResNet decoder + tanh at the full batch."""
    from oracle import resampler_ref as R
    from oracle import unet_ref as U
    from utils import synthetic as syn
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd_all = {k: v.detach().cpu() for k, v in state_dict.items()}
    sd = {k[len("latent_model."):]: v for k, v in sd_all.items() if k.startswith("latent_model.")}
    x = syn.synthetic_normal((batch, 8, 32, 32), "bench.cpu.x")
    t = torch.full((batch,), 500, dtype=torch.long)
    with torch.no_grad():
        U.unet_forward(sd, cfg, x, t)
        t0 = time.perf_counter()
        reps = 3          # 1 warm-up + >= 3 timed UNet steps
        for _ in range(reps):
            U.unet_forward(sd, cfg, x, t)
        dt = (time.perf_counter() - t0) / reps
        t1 = time.perf_counter()
        img = R.rescaled_upsample(sd_all, cfg, torch.tanh(x))
        t_dec = time.perf_counter() - t1
    assert img.shape == (batch, 3, 256, 256)
    return dict(value=batch / (T_STEPS * dt + t_dec), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"1 warm-up + {reps} timed UNet steps (8x32x32 latents, batch {batch}, {dt * 1e3:.0f} ms/step) extrapolated to "
                       f"T={T_STEPS}, + one decoder pass at batch {batch} ({t_dec:.1f} s)")


def train_step_ms(device, steps=5):
    """Secondary figure: one optimiser step of cfg3 (CelebA 64x64 dDDPM -downsample 2, batch 64:
    2 accumulation micro-batches, clip, Adam) with the accumulation passes replayed as one device graph.  Timed in the form the
    product trainer runs by default -- the step's two micro-batches as ONE pass over their 128 samples (trainers/trainer_ddpm.py,
    merge_micro_batches: the objective is a mean of per-sample terms, so the gradient is the same) -- and, next to it, as the
    reference's two passes."""
    from models import DownsampleDDPMAutoencoder, Unet
    from trainers.graph_step import GraphedAccumulation
    from trainers.optim import FusedAdam
    from utils import synthetic as syn
    c = dict(unet_chan=128, unet_in=8, unet_dims=(1, 2, 2, 2), unet_dropout=0.1, image_size=64, T=1000, loss_type="simple",
             beta_schedule="linear", loss_flat="sum", ema_decay=0.995, d_mode="convolutional_res", u_mode="convolutional_res",
             d_dropout=0, d_chans=64, d_n_blocks=3, u_n_blocks=3, ae_loss=True, t_rec_max=100, force_latent=True, n_downsamples=2)
    model = DownsampleDDPMAutoencoder(c, Unet(c), "cuda", 3).to(device).train()
    model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    opt = FusedAdam(model, lr=2e-4)
    xs = [torch.rand((64, 3, 64, 64), device=device) * 2 - 1 for _ in range(2)]

    def timed(batches):
        ga = GraphedAccumulation(model, len(batches)).capture(batches)
        opt.zero_grad()

        def step():
            ga.replay(batches)
            opt.step()
            opt.zero_grad()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    ms_two = timed(xs)
    ms = timed([torch.cat(xs)])

    def gflops(batch, passes):
        """(algorithmic, issued) GFLOP of `passes` forward + backward passes over `batch` samples each.  Algorithmic: 3 x 2 MAC of
        every conv / projection / attention product (forward, input gradient, weight gradient).  Issued: the forward as the plan
        dispatches it (ddk_unet_flops_executed: Winograd 3x3 convs issue 16/36 of their direct multiplies); the input gradient
        priced like the forward (the gradient of a stride-1 3x3 conv is again a stride-1 3x3 conv, of dY, on the same kernels);
        the weight gradient direct.  The encoder / decoder blocks run direct kernels throughout."""
        unet, dn, up = model.latent_model, model.downsample, model.upsample
        edge = dn.flops(batch, 64, 64) + up.flops(batch, 16, 16)
        alg = passes * 3 * (edge + unet.flops(batch, 16, 16))
        issued = passes * (3 * edge + 2 * unet.flops_executed(batch, 16, 16) + unet.flops(batch, 16, 16))
        return alg / 1e9, issued / 1e9
    alg1, iss1 = gflops(128, 1)
    alg2, iss2 = gflops(64, 2)
    return ms, {"kernel": "one optimiser step of cfg3 (CelebA 64x64 dDDPM -downsample 2, batch 64 = 2 micro-batches): encoder + UNet + "
                          "decoder forward and backward, clip, Adam, EMA -- schedule `merged`: the two micro-batches as ONE pass over "
                          "their 128 samples (the product trainer's default where memory allows; same gradient, different RNG draw order "
                          "and 2x activation memory vs the reference's trainer_ddpm.py:118-128); `two_passes` below is the reference's "
                          "own pass-by-pass sequence",
                "schedule": "merged", "bound": "mfma", "ms_per_step": ms, "executed_gflop": iss1, "achieved": iss1 / ms, "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": iss1 / ms / FP32_PEAK_TFLOPS, "algorithmic_gflop": alg1, "algorithmic_equiv_tflops": alg1 / ms,
                "algorithmic_frac": alg1 / ms / FP32_PEAK_TFLOPS,
                "two_passes": {"schedule": "two_passes", "ms_per_step": ms_two, "executed_gflop": iss2, "achieved": iss2 / ms_two,
                               "frac": iss2 / ms_two / FP32_PEAK_TFLOPS, "algorithmic_gflop": alg2,
                               "algorithmic_frac": alg2 / ms_two / FP32_PEAK_TFLOPS},
                "ms_per_step_two_passes": ms_two,
                "note": "frac prices the MFMA / FMA FLOPs ISSUED (forward as dispatched by the plan, input gradient priced like the forward, "
                        "weight gradient direct) over the measured step time (wall clock incl. the optimiser) against the fp32 peak; "
                        "algorithmic_frac prices 2 micro-batches x 3 passes x 2 MAC of the direct algorithm on the same time (a statement "
                        "about the algorithm, not the chip)"}


def visible_gpus():
    """GPUs this process may use, counted WITHOUT the HIP runtime (torch.cuda.device_count() falls back to hipGetDeviceCount when
    amdsmi is missing, and that initialises HSA in the caller): GPU nodes of the KFD topology (`simd_count` > 0 in
    /sys/class/kfd/kfd/topology/nodes/*/properties), narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    when one of them is set.  None when the topology cannot be read (the children then find out for themselves)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.exists(root) and not os.path.exists("/dev/kfd"):
        return 0                     # no amdgpu compute driver in this system at all
    try:
        n = 0
        for node in sorted(os.listdir(root)):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


EADDRINUSE_EXIT = 98      # a rank that could not bind / reach the rendezvous port leaves with this code; the parent picks another port


def _make_child_preexec():
    """The function a child runs between fork and exec, i.e. before the child's program (and any GPU call) starts: the kernel sends
    the child SIGKILL the moment its parent dies, however the parent dies (SIGKILL included: no handler of the parent's is
    involved).  libc is resolved HERE, in the parent, so that nothing is imported between fork and exec."""
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)
    parent = os.getpid()
    PR_SET_PDEATHSIG, SIGKILL = 1, 9

    def preexec():
        libc.prctl(PR_SET_PDEATHSIG, SIGKILL, 0, 0, 0)
        if os.getppid() != parent:   # the parent died before the prctl took effect
            os._exit(1)
    return preexec


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher in front (WORLD_SIZE unset): start the N rank processes here.

    This parent NEVER touches the GPU: it does not import torch, it counts devices from the KFD topology in sysfs (visible_gpus),
    picks a free rendezvous port, starts N fresh children of this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (one
    process per GPU, child r bound to cuda:r), relays rank 0's JSON line to stdout (everything else of the children goes to stderr)
    and exits non-zero when any child fails or the job exceeds DDK_BENCH_TIMEOUT seconds (default 480: under the driver's 600 s
    limit, so it is this parent that stops a stalled job, not a kill of the parent).  Nothing is ever re-exec'd.

    No child outlives the parent: every child asks the kernel for SIGKILL on the parent's death (PR_SET_PDEATHSIG, set before
    its program starts), and SIGTERM / SIGINT / SIGHUP to the parent kill the children's process groups before the parent
    leaves with 128 + signal."""
    import signal
    import socket
    import subprocess
    import threading
    n = args.gpus
    same = bool(os.environ.get("DDK_BENCH_SAME_DEVICE"))
    have = visible_gpus()
    if have is not None and have < n and not same:
        log(f"bench.py: --gpus {n} but only {have} GPU(s) visible (one process per GPU over RCCL); "
            "DDK_BENCH_SAME_DEVICE=1 rehearses N ranks on cuda:0 over gloo")
        return 2
    timeout = float(os.environ.get("DDK_BENCH_TIMEOUT", "480"))
    procs, lines = [], []
    preexec = _make_child_preexec()

    class Stopped(Exception):
        pass

    def on_signal(signum, _frame):
        raise Stopped(signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, on_signal)

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)       # the session this function started for exactly that child
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:   # noqa: BLE001
                pass

    def start(port):
        base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL between processes)
        del procs[:]
        del lines[:]
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, text=True,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True,
                                          preexec_fn=preexec))
        out0 = procs[0].stdout

        def relay():
            for ln in out0:
                lines.append(ln)
        th = threading.Thread(target=relay, daemon=True)
        th.start()
        return th

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    deadline = time.monotonic() + timeout
    rc, th = 0, None
    try:
        for attempt in range(3):
            th = start(free_port())
            rc = 0
            while True:
                codes = [p.poll() for p in procs]
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                if bad:
                    rc = EADDRINUSE_EXIT if any(c == EADDRINUSE_EXIT for _, c in bad) else 1
                    log(f"bench.py: rank {bad[0][0]} exited with code {bad[0][1]}; stopping the other ranks")
                    break
                if all(c == 0 for c in codes):
                    break
                if time.monotonic() > deadline:
                    log(f"bench.py: the {n}-rank job exceeded DDK_BENCH_TIMEOUT = {timeout:.0f} s; stopping it")
                    rc = 3
                    break
                time.sleep(0.2)
            if rc != EADDRINUSE_EXIT:
                break
            kill_all()                      # the port chosen by bind-then-close was taken in between: once more on another one
            log("bench.py: the rendezvous port was taken by another process; starting the ranks again on a new port")
            rc = 1
    except Stopped as e:
        sg = int(e.args[0])
        log(f"bench.py: signal {sg} received; stopping the {n} ranks")
        rc = 128 + sg
    finally:
        kill_all()
    if th is not None:
        th.join(timeout=10)
    out = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in out:
            sys.stderr.write(ln)
    if rc == 0 and not out:
        log("bench.py: rank 0 printed no JSON line")
        rc = 4
    if rc == 0:
        sys.stdout.write(out[-1])
        sys.stdout.flush()
    return rc


def rank_report(device, world, payload):
    """What the line says about the job's ranks, learnt over the process group itself: a REAL all-reduce of (rank + 1) must sum
    to world (world + 1) / 2 -- `rccl_world` is the group size that all-reduce ran over -- and every rank's own figures
    (`payload`: ms per step, the in-launch GroupNorm option it ended with, its device) are gathered, not assumed from rank 0."""
    import torch.distributed as dist
    tok = torch.tensor([float(dist.get_rank() + 1)], device=device, dtype=torch.float64)
    dist.all_reduce(tok, op=dist.ReduceOp.SUM)
    w = dist.get_world_size()
    if abs(float(tok[0]) - w * (w + 1) / 2) > 1e-9:
        raise RuntimeError(f"all-reduce over {w} ranks summed to {float(tok[0])}, expected {w * (w + 1) / 2}")
    box = [None] * w
    dist.all_gather_object(box, payload)
    return {"rccl_world": w if dist.get_backend() == "nccl" else None, "world": w, "backend": dist.get_backend(), "ranks": box}


def cfg5_train():
    """BASELINE.json config 5 as train.py builds it (reference train.py:19-46 + -d celeba_hq -bs 8 -is 256): the full-resolution DDPM"""
    return dict(model="ddpm", dataset="celeba_hq", batch_size=8, image_size=256, n_steps=10 ** 9, lr=2e-4, unet_chan=128,
                unet_dims=(1, 2, 2, 2), unet_dropout=0.1, T=T_STEPS, loss_type="simple", beta_schedule="linear", ema_decay=0.995,
                loss_flat="sum", val_split=0, n_downsamples=0, n_samples=4, rnd_flip=False)


def cfg3_train():
    return dict(model="dddpm", dataset="celeba", batch_size=64, image_size=64, n_steps=10 ** 9, lr=2e-4, unet_chan=128,
                unet_dims=(1, 2, 2, 2), unet_dropout=0.1, T=T_STEPS, loss_type="simple", beta_schedule="linear", ema_decay=0.995,
                loss_flat="sum", val_split=0, n_downsamples=2, n_samples=4, rnd_flip=False, d_mode="convolutional_res",
                u_mode="convolutional_res", d_dropout=0, d_chans=64, d_n_blocks=3, u_n_blocks=3, unet_in=8, ae_loss=True,
                t_rec_max=100, force_latent=True)


def train_dp(args, rank, world, device, dist_on, fence):
    """--train-dp: the data-parallel training line (BASELINE.json config 5: 256x256 DDPM, -bs 8 per GPU; reference loop
    trainers/trainer_ddpm.py:113-158).  Every rank is ONE product trainer (trainers.setup_trainer: model build, C1 weight broadcast,
    FusedAdam on the flat buckets); a step = TrainerDDPM._accumulate (2 micro-batches of 8, captured device graph) ->
    optimizer_step (C2: ONE all-reduce of the flat fp32 gradient bucket, then clip + Adam) -> update_ema.  Inputs are synthetic
    batches resident in HBM.  Timed three ways over K steps each: the whole step, the step with the all-reduce skipped, and the
    all-reduce of the bucket alone."""
    import torch.distributed as dist
    from parallel import all_reduce_flat_
    from trainers import setup_trainer
    from trainers.trainer_ddpm import _invalidate
    cfg = {"cfg5": cfg5_train, "cfg3": cfg3_train}[args.train_config]()
    if args.train_batch:
        cfg["batch_size"] = args.train_batch
    if args.train_image_size:
        cfg["image_size"] = args.train_image_size
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the trainer's own prints are diagnostics here: stdout carries ONE JSON line
        trainer, cfg = setup_trainer(cfg, True, None, "bench_dp", seed=0)
    B, S = cfg["batch_size"], cfg["image_size"]
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    pool = [((torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(device), 0) for _ in range(4)]

    def resident():
        while True:
            yield from pool
    trainer.train_loader = resident()
    opt = trainer.opt

    def step(allreduce=True):
        trainer.model.train()
        trainer._accumulate()
        if allreduce:
            trainer.optimizer_step()
        else:
            opt.step()
            opt.zero_grad()
            _invalidate(trainer.model)
        if trainer.use_ema:
            trainer.update_ema()
        trainer.step += 1

    def timed(fn, k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        fence()
        dt = (time.perf_counter() - t0) / k
        if dist_on:
            tm = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            return dt, float(tm[0])
        return dt, dt

    for _ in range(max(args.warmup, 2)):          # the first call captures the accumulation graph
        step()
    own_ms, t_full = timed(step, args.steps)
    _, t_noar = timed(lambda: step(False), args.steps)
    bucket = opt.fp.grad
    _, t_ar = timed(lambda: all_reduce_flat_(bucket, average=True, force=dist_on), args.steps) if dist_on else (0.0, 0.0)
    opt.zero_grad()
    model = trainer.model
    if cfg["model"] == "ddpm":
        fwd = model.latent_model.flops(B, S, S)
    else:
        lat = S >> cfg["n_downsamples"]
        fwd = model.downsample.flops(B, S, S) + model.latent_model.flops(B, lat, lat) + model.upsample.flops(B, lat, lat)
    acc = trainer.gradient_accumulate_every
    merged = bool(getattr(trainer, "merge_micro_batches", False)) and acc > 1
    gflop = acc * 3 * fwd / 1e9
    if cfg["model"] == "ddpm":
        fx = model.latent_model.flops_executed(B * acc, S, S) if merged else acc * model.latent_model.flops_executed(B, S, S)
        issued = (2 * fx + acc * fwd) / 1e9
    else:
        edge = model.downsample.flops(B, S, S) + model.upsample.flops(B, lat, lat)
        fx = model.latent_model.flops_executed(B * acc, lat, lat) if merged else acc * model.latent_model.flops_executed(B, lat, lat)
        issued = (3 * acc * edge + 2 * fx + acc * model.latent_model.flops(B, lat, lat)) / 1e9
    info = rank_report(device, world, {"rank": rank, "device": str(device), "ms_per_step": own_ms * 1e3}) if dist_on else None
    if rank != 0:
        return
    nbytes = bucket.numel() * 4
    out = {"metric": f"training images/sec, {args.train_config} data-parallel optimiser step (gradient all-reduce included)",
           "value": acc * B * world / t_full, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2),
           "ms_per_step": t_full * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": (f"{args.train_config}: {cfg['dataset']} {S}x{S} {cfg['model']} -downsample {cfg['n_downsamples']}, "
                                   f"-bs {B} per GPU, one optimiser step = {acc} micro-batches of {B} (captured device graph) + ONE "
                                   "all-reduce of the flat fp32 gradient bucket + clip + Adam + EMA, through trainers.setup_trainer / "
                                   "TrainerDDPM.optimizer_step"),
                      "batch_per_gpu": B, "global_batch": B * world, "accumulate": acc,
                      "ms_per_step_no_allreduce": t_noar * 1e3, "allreduce_alone_ms": t_ar * 1e3 if dist_on else None,
                      "grad_bucket_bytes": nbytes,
                      "allreduce_bus_gbps": (2 * (world - 1) / world * nbytes / t_ar / 1e9) if dist_on and world > 1 and t_ar > 0 else None,
                      "n_params": int(bucket.numel()), "graph_train": bool(trainer._graph)},
           "roofline_train": {"kernel": "one optimiser step: forward + input-gradient + weight-gradient passes of every conv / projection "
                                        "/ attention product, clip, Adam, EMA -- schedule `" + ("merged" if merged else "two_passes") + "`",
                              "schedule": "merged" if merged else "two_passes", "bound": "mfma", "executed_gflop": issued,
                              "achieved": issued / (t_full * 1e3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": issued / (t_full * 1e3) / FP32_PEAK_TFLOPS, "algorithmic_gflop": gflop,
                              "algorithmic_frac": gflop / (t_full * 1e3) / FP32_PEAK_TFLOPS, "per": "GPU",
                              "note": "frac: FLOPs issued (forward as the plan dispatches it, input gradient priced like the forward, "
                                      "weight gradient direct) / step time / fp32 peak; algorithmic_frac: the direct algorithm's FLOPs "
                                      "on the same time"}}
    if info:
        out["rccl_world"], out["dist"] = info["rccl_world"], info
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200 reverse steps; 5 optimiser steps with --train-dp)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=32, help="latents per GPU (cfg4: 32)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary cfg3 training-step timing")
    ap.add_argument("--no-full-chain", action="store_true", help="skip the one full T=1000 chain + decode behind the timed steps")
    ap.add_argument("--no-b192", action="store_true", help="skip the secondary figure at the reference CLI's default batch of 192")
    ap.add_argument("--train-dp", action="store_true", help="data-parallel TRAINING line instead of the sampling line")
    ap.add_argument("--train-config", choices=("cfg5", "cfg3"), default="cfg5")
    ap.add_argument("--train-batch", type=int, default=None, help="override -bs of --train-config (tests)")
    ap.add_argument("--train-image-size", type=int, default=None, help="override -is of --train-config (tests)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 5 if args.train_dp else 200
    if args.warmup is None:
        args.warmup = 2 if args.train_dp else 10

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))     # the parent: torch is not even imported in this process

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    probe = os.environ.get("DDK_BENCH_LAUNCH_PROBE")
    if probe:
        # launcher self-test (tests/test_host_logic.py, no GPU, torch never imported): a rank reports the environment it was started
        # with and leaves; "fail<r>" makes rank r exit 7 while the others would run for a minute, "hang" makes every rank outlive the
        # timeout, "pids" makes every rank leave its pid in DDK_BENCH_PROBE_DIR and stay for a minute (what survives the parent?),
        # "eaddr" makes rank 0 report a taken rendezvous port on the job's first start only
        pdir = os.environ.get("DDK_BENCH_PROBE_DIR", "")
        if probe == f"fail{rank}":
            sys.exit(7)
        if probe == "pids":
            with open(os.path.join(pdir, f"rank{rank}.pid"), "w") as f:
                f.write(str(os.getpid()))
            time.sleep(60)
        if probe == "eaddr":
            mark = os.path.join(pdir, "second_start")
            if rank == 0 and not os.path.exists(mark):
                open(mark, "w").close()
                sys.exit(EADDRINUSE_EXIT)
            if rank != 0 and not os.path.exists(mark):
                time.sleep(60)
        if probe.startswith("fail") or probe == "hang":
            time.sleep(60)
        if rank == 0:
            print(json.dumps({"probe": True, "n_gpus": world, "rank": rank, "local_rank": local, "argv": sys.argv[1:],
                              "master": [os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")]}), flush=True)
        else:
            print(f"probe rank {rank} of {world}", flush=True)      # relayed to stderr by the parent
        return
    _import_torch()
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    same_device = bool(os.environ.get("DDK_BENCH_SAME_DEVICE"))
    if same_device:       # rehearsal of the N>1 path on a 1-GPU box (gloo, all ranks on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # DDK_BENCH_FORCE_DIST=1: take the N > 1 branches (process group, C1 weight broadcast, barriers, max-over-ranks) on a world
    # of ONE rank too -- tests/test_rccl_gpu.py walks the RCCL path of this file on the 1-GPU box before an 8-GPU node sees it
    dist_on = world > 1 or bool(os.environ.get("DDK_BENCH_FORCE_DIST"))
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL wants one device per rank: several ranks on cuda:0 can only rendezvous over gloo
        backend = os.environ.get("DDK_BENCH_BACKEND", "gloo" if (same_device and world > 1) else "nccl")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:   # noqa: BLE001
            if "EADDRINUSE" in str(e) or "address already in use" in str(e).lower():
                log(f"rank {rank}: rendezvous port {os.environ.get('MASTER_PORT')} is taken: {e}")
                sys.exit(EADDRINUSE_EXIT)       # launch_ranks starts the job again on another port
            raise

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.train_dp:
        from ddk import lib
        assert lib.load().ddk_device_ok() == 1, lib.last_error()
        with_env = dict(LOCAL_RANK=str(local))
        os.environ.update(with_env)              # setup_trainer binds cuda:LOCAL_RANK (0 in the same-device rehearsal)
        train_dp(args, rank, world, device, dist_on, fence)
        if dist_on:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    from ddk import lib, ops
    assert lib.load().ddk_device_ok() == 1, lib.last_error()
    from models import DownsampleDDPM, Unet
    from utils import synthetic as syn

    cfg = cfg4()
    model = DownsampleDDPM(cfg, Unet(cfg), "cuda", 3)
    if rank == 0:
        model.load_state_dict(syn.fill_state_dict(model.state_dict(), skip=syn.SCHEDULE_KEYS))
    model = model.to(device).eval()
    from parallel import dist as pdist
    bcast_bytes = pdist.broadcast_module_(model, src=0, force=True) if dist_on else 0   # C1: one flat fp32 bucket over RCCL
    model.rng_stream_id = rank
    model.use_graph = not args.no_graph

    B, C, S = args.batch, 8, 32
    unet = model.latent_model
    plan = unet.plan()
    if same_device and world > 1:
        # ranks that share one GPU cannot host a whole cluster of the in-launch GroupNorm (generate_model_samples.py does the same)
        plan.set_option(plan.OPT_CLUSTER_GROUPNORM, 0)
    tables = model._tables()
    x = ops.randn((B, S, S, C), device, seed=1234, step=T_STEPS, stream_id=rank)

    def run_steps(k):
        done = 0
        while done < k:
            n = min(T_STEPS, k - done)
            plan.sample_nhwc(x, tables, T_STEPS - 1, T_STEPS - n, seed=1234, stream_id=rank, use_graph=model.use_graph)
            done += n

    def decode():
        return model.rescaled_upsample(ops.nhwc_to_nchw(x))

    def settle_clock(min_s=0.5, max_s=2.0, seg=96, tol=0.01):
        """Warm up by TIME, whatever --warmup says: the shader clock this chip holds under the step's load is reached only after a
        few hundred ms (a 5-step warm-up left the 20 timed steps of the round-5 record at 2.336 GHz while its full chain ran at
        2.393).  Runs segments of `seg` reverse steps, each bracketed by a clock probe, until at least `min_s` seconds have passed
        AND two consecutive segments held the same clock within `tol` (or `max_s` is over).  Untimed; every rank does the same."""
        t_start, prev, ghz, segs = time.perf_counter(), None, None, 0
        while True:
            cp = ops.ClockProbe(device)
            cp.probe()
            run_steps(seg)
            cp.probe()
            torch.cuda.synchronize()
            ghz = cp.ghz()
            segs += 1
            dt = time.perf_counter() - t_start
            if dt >= min_s and prev and ghz and abs(ghz / prev - 1) <= tol:
                return {"seconds": dt, "segments": segs, "steps": segs * seg, "ghz_last": ghz, "ghz_before": prev, "settled": True}
            if dt >= max_s:
                return {"seconds": dt, "segments": segs, "steps": segs * seg, "ghz_last": ghz, "ghz_before": prev, "settled": False}
            prev = ghz

    clk_steps, clk_chain = ops.ClockProbe(device), ops.ClockProbe(device)
    with torch.no_grad():
        run_steps(max(args.warmup, 1))
        decode()
        settled = settle_clock()
        fence()
        clk_steps.probe()
        t0 = time.perf_counter()
        run_steps(args.steps)
        clk_steps.probe()
        fence()
        elapsed = time.perf_counter() - t0
        dec = []
        for _ in range(3):
            fence()
            t1 = time.perf_counter()
            img = decode()
            fence()
            dec.append(time.perf_counter() - t1)
        t_decode = min(dec)
        assert torch.isfinite(img).all() and img.shape == (B, 3, 256, 256)
        # ONE real job unit, not an extrapolation: a full T = 1000 chain from fresh x_T plus its decode (what
        # generate_model_samples.py times per batch, reference generate_model_samples.py:42-58), under a sustained clock
        full_chain_s = None
        if not args.no_full_chain:
            x.copy_(ops.randn((B, S, S, C), device, seed=4321, step=T_STEPS, stream_id=rank))
            fence()
            clk_chain.probe()
            t2 = time.perf_counter()
            run_steps(T_STEPS)
            img = decode()
            clk_chain.probe()
            fence()
            full_chain_s = time.perf_counter() - t2
            assert torch.isfinite(img).all()

    b192 = None
    if not args.no_b192 and not same_device:
        b192 = time_b192(model, plan, tables, device, rank)

    dist_info = None
    if dist_on:
        own = {"rank": rank, "device": str(device), "ms_per_step": elapsed / args.steps * 1e3, "decode_ms": t_decode * 1e3,
               "full_chain_s": full_chain_s, "in_launch_groupnorm": plan._cluster, "cluster_timeouts": plan.cluster_timeouts()}
        tmax = torch.tensor([elapsed, t_decode, full_chain_s or 0.0], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed, t_decode = float(tmax[0]), float(tmax[1])
        full_chain_s = float(tmax[2]) if full_chain_s is not None else None
        dist_info = rank_report(device, world, own)

    t_step = elapsed / args.steps
    images_per_sec = B * world / (T_STEPS * t_step + t_decode)
    flops_step = unet.flops(B, S, S)
    flops_exec = unet.flops_executed(B, S, S)

    if rank == 0:
        roof = time_conv_roofline(device)
        roof["variants"], roof["family_frac"], roof["family_us_per_step"] = time_wino_variants(device, B)
        roof_hbm = time_hbm_rooflines(device)
        out = {
            "metric": "images/sec (T=1000 sampling), 256x256 dDDPM-x3",
            "value": images_per_sec, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg4: CelebAMask-HQ 256x256 dDDPM -downsample 3, 32 latents (8x32x32) per GPU, T=1000, "
                                   "UNet chan 128 dims (1,2,2,2) + x3 ConvResNet decoder; batch-sharded, no collective in the loop",
                       "batch_per_gpu": B, "global_batch": B * world, "T": T_STEPS, "unet_step_ms": t_step * 1e3,
                       "decode_ms": t_decode * 1e3, "hip_graph": model.use_graph,
                       "weights_broadcast_bytes": bcast_bytes,
                       "shader_clock_ghz_timed_steps": clk_steps.ghz(),
                       "clock_warmup": settled,
                       "in_launch_groupnorm": plan._cluster},
            "roofline": roof,
            "roofline_cluster": time_conv_cluster_roofline(device),
            "roofline_step": {"bound": "mfma", "algorithmic_gflop": flops_step / 1e9, "executed_gflop": flops_exec / 1e9,
                              "ms_per_step": t_step * 1e3, "achieved": flops_exec / t_step / 1e12, "peak": FP32_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops_exec / t_step / 1e12 / FP32_PEAK_TFLOPS,
                              "algorithmic_equiv_tflops": flops_step / t_step / 1e12,
                              "note": "the whole reverse step: FLOPs issued by every kernel of the plan (Winograd convs at 16/36 of "
                                      "their direct multiplies, ddk_unet_flops_executed) over the measured step time"},
            "roofline_hbm": roof_hbm,
            "roofline_local": time_local_roofline(device),
        }
        try:
            out["roofline_chain"] = time_chain_roofline(device)
        except Exception as e:   # noqa: BLE001 -- secondary figure
            out["roofline_chain"] = None
            log(f"roofline_chain: {type(e).__name__}: {e}")
        if full_chain_s is not None:
            vfc = B * world / full_chain_s
            out["value_full_chain"] = vfc
            out["config"]["full_chain_s"] = full_chain_s
            out["config"]["shader_clock_ghz_full_chain"] = clk_chain.ghz()
            out["config"]["full_chain_vs_extrapolated"] = vfc / images_per_sec
            out["config"]["full_chain_agrees_within_3pct"] = bool(abs(vfc / images_per_sec - 1) <= 0.03)
            if not out["config"]["full_chain_agrees_within_3pct"]:
                log(f"WARNING: the full T={T_STEPS} chain ({vfc:.2f} images/s) and the value extrapolated from {args.steps} timed steps "
                    f"({images_per_sec:.2f}) differ by more than 3 %")
        if dist_info:
            ms = [r["ms_per_step"] for r in dist_info["ranks"]]
            out["rccl_world"] = dist_info["rccl_world"]
            out["dist"] = dist_info
            out["config"]["ms_per_step_min_rank"], out["config"]["ms_per_step_max_rank"] = min(ms), max(ms)
            out["config"]["in_launch_groupnorm_per_rank"] = [r["in_launch_groupnorm"] for r in dist_info["ranks"]]
        if b192:
            out["value_b192"] = b192["value"] * world
            out["config"]["b192"] = b192
        if world == 1 and not args.no_train:
            try:
                out["config"]["train_step_ms_cfg3_bs64"], out["roofline_train"] = train_step_ms(device)
            except Exception as e:   # noqa: BLE001 -- secondary figure: report the failure, keep the headline line
                out["config"]["train_step_ms_cfg3_bs64"] = None
                log(f"training-step timing failed: {type(e).__name__}: {e}")
        if world == 1 and not args.no_cpu_baseline:          # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, model.state_dict(), B)
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
