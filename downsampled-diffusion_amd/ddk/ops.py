"""NHWC functional wrappers over the C ABI: one Python function per HIP kernel family.

Tensors are fp32, contiguous, on a ROCm device; activations are [B, H, W, C].  torch is used
only to allocate outputs / workspaces and to supply the current stream.
"""
import ctypes as C

import torch

from . import lib as L

CONV3X3_S1, CONV3X3_S2, CONV1X1, CONVT4X4_S2 = 0, 1, 2, 3
GN_GROUPS, GN_EPS, LN_EPS = 8, 1e-5, 1e-5


def pad32(c):
    return (c + 31) // 32 * 32


# ---------------------------------------------------------------- kernel-layout copies of weights (training path)
# The autograd Functions need a canonical weight in kernel layout for every conv forward / input-gradient.  Within one optimiser
# step the weights do not change between the accumulation micro-batches, so the copies are cached per live parameter and marked
# with the state they are current for: the parameter's torch version counter + an epoch that every in-place weight update done by
# OUR kernels bumps (FusedAdam.step, EMA, load_state_dict paths call weights_changed(); torch ops bump the version counter).
# A stale copy is not thrown away: its buffer is refreshed in place, and the first stale hit after a weight update refreshes EVERY
# registered copy with one launch (ddk_pack_jobs: 285 single-tensor pack launches per cfg3 optimiser step before).
class _PackEntry:
    __slots__ = ("wref", "out", "job", "fresh", "pooled")


_pack_cache = {}              # (tag, data_ptr, shape) -> _PackEntry
_weights_epoch = [0]
_pack_table = {}              # "keys": tuple of cache keys in table order, "dev": device table (uint8), "n", "blocks"
pack_stats = {"single": 0, "batched_launches": 0, "batched_jobs": 0}
_graph_tables = []            # what captured launches point at when no owner is registered (kept for the process' life)
_graph_keep = [_graph_tables]  # innermost registered owner list last; graph_owner() pushes the capturing object's own list


class graph_owner:
    """`with ops.graph_owner(keep):` around a stream capture: every buffer a captured launch of this module addresses through a
    device-side table -- the job table of ddk_pack_jobs AND the weights / kernel-layout copies its jobs read and write, multi_add
    tables -- is appended to `keep`.  The object that owns the captured graph keeps that list for as long as it may replay, so a
    replay never writes through memory another model (or a replaced cache entry) has given back; when the graph goes, so do they."""

    def __init__(self, keep):
        self.keep = keep

    def __enter__(self):
        _graph_keep.append(self.keep)
        return self.keep

    def __exit__(self, *exc):
        for i in range(len(_graph_keep) - 1, 0, -1):       # by identity (two empty lists compare equal); entry 0 is the process-wide list
            if _graph_keep[i] is self.keep:
                del _graph_keep[i]
                break
        return False


def weights_changed():
    """Every cached kernel-layout copy is stale from here on.  Copies that were allocated while a graph was being captured live in
    that graph's private pool (each replay rewrites them): they are dropped, so no eager call is ever served one."""
    _weights_epoch[0] += 1
    dead = [k for k, e in _pack_cache.items() if e.pooled or e.wref() is None]
    for k in dead:
        del _pack_cache[k]
    if dead:
        _pack_table.clear()


def _pack_job(tag, w):
    """(kind, parameters) of ddk_pack_job for the copy `tag` of weight w, or None when the copy has no batched form"""
    name = tag if isinstance(tag, str) else tag[0]
    if w.dim() != 4:
        return None
    o, i, kh, kw = w.shape
    if name == "fwd":
        return (0, [o, i, kh * kw, pad32(i), i, pad32(i)])
    if name == "fwdT" and (kh, kw) == (4, 4):                 # (I, O, 4, 4)
        return (1, [o, i, o, i])
    if name == "dgrad":
        return (2, [o, i, kh * kw, tag[1], o])
    if name == "wino" and (kh, kw) == (3, 3):
        return (3, [o, i, pad32(i)])
    if name == "wino_dgrad" and (kh, kw) == (3, 3):
        return (4, [o, i, tag[1], tag[2], pad32(o)])
    return None


def _state(w):
    return (w._version, _weights_epoch[0])


def _repack_stale():
    """Refresh every stale registered copy in place with ONE launch.  Returns False when that is not possible right now (a new
    job set while a graph is being captured: building its device table would be a host-to-device copy inside the capture)."""
    keys, live = [], []
    for k, e in _pack_cache.items():
        w = e.wref()
        if w is None or e.job is None or e.fresh == _state(w):
            continue
        keys.append((k, w.data_ptr(), e.out.data_ptr()))     # the cached device table is valid for exactly these buffers
        live.append((e, w))
    if not keys:
        return True
    keys = tuple(keys)
    tab = _pack_table
    if tab.get("keys") != keys:
        if torch.cuda.is_current_stream_capturing():
            return False
        jobs = (L.PackJob * len(keys))()
        for j, (e, w) in zip(jobs, live):
            j.src, j.dst, j.kind = w.data_ptr(), e.out.data_ptr(), e.job[0]
            for q, v in enumerate(e.job[1]):
                j.p[q] = int(v)
        blocks = L.load().ddk_pack_jobs_layout(jobs, len(keys))
        if blocks < 0:
            L.check(-1, "pack_jobs_layout")
        host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        tab.clear()
        tab.update(keys=keys, dev=host.to(live[0][1].device), n=len(keys), blocks=int(blocks))
    if torch.cuda.is_current_stream_capturing():
        # a captured launch reads this table at every replay and writes through the addresses in it: the graph's owner keeps the
        # table, every source weight and every destination copy alive (the cache may drop or replace its entries meanwhile)
        _graph_keep[-1].append((tab["dev"], [(w, e.out) for e, w in live]))
    L.check(L.load().ddk_pack_jobs(tab["dev"].data_ptr(), tab["n"], tab["blocks"], L.stream()), "pack_jobs")
    pack_stats["batched_launches"] += 1
    pack_stats["batched_jobs"] += len(keys)
    for e, w in live:
        e.fresh = _state(w)
    return True


def cached_pack(tag, w, fn):
    """The kernel-layout copy `tag` of the live parameter `w`, current for its present contents (a weakref guards against a new
    tensor re-using the address).  fn(w.detach()) makes a copy the first time; later refreshes go through ddk_pack_jobs."""
    import weakref
    key = (tag, w.data_ptr(), tuple(w.shape))
    e = _pack_cache.get(key)
    if e is not None and e.wref() is w:
        if e.fresh == _state(w):
            return e.out
        if e.job is not None and w.is_contiguous() and _repack_stale() and e.fresh == _state(w):
            return e.out
    e = _PackEntry()
    e.wref, e.out, e.fresh = weakref.ref(w), fn(w.detach()), _state(w)
    e.job = _pack_job(tag, w) if w.is_contiguous() and w.dtype == torch.float32 else None
    e.pooled = bool(w.is_cuda and torch.cuda.is_current_stream_capturing())
    pack_stats["single"] += 1
    _pack_table.clear()           # a new or replaced copy: the device job table (source / destination addresses) is out of date
    _pack_cache[key] = e
    return e.out


def _f32(t):
    if t.dtype != torch.float32:
        raise L.DDKError(f"expected float32, got {t.dtype}")
    return t


# ------------------------------------------------------------------ layout / packing
def nchw_to_nhwc(x, c_pad=None):
    b, c, h, w = x.shape
    c_pad = c if c_pad is None else c_pad
    out = torch.empty((b, h, w, c_pad), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_nchw_to_nhwc(L.ptr(_f32(x)), L.ptr(out), b, c, h, w, c_pad, L.stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, c=None):
    b, h, w, cs = x.shape
    c = cs if c is None else c
    out = torch.empty((b, c, h, w), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_nhwc_to_nchw(L.ptr(_f32(x)), L.ptr(out), b, c, h, w, cs, L.stream()), "nhwc_to_nchw")
    return out


def pad_channels(x, c_pad):
    c = x.shape[-1]
    out = torch.empty((*x.shape[:-1], c_pad), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_pad_channels(L.ptr(_f32(x)), L.ptr(out), x.numel() // c, c, c_pad, L.stream()), "pad_channels")
    return out


def pack_conv_weight(w):
    """OIHW -> [O][KH*KW][pad32(I)]"""
    o, i, kh, kw = w.shape
    out = torch.empty((o, kh * kw, pad32(i)), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv_weight(L.ptr(_f32(w.contiguous())), L.ptr(out), o, i, kh, kw, pad32(i), L.stream()),
            "pack_conv_weight")
    return out


def pack_convT_weight(w):
    """ConvTranspose2d (I,O,4,4) -> [4 phases][O][4 taps][I]"""
    i, o, kh, kw = w.shape
    if (kh, kw) != (4, 4):
        raise L.DDKError("pack_convT_weight: kernel must be 4x4")
    out = torch.empty((4, o, 4, i), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_convT_weight(L.ptr(_f32(w.contiguous())), L.ptr(out), i, o, L.stream()), "pack_convT_weight")
    return out


def pack_conv_weight_wino(w):
    """OIHW 3x3 -> Winograd-domain filter [pad32(I)/32][16][O][32] (ddk_pack_conv_weight_wino)."""
    o, i, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise L.DDKError("pack_conv_weight_wino: kernel must be 3x3")
    out = torch.empty((pad32(i) // 32, 16, o, 32), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv_weight_wino(L.ptr(_f32(w.contiguous())), L.ptr(out), o, i, pad32(i), L.stream()), "pack_conv_weight_wino")
    return out


def pack_convT_weight_wino(w):
    """ConvTranspose2d weight (I, O, 4, 4) -> Winograd F(2x2, 2x2) filters per output phase [4][pad32(I)/32][9][O][32]
    (ddk_pack_convT_weight_wino); pass as conv(CONVT4X4_S2, ..., w_wino=...)."""
    i, o, kh, kw = w.shape
    if (kh, kw) != (4, 4):
        raise L.DDKError("pack_convT_weight_wino: kernel must be 4x4")
    out = torch.empty((4, pad32(i) // 32, 9, o, 32), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_convT_weight_wino(L.ptr(_f32(w.contiguous())), L.ptr(out), i, o, pad32(i), L.stream()), "pack_convT_weight_wino")
    return out


def pack_conv_weight_local(w):
    """OIHW 3x3 -> the operand order of conv3x3_gn_mish (ddk_pack_conv_weight_local): [O/32][9][pad32(I)/32][1024]."""
    o, i, kh, kw = w.shape
    if (kh, kw) != (3, 3) or o % 32:
        raise L.DDKError("pack_conv_weight_local: kernel must be 3x3 and O % 32 == 0")
    out = torch.empty((o // 32, 9, pad32(i) // 32, 1024), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv_weight_local(L.ptr(_f32(w.contiguous())), L.ptr(out), o, i, pad32(i), L.stream()), "pack_conv_weight_local")
    return out


def pack_conv_weight_wino_local(w):
    """OIHW 3x3 -> Winograd-domain filter in the operand order of conv3x3_gn_mish_wino: [O/32][pad32(I)/32][16][1024]."""
    o, i, kh, kw = w.shape
    if (kh, kw) != (3, 3) or o % 32:
        raise L.DDKError("pack_conv_weight_wino_local: kernel must be 3x3 and O % 32 == 0")
    out = torch.empty((o // 32, pad32(i) // 32, 16, 1024), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv_weight_wino_local(L.ptr(_f32(w.contiguous())), L.ptr(out), o, i, pad32(i), L.stream()),
            "pack_conv_weight_wino_local")
    return out


def wino_weight(weight, x_shape, c_lo=None, c_hi=None, dgrad=False):
    """Cached Winograd-domain copy of a canonical OIHW 3x3 filter for a conv on NHWC input shape `x_shape`, or None when the
    shape is not eligible (ddk_conv_wino_splits == 0).  dgrad=True: the filter of the INPUT-gradient conv for input channels
    [c_lo, c_hi) -- g'[ci][n][a][b] = w[n][ci][2-a][2-b] -- which is again a 3x3 stride-1 conv (of dY)."""
    b, h, w_, cin_x = x_shape
    o, i = weight.shape[0], weight.shape[1]
    if tuple(weight.shape[2:]) != (3, 3):
        return None
    if dgrad:
        lo, hi = (0 if c_lo is None else c_lo), (i if c_hi is None else c_hi)
        n_out, cin = hi - lo, o
    else:
        lo, hi, n_out, cin = 0, i, o, pad32(i)
    if L.load().ddk_conv_wino_splits(b, h, w_, pad32(cin), n_out) <= 0:
        return None
    if dgrad:
        def pack(t):       # flip + transpose + Winograd transform in ONE kernel (was three torch kernels + the pack)
            out = torch.empty((pad32(o) // 32, 16, hi - lo, 32), device=t.device, dtype=torch.float32)
            L.check(L.load().ddk_pack_conv_weight_wino_dgrad(L.ptr(_f32(t.contiguous())), L.ptr(out), o, i, lo, hi, pad32(o), L.stream()),
                    "pack_conv_weight_wino_dgrad")
            return out
        return cached_pack(("wino_dgrad", lo, hi), weight, pack)
    return cached_pack("wino", weight, pack_conv_weight_wino)


# ------------------------------------------------------------------ conv family
def conv(kind, x, w_packed, bias=None, n_out=None, x2=None, resid=None, pre_mish=False, post_mish=False, w_wino=None, mish_out=None,
         dmish_src=None, leave_slabs=False):
    """Implicit-GEMM conv on NHWC x (optionally channel-concatenated with x2 without materialising it).
    w_wino: the same 3x3 filter in the Winograd domain -> the F(2x2,3x3) kernel runs when the shape is eligible.
    mish_out: tensor that also receives Mish(out); dmish_src: out = (conv + bias) * Mish'(dmish_src) (+ resid) (ddk_conv_args).
    leave_slabs (plain convs only: no resid / Mish): returns (out, slabs) -- when the launch splits k, `slabs` [S,B,Ho,Wo,N] holds the
    S >= 2 partial sums WITHOUT the bias and `out` is unwritten (ddk_conv_args.defer_reduce: the GroupNorm that follows sums them
    while it loads, ops.groupnorm_mish_train / groupnorm_mish_bwd); otherwise slabs is None and `out` is complete."""
    b, h, w_, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[-1]
    n = n_out if n_out is not None else (w_packed.shape[1] if kind == CONVT4X4_S2 else w_packed.shape[0])
    if kind == CONV3X3_S2:
        ho, wo = (h - 1) // 2 + 1, (w_ - 1) // 2 + 1
    elif kind == CONVT4X4_S2:
        ho, wo = 2 * h, 2 * w_
    elif kind == 4:   # CONV4X4_S2
        ho, wo = h // 2, w_ // 2
    else:
        ho, wo = h, w_
    out = torch.empty((b, ho, wo, n), device=x.device, dtype=torch.float32)
    lib = L.load()
    ws_bytes = lib.ddk_conv_workspace_bytes(kind, b, h, w_, c0 + c1, n)
    if mish_out is not None or dmish_src is not None:
        w_wino = None                                   # these epilogues live on the im2col kernels
    wsplits = 0
    if w_wino is not None and kind == CONV3X3_S1 and not pre_mish:
        wsplits = lib.ddk_conv_wino_splits(b, h, w_, c0 + c1, n)
        if wsplits > 0:
            ws_bytes = wsplits * out.numel() * 4 if wsplits > 1 else 0
    nslab = 1
    if leave_slabs:
        if resid is not None or pre_mish or post_mish or mish_out is not None or dmish_src is not None:
            raise L.DDKError("conv(leave_slabs): only for a plain conv (bias is the consumer's to add)")
        nslab = wsplits if (w_wino is not None and kind == CONV3X3_S1 and wsplits > 0) else lib.ddk_conv_splits(kind, b, h, w_, c0 + c1, n)
        if nslab > 1 and ws_bytes < nslab * out.numel() * 4:
            raise L.DDKError("conv(leave_slabs): internal: workspace smaller than the slabs")
    ws = torch.empty(max(ws_bytes, 16) // 4, device=x.device, dtype=torch.float32) if ws_bytes else None
    a = L.ConvArgs(kind, L.ptr(_f32(x)), L.ptr(x2), c0, c1, L.ptr(w_packed), L.ptr(bias), L.ptr(resid), L.ptr(out),
                   b, h, w_, n, int(pre_mish), int(post_mish), int(leave_slabs and nslab > 1), L.ptr(ws), ws_bytes, L.ptr(w_wino), None, 0,
                   L.ptr(mish_out), L.ptr(dmish_src))
    L.check(lib.ddk_conv_forward(C.byref(a), L.stream()), "conv_forward")
    if leave_slabs:
        return out, (ws[:nslab * out.numel()].view(nslab, b, ho, wo, n) if nslab > 1 else None)
    return out


def conv_with_gn_partials(x, w_packed, bias, w_wino, x2=None, groups=GN_GROUPS):
    """3x3 conv on the Winograd kernel that also emits GroupNorm partials (ddk_conv_args.gn_partials): -> (raw output, partials
    [B * tiles_per_image][groups][2] = {mean, M2} per 128-pixel tile, tiles_per_image).  Only where ddk_conv_gn_partials() > 0."""
    b, h, w_, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[-1]
    n = w_packed.shape[0]
    lib = L.load()
    np_ = lib.ddk_conv_gn_partials(b, h, w_, c0 + c1, n, groups)
    if np_ <= 0:
        raise L.DDKError(f"conv_with_gn_partials: shape {tuple(x.shape)} (+{c1}) -> {n} cannot emit GroupNorm partials")
    raw = torch.empty((b, h, w_, n), device=x.device, dtype=torch.float32)
    part = torch.empty((b * np_, groups, 2), device=x.device, dtype=torch.float32)
    a = L.ConvArgs(CONV3X3_S1, L.ptr(_f32(x)), L.ptr(x2), c0, c1, L.ptr(w_packed), L.ptr(bias), None, L.ptr(raw), b, h, w_, n, 0, 0, 0,
                   None, 0, L.ptr(w_wino), L.ptr(part), groups)
    L.check(lib.ddk_conv_forward(C.byref(a), L.stream()), "conv_forward(gn_partials)")
    return raw, part, np_


def groupnorm_mish_from_partials(x, part, tiles_per_image, gamma, beta, temb=None, addend=None, groups=GN_GROUPS, eps=GN_EPS):
    """GroupNorm+Mish(+temb)(+addend) of x from the partials of conv_with_gn_partials: one read, one write (ddk_groupnorm_mish_partials)."""
    b, h, w_, n = x.shape
    out = torch.empty_like(x)
    stride = temb.stride(0) if temb is not None else 0
    L.check(L.load().ddk_groupnorm_mish_partials(L.ptr(_f32(x)), L.ptr(part), tiles_per_image, L.ptr(gamma), L.ptr(beta),
                                                 temb.data_ptr() if temb is not None else None, stride, L.ptr(addend), L.ptr(out),
                                                 b, h * w_, n, groups, eps, L.stream()), "groupnorm_mish_partials")
    return out


def pack_conv_weight_first(w):
    """OIHW 3x3 with 1 <= I <= 8 -> conv_first's operand order [O/32][ceil(9 I / 2)][64] (ddk_pack_conv_weight_first)."""
    o, i, kh, kw = w.shape
    if (kh, kw) != (3, 3) or o % 32 or not 1 <= i <= 8:
        raise L.DDKError("pack_conv_weight_first: kernel must be 3x3, O % 32 == 0, 1 <= I <= 8")
    out = torch.empty((o // 32, (9 * i + 1) // 2, 64), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv_weight_first(L.ptr(_f32(w.contiguous())), L.ptr(out), o, i, L.stream()), "pack_conv_weight_first")
    return out


def conv_first(x, w_first, bias, n_out, groups=GN_GROUPS, partials=True):
    """The network's first 3x3 conv on the unpadded NHWC input (C_in <= 8): -> (raw output, GroupNorm partials or None, tiles per
    image).  ddk_conv_first."""
    b, h, w_, cin = x.shape
    lib = L.load()
    if not lib.ddk_conv_first_ok(cin, n_out, h, w_, groups):
        raise L.DDKError(f"conv_first: shape {tuple(x.shape)} -> {n_out} not eligible")
    raw = torch.empty((b, h, w_, n_out), device=x.device, dtype=torch.float32)
    tiles = h * w_ // 128
    part = torch.empty((b * tiles, groups, 2), device=x.device, dtype=torch.float32) if partials else None
    L.check(lib.ddk_conv_first(L.ptr(_f32(x)), L.ptr(w_first), L.ptr(bias), L.ptr(raw), L.ptr(part), b, h, w_, cin, n_out, groups,
                               L.stream()), "conv_first")
    return raw, part, tiles


def conv1x1_ws(x, w_packed, bias=None, resid=None, ln=None, eps=1e-5):
    """1x1 conv of a 128-channel NHWC tensor by the weights-stationary kernel (ddk_conv1x1_ws); w_packed [N][128].
    ln = (c1, c2): LayerNorm folded in (w_packed then holds W o g)."""
    b, h, w_, k = x.shape
    n = w_packed.shape[0]
    lib = L.load()
    if not lib.ddk_conv1x1_ws_ok(b * h * w_, k, n):
        raise L.DDKError(f"conv1x1_ws: shape {tuple(x.shape)} -> {n} not eligible")
    out = torch.empty((b, h, w_, n), device=x.device, dtype=torch.float32)
    c1, c2 = ln if ln is not None else (None, None)
    L.check(lib.ddk_conv1x1_ws(L.ptr(_f32(x)), L.ptr(w_packed), L.ptr(bias), L.ptr(resid), L.ptr(out), b * h * w_, n, L.ptr(c1), L.ptr(c2),
                               eps, L.stream()), "conv1x1_ws")
    return out


def attention_folded(x, w_qkv, ln_g, ln_b, w_out, b_out, heads=4, eps=LN_EPS):
    """PreNorm(LinearAttention) + Residual (blocks.py:8-14,57-71,116-134) on a map with more than 256 pixels and C = heads * 32 = 128,
    in the folded form: k, v projection -> context -> per-image matrix W_out ctx^T W_q (ddk_attention_fold) -> one per-image 1x1 conv
    of x with the LayerNorm folded in and the residual added (ddk_conv1x1_ws_images).  x [B,H,W,128]; w_qkv [384,128(,1,1)], w_out
    [128,128(,1,1)], b_out [128] canonical; ln_g / ln_b the channel LayerNorm's g, b."""
    b, h, w_, c = x.shape
    hc = heads * 32
    wq = w_qkv.reshape(3 * hc, c).to(torch.float32)
    g, be = ln_g.reshape(-1).to(torch.float32), ln_b.reshape(-1).to(torch.float32)
    wg = (wq * g.view(1, c)).contiguous()                   # W o g
    c1, c2 = (wq @ g).contiguous(), (wq @ be).contiguous()  # W g, W b
    lib = L.load()
    m = b * h * w_
    kv = torch.empty((b, h, w_, 2 * hc), device=x.device, dtype=torch.float32)
    L.check(lib.ddk_conv1x1_ws(L.ptr(_f32(x)), L.ptr(wg[hc:].contiguous()), None, None, L.ptr(kv), m, 2 * hc, L.ptr(c1[hc:].contiguous()),
                               L.ptr(c2[hc:].contiguous()), eps, L.stream()), "conv1x1_ws(kv)")
    ctx = torch.empty((b, heads, 32, 32), device=x.device, dtype=torch.float32)
    nbytes = lib.ddk_linattn_context_workspace_bytes(b, h * w_, heads)
    ws = _ws(x.device, nbytes, "linattn") if nbytes else None
    L.check(lib.ddk_linattn_context_kv(L.ptr(kv), L.ptr(ctx), b, h * w_, heads, L.ptr(ws), nbytes, L.stream()), "linattn_context_kv")
    a_mat = torch.empty((b, c, c), device=x.device, dtype=torch.float32)
    a1 = torch.empty((b, c), device=x.device, dtype=torch.float32)
    a2 = torch.empty((b, c), device=x.device, dtype=torch.float32)
    L.check(lib.ddk_attention_fold(L.ptr(ctx), L.ptr(wg[:hc].contiguous()), L.ptr(c1[:hc].contiguous()), L.ptr(c2[:hc].contiguous()),
                                   L.ptr(w_out.reshape(c, hc).contiguous()), L.ptr(b_out), L.ptr(a_mat), L.ptr(a1), L.ptr(a2), b, c, heads,
                                   L.stream()), "attention_fold")
    out = torch.empty_like(x)
    L.check(lib.ddk_conv1x1_ws_images(L.ptr(x), L.ptr(a_mat), None, L.ptr(x), L.ptr(out), m, c, L.ptr(a1), L.ptr(a2), eps, b, L.stream()),
            "conv1x1_ws_images")
    return out


def attention_kv_context(x, w_qkv, ln_g, ln_b, heads=4, eps=LN_EPS):
    """ctx [B,4,32,32] of PreNorm(LinearAttention) (blocks.py:57-60,123,129-131) from x [B,H,W,128] in ONE launch + the split merge
    (ddk_attention_kv_context): the k and v thirds of to_qkv with the LayerNorm folded in, softmax over the pixels of k, k v^T -- no kv
    tensor.  w_qkv [384,128(,1,1)] canonical; ln_g / ln_b the channel LayerNorm's g, b."""
    b, h, w_, c = x.shape
    hc = heads * 32
    lib = L.load()
    if not lib.ddk_attention_kv_context_ok(b, h * w_, c, heads):
        raise L.DDKError(f"attention_kv_context: shape {tuple(x.shape)} with {heads} heads not eligible (C = 128, 4 heads, H*W % 64 == 0)")
    wq = w_qkv.reshape(3 * hc, c).to(torch.float32)
    g, be = ln_g.reshape(-1).to(torch.float32), ln_b.reshape(-1).to(torch.float32)
    wkv = (wq * g.view(1, c))[hc:].contiguous()
    c1, c2 = (wq @ g)[hc:].contiguous(), (wq @ be)[hc:].contiguous()
    ctx = torch.empty((b, heads, 32, 32), device=x.device, dtype=torch.float32)
    nbytes = lib.ddk_attention_kv_context_workspace_bytes(b, h * w_)
    ws = _ws(x.device, nbytes, "kvctx")
    L.check(lib.ddk_attention_kv_context(L.ptr(_f32(x)), L.ptr(wkv), L.ptr(c1), L.ptr(c2), eps, L.ptr(ctx), b, h * w_, L.ptr(ws), nbytes,
                                         L.stream()), "attention_kv_context")
    return ctx


def groupnorm_mish_from_partials_res1x1(x, part, tiles_per_image, gamma, beta, res_x, res_w, res_b, temb=None, groups=GN_GROUPS,
                                        eps=GN_EPS):
    """groupnorm_mish_from_partials with the addend res_b + res_x @ res_w^T (a 1x1 conv of a <= 8-channel tensor) computed on the
    fly (ddk_groupnorm_mish_partials_res1x1)."""
    b, h, w_, n = x.shape
    cin = res_x.shape[-1]
    out = torch.empty_like(x)
    stride = temb.stride(0) if temb is not None else 0
    L.check(L.load().ddk_groupnorm_mish_partials_res1x1(
        L.ptr(_f32(x)), L.ptr(part), tiles_per_image, L.ptr(gamma), L.ptr(beta), temb.data_ptr() if temb is not None else None, stride,
        L.ptr(_f32(res_x)), L.ptr(res_w.reshape(n, cin).contiguous()), L.ptr(res_b), cin, L.ptr(out), b, h * w_, n, groups, eps,
        L.stream()), "groupnorm_mish_partials_res1x1")
    return out


def final_tail(raw, part, tiles_per_image, gamma, beta, w, bias, x=None, t=None, tables=None, noise=None, seed=0, stream_id=0,
               want_eps=True, groups=GN_GROUPS, eps=GN_EPS):
    """GroupNorm (from conv partials) + Mish + 1x1 projection to n_out <= 8 channels, and -- with x, t, tables -- the in-place
    reverse-step update of x, in one launch (ddk_final_tail).  Returns eps_hat (or None when want_eps is False)."""
    b, h, w_, c = raw.shape
    n_out = w.shape[0]
    eps_out = torch.empty((b, h, w_, n_out), device=raw.device, dtype=torch.float32) if want_eps else None
    tb = tables or {}
    L.check(L.load().ddk_final_tail(L.ptr(_f32(raw)), L.ptr(part), tiles_per_image, L.ptr(gamma), L.ptr(beta), eps,
                                    L.ptr(w.reshape(n_out, c).contiguous()), L.ptr(bias), n_out, L.ptr(eps_out), L.ptr(x), L.ptr(noise),
                                    L.ptr(t), L.ptr(tb.get("c_recip")), L.ptr(tb.get("c_recipm1")), L.ptr(tb.get("c1")),
                                    L.ptr(tb.get("c2")), L.ptr(tb.get("sigma")), seed, stream_id, b, h * w_, c, groups, L.stream()),
            "final_tail")
    return eps_out


def conv3x3_gn_mish_cluster(x, w_wino, bias, gamma, beta, x2=None, temb=None, addend=None, groups=GN_GROUPS, eps=GN_EPS, check=True):
    """Block (conv3x3 + GroupNorm + Mish + shift + residual) in ONE Winograd launch whose workgroups exchange tile statistics
    (ddk_conv3x3_gn_mish_cluster); raises when the shape or the device is not eligible.  check=True (default) waits for the
    stream and raises DDKError when an exchange timed out (shared GPU): the output then holds NaN tiles and must not be used;
    timing loops pass check=False and call cluster_check() once at their end."""
    b, h, w_, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[-1]
    n = w_wino.shape[2]
    lib = L.load()
    if lib.ddk_conv3x3_gn_mish_cluster_ok(b, h, w_, c0 + c1, n, groups) <= 0:
        raise L.DDKError(f"conv3x3_gn_mish_cluster: shape {tuple(x.shape)} (+{c1}) -> {n} not eligible")
    nbytes = max(lib.ddk_conv3x3_gn_mish_cluster_workspace_bytes(b, h, w_, n),
                 lib.ddk_conv3x3_gn_mish_cluster_split_workspace_bytes(b, h, w_, c0 + c1, n, groups))      # (the k-split kind of shape)
    ws = _ws(x.device, nbytes, "cluster")
    out = torch.empty((b, h, w_, n), device=x.device, dtype=torch.float32)
    stride = temb.stride(0) if temb is not None else 0
    L.check(lib.ddk_conv3x3_gn_mish_cluster(L.ptr(_f32(x)), c0, L.ptr(x2), c1, L.ptr(w_wino), L.ptr(bias), L.ptr(gamma), L.ptr(beta),
                                            temb.data_ptr() if temb is not None else None, stride, L.ptr(addend), L.ptr(out), b, h, w_, n,
                                            groups, eps, L.ptr(ws), nbytes, L.stream()), "conv3x3_gn_mish_cluster")
    if check:
        L.check(lib.ddk_conv3x3_gn_mish_cluster_check(L.ptr(ws), b, L.stream()), "conv3x3_gn_mish_cluster")
    return out


def cluster_check(device, b):
    """ddk_conv3x3_gn_mish_cluster_check on the scratch the wrapper above used on `device` (a tensor's .device) and the current
    stream: raises DDKError if any launch since the last check gave up, and if no such scratch exists (nothing to check is an
    error in a timing loop, not a pass)."""
    ws = _scratch.get((str(device), "cluster", torch.cuda.current_stream().cuda_stream))
    if ws is None:
        raise L.DDKError(f"cluster_check: no cluster workspace on {device} for the current stream")
    L.check(L.load().ddk_conv3x3_gn_mish_cluster_check(L.ptr(ws), b, L.stream()), "conv3x3_gn_mish_cluster")


class ClockProbe:
    """Shader clock held over a region of a stream: probe() before and after, then ghz().  (ddk_debug_clock_probe: one record per
    workgroup {XCC, s_memtime, s_memrealtime}; records of the same XCC are differenced, the median over XCCs is returned.)"""

    def __init__(self, device, workgroups=64):
        self.n = workgroups
        self.a = torch.zeros((workgroups, 4), device=device, dtype=torch.int64)
        self.b = torch.zeros((workgroups, 4), device=device, dtype=torch.int64)
        self._first = True

    def probe(self):
        buf = self.a if self._first else self.b
        self._first = False
        L.check(L.load().ddk_debug_clock_probe(buf.data_ptr(), self.n, L.stream()), "debug_clock_probe")

    def ghz(self):
        """Median over the workgroups whose two records come from the same XCC (workgroup i of both probe launches: the same dispatch
        slot) of d(s_memtime) / d(s_memrealtime).  Pairing records of different workgroups of an XCC (rounds 4-5 took the first record
        per XCC of each probe) mixes counters that are a constant apart: invisible over a second, 30 % off over 25 ms."""
        a, b = self.a.cpu().numpy(), self.b.cpu().numpy()
        rates = []
        for i in range(min(len(a), len(b))):
            if int(a[i, 3]) == 1 and int(b[i, 3]) == 1 and int(a[i, 0]) == int(b[i, 0]):
                dc, dt = int(b[i, 1]) - int(a[i, 1]), int(b[i, 2]) - int(a[i, 2])
                if dt > 0 and dc > 0:
                    rates.append(dc / dt * 0.1)        # cycles per 10 ns tick -> GHz
        rates.sort()
        return rates[len(rates) // 2] if rates else None


def cluster_timeouts():
    return int(L.load().ddk_debug_cluster_timeouts())


def conv3x3_groupnorm_mish(x, w_packed, bias, gamma, beta, x2=None, temb=None, addend=None, groups=GN_GROUPS, eps=GN_EPS, w_wino=None):
    """conv3x3(pad 1) -> GroupNorm -> Mish (+temb)(+addend), two launches.  When the conv splits k its partial slabs are summed by
    the GroupNorm kernel's load (ddk_groupnorm_mish_slabs) instead of a separate reduce pass.  w_wino: the Winograd-domain filter
    (pack_conv_weight_wino) -- the conv then runs on the Winograd kernel where the shape is eligible."""
    b, h, w_, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[-1]
    n = w_packed.shape[0]
    lib = L.load()
    wino = w_wino is not None and lib.ddk_conv_wino_splits(b, h, w_, c0 + c1, n) > 0
    if wino and lib.ddk_conv_gn_partials(b, h, w_, c0 + c1, n, groups) > 0:
        # one-pass Winograd conv leaves per-tile {mean, M2}; GroupNorm reads the tensor once
        raw, part, np_ = conv_with_gn_partials(x, w_packed, bias, w_wino, x2=x2, groups=groups)
        return groupnorm_mish_from_partials(raw, part, np_, gamma, beta, temb=temb, addend=addend, groups=groups, eps=eps)
    splits = lib.ddk_conv_wino_splits(b, h, w_, c0 + c1, n) if wino else lib.ddk_conv_splits(CONV3X3_S1, b, h, w_, c0 + c1, n)
    if splits == 1 or lib.ddk_groupnorm_workspace_bytes(b, h * w_, n, groups) != 0:
        return groupnorm_mish(conv(CONV3X3_S1, x, w_packed, bias, x2=x2, w_wino=w_wino), gamma, beta, temb=temb, addend=addend,
                              groups=groups, eps=eps)
    ws_bytes = splits * b * h * w_ * n * 4 if wino else lib.ddk_conv_workspace_bytes(CONV3X3_S1, b, h, w_, c0 + c1, n)
    ws = torch.empty(ws_bytes // 4, device=x.device, dtype=torch.float32)
    out = torch.empty((b, h, w_, n), device=x.device, dtype=torch.float32)
    a = L.ConvArgs(CONV3X3_S1, L.ptr(_f32(x)), L.ptr(x2), c0, c1, L.ptr(w_packed), None, None, L.ptr(out), b, h, w_, n, 0, 0, 1,
                   L.ptr(ws), ws_bytes, L.ptr(w_wino) if wino else None)
    L.check(lib.ddk_conv_forward(C.byref(a), L.stream()), "conv_forward(defer)")
    stride = temb.stride(0) if temb is not None else 0
    L.check(lib.ddk_groupnorm_mish_slabs(L.ptr(ws), splits, b * h * w_ * n, L.ptr(bias), L.ptr(gamma), L.ptr(beta),
                                         temb.data_ptr() if temb is not None else None, stride, L.ptr(addend), L.ptr(out),
                                         b, h * w_, n, groups, eps, L.stream()), "groupnorm_mish_slabs")
    return out


# ------------------------------------------------------------------ norms / activations
def groupnorm_mish(x, gamma, beta, temb=None, addend=None, groups=GN_GROUPS, eps=GN_EPS):
    b, h, w, c = x.shape
    out = torch.empty_like(x)
    lib = L.load()
    ws_bytes = lib.ddk_groupnorm_workspace_bytes(b, h * w, c, groups)
    ws = torch.empty(max(ws_bytes, 16) // 4, device=x.device, dtype=torch.float32) if ws_bytes else None
    stride = temb.stride(0) if temb is not None else 0
    L.check(lib.ddk_groupnorm_mish(L.ptr(_f32(x)), L.ptr(gamma), L.ptr(beta),
                                   temb.data_ptr() if temb is not None else None, stride, L.ptr(addend), L.ptr(out),
                                   b, h * w, c, groups, eps, L.ptr(ws), ws_bytes, L.stream()), "groupnorm_mish")
    return out


def conv3x3_gn_mish(x, w_local, bias, gamma, beta, temb=None, addend=None, x2=None, groups=GN_GROUPS, eps=GN_EPS):
    """Block (blocks.py:75-84) in one launch on a 4x4 / 8x8 map: x [B,H,W,c0] (+ x2 [B,H,W,c1], the concat of unet.py:97),
    w_local = pack_conv_weight_local(weight) -> Mish(GN(conv(x) + bias)) (+ temb[b]) (+ addend)."""
    b, h, w, c0 = x.shape
    c1 = x2.shape[-1] if x2 is not None else 0
    n = w_local.shape[0] * 32
    w_packed = w_local
    if w_local.shape[2] * 32 != c0 + c1:
        raise L.DDKError(f"conv3x3_gn_mish: filter packed for {w_local.shape[2] * 32} input channels, input has {c0 + c1}")
    lib = L.load()
    if not lib.ddk_conv3x3_gn_mish_ok(h, w, c0 + c1, c0, n, groups):
        raise L.DDKError(f"conv3x3_gn_mish: shape {tuple(x.shape)} (+{c1}) -> {n} not eligible")
    out = torch.empty((b, h, w, n), device=x.device, dtype=torch.float32)
    stride = temb.stride(0) if temb is not None else 0
    L.check(lib.ddk_conv3x3_gn_mish(L.ptr(_f32(x)), c0, L.ptr(_f32(x2)) if x2 is not None else None, c1, L.ptr(w_packed), L.ptr(bias),
                                    L.ptr(gamma), L.ptr(beta), temb.data_ptr() if temb is not None else None, stride, L.ptr(addend),
                                    L.ptr(out), b, h, w, n, groups, eps, L.stream()), "conv3x3_gn_mish")
    return out


def conv3x3_gn_mish_wino(x, w_wl, bias, gamma, beta, temb=None, addend=None, x2=None, groups=GN_GROUPS, eps=GN_EPS):
    """Block (blocks.py:75-84) in one launch on a 64-pixel map (8x8), Winograd form; w_wl = pack_conv_weight_wino_local(weight)."""
    b, h, w, c0 = x.shape
    c1 = x2.shape[-1] if x2 is not None else 0
    n = w_wl.shape[0] * 32
    lib = L.load()
    if w_wl.shape[1] * 32 != c0 + c1:
        raise L.DDKError(f"conv3x3_gn_mish_wino: filter packed for {w_wl.shape[1] * 32} input channels, input has {c0 + c1}")
    if not lib.ddk_conv3x3_gn_mish_wino_ok(h, w, c0 + c1, c0, n, groups):
        raise L.DDKError(f"conv3x3_gn_mish_wino: shape {tuple(x.shape)} (+{c1}) -> {n} not eligible")
    out = torch.empty((b, h, w, n), device=x.device, dtype=torch.float32)
    stride = temb.stride(0) if temb is not None else 0
    L.check(lib.ddk_conv3x3_gn_mish_wino(L.ptr(_f32(x)), c0, L.ptr(_f32(x2)) if x2 is not None else None, c1, L.ptr(w_wl), L.ptr(bias),
                                         L.ptr(gamma), L.ptr(beta), temb.data_ptr() if temb is not None else None, stride, L.ptr(addend),
                                         L.ptr(out), b, h, w, n, groups, eps, L.stream()), "conv3x3_gn_mish_wino")
    return out


def conv3x3_gn_mish_slabs(src_slabs, src_bias, w_packed, bias, gamma, beta, temb=None, addend_slabs=None, addend_bias=None, groups=GN_GROUPS,
                          eps=GN_EPS):
    """The one-launch Block of the 4x4 / 8x8 maps on operands still in split-K form (ddk_conv3x3_gn_mish_slabs): src_slabs [S][B][H][W][C]
    partial sums of the input (+ src_bias[c]), addend_slabs [S'][B][H][W][N] of the residual (+ addend_bias[c]) or None.  w_packed:
    pack_conv_weight_local (H*W == 16) / pack_conv_weight_wino_local (H*W == 64)."""
    s_, b, h, w, c0 = src_slabs.shape
    n = w_packed.shape[0] * 32
    out = torch.empty((b, h, w, n), device=src_slabs.device, dtype=torch.float32)
    stride = temb.stride(0) if temb is not None else 0
    a_n, a_stride = (addend_slabs.shape[0], addend_slabs.stride(0)) if addend_slabs is not None else (1, 0)
    L.check(L.load().ddk_conv3x3_gn_mish_slabs(L.ptr(_f32(src_slabs)), s_, src_slabs.stride(0), L.ptr(src_bias), c0, L.ptr(w_packed), L.ptr(bias),
                                               L.ptr(gamma), L.ptr(beta), temb.data_ptr() if temb is not None else None, stride,
                                               L.ptr(addend_slabs), a_n, a_stride, L.ptr(addend_bias), L.ptr(out), b, h, w, n, groups, eps,
                                               L.stream()), "conv3x3_gn_mish_slabs")
    return out


def chan_layernorm(x, g, b, eps=LN_EPS):
    c = x.shape[-1]
    out = torch.empty_like(x)
    L.check(L.load().ddk_chan_layernorm(L.ptr(_f32(x)), L.ptr(g.reshape(-1)), L.ptr(b.reshape(-1)), L.ptr(out),
                                        x.numel() // c, c, eps, L.stream()), "chan_layernorm")
    return out


def mish(x):
    out = torch.empty_like(x)
    L.check(L.load().ddk_mish(L.ptr(_f32(x)), L.ptr(out), x.numel(), L.stream()), "mish")
    return out


def tanh(x):
    out = torch.empty_like(x)
    L.check(L.load().ddk_tanh(L.ptr(_f32(x)), L.ptr(out), x.numel(), L.stream()), "tanh")
    return out


def add(a, b):
    out = torch.empty_like(a)
    L.check(L.load().ddk_add(L.ptr(_f32(a)), L.ptr(_f32(b)), L.ptr(out), a.numel(), L.stream()), "add")
    return out


def avgpool2(x):
    b, h, w, c = x.shape
    out = torch.empty((b, h // 2, w // 2, c), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_avgpool2(L.ptr(_f32(x)), L.ptr(out), b, h, w, c, L.stream()), "avgpool2")
    return out


def upsample_nearest2(x):
    b, h, w, c = x.shape
    out = torch.empty((b, 2 * h, 2 * w, c), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_upsample_nearest2(L.ptr(_f32(x)), L.ptr(out), b, h, w, c, L.stream()), "upsample_nearest2")
    return out


def fix_samples(x):
    """x [B,C,H,W] device fp32 -> [B,H,W,C] device fp32 in [0,255] (per-image min-max, utils/eval_helpers.py:37-41)."""
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_fix_samples(L.ptr(_f32(x)), L.ptr(out), b, c, h, w, L.stream()), "fix_samples")
    return out


# ------------------------------------------------------------------ linear attention core
def linattn(qkv, heads=4, fused_up_to=256):
    """qkv [B,H,W,3*heads*32] -> attention output [B,H,W,heads*32] (before to_out).  Maps of up to `fused_up_to` pixels (<= 256)
    run context + apply in one launch, as in the UNet plan."""
    b, h, w, c3 = qkv.shape
    ctx = torch.empty((b, heads, 32, 32), device=qkv.device, dtype=torch.float32)
    out = torch.empty((b, h, w, c3 // 3), device=qkv.device, dtype=torch.float32)
    lib = L.load()
    if h * w <= min(fused_up_to, 256):      # small maps: context + apply in one launch
        L.check(lib.ddk_linattn_fused_small(L.ptr(_f32(qkv)), L.ptr(ctx), L.ptr(out), b, h * w, heads, L.stream()), "linattn_fused_small")
        return out, ctx
    nbytes = lib.ddk_linattn_context_workspace_bytes(b, h * w, heads)
    ws = torch.empty(nbytes // 4, device=qkv.device, dtype=torch.float32) if nbytes else None
    L.check(lib.ddk_linattn_context(L.ptr(_f32(qkv)), L.ptr(ctx), b, h * w, heads, L.ptr(ws), nbytes, L.stream()), "linattn_context")
    L.check(lib.ddk_linattn_apply(L.ptr(qkv), L.ptr(ctx), L.ptr(out), b, h * w, heads, L.stream()), "linattn_apply")
    return out, ctx


# ------------------------------------------------------------------ time embedding
def time_mlp(t, freqs, w1t, b1, w2t, b2):
    """t int64 [B] -> (Mish(time_mlp(t)) [B,dim], time_mlp(t) [B,dim])"""
    bsz, dim = t.shape[0], w2t.shape[1]
    act = torch.empty((bsz, dim), device=t.device, dtype=torch.float32)
    raw = torch.empty((bsz, dim), device=t.device, dtype=torch.float32)
    if t.dtype != torch.int64:
        raise L.DDKError("time_mlp: t must be int64")
    L.check(L.load().ddk_time_mlp(L.ptr(t), L.ptr(freqs), L.ptr(w1t), L.ptr(b1), L.ptr(w2t), L.ptr(b2), L.ptr(act),
                                  L.ptr(raw), bsz, dim, L.stream()), "time_mlp")
    return act, raw


def time_proj(act, wt, bias):
    bsz, dim = act.shape
    n_out = wt.shape[1]
    out = torch.empty((bsz, n_out), device=act.device, dtype=torch.float32)
    L.check(L.load().ddk_time_proj(L.ptr(_f32(act)), L.ptr(wt), L.ptr(bias), L.ptr(out), bsz, dim, n_out, L.stream()),
            "time_proj")
    return out


def conv1x1_small_n(x, w, bias):
    c = x.shape[-1]
    n_out = w.shape[0]
    out = torch.empty((*x.shape[:-1], n_out), device=x.device, dtype=torch.float32)
    L.check(L.load().ddk_conv1x1_small_n(L.ptr(_f32(x)), L.ptr(w.reshape(n_out, c)), L.ptr(bias), L.ptr(out),
                                         x.numel() // c, c, n_out, L.stream()), "conv1x1_small_n")
    return out


# ------------------------------------------------------------------ noise-schedule arithmetic
def q_sample(x, eps, t, sqrt_acp, sqrt_1m_acp):
    out = torch.empty_like(x)
    b = x.shape[0]
    L.check(L.load().ddk_q_sample(L.ptr(_f32(x)), L.ptr(_f32(eps)), L.ptr(t), L.ptr(sqrt_acp), L.ptr(sqrt_1m_acp),
                                  L.ptr(out), b, x.numel() // b, L.stream()), "q_sample")
    return out


def p_sample_update_(x, eps_hat, t, c_recip, c_recipm1, c1, c2, sigma, noise=None, seed=0, stream_id=0):
    """In-place reverse-step update of x (any layout, as long as x / eps_hat / noise agree)."""
    b = x.shape[0]
    L.check(L.load().ddk_p_sample_update(L.ptr(_f32(x)), L.ptr(_f32(eps_hat)), L.ptr(noise), L.ptr(t), L.ptr(c_recip),
                                         L.ptr(c_recipm1), L.ptr(c1), L.ptr(c2), L.ptr(sigma), b, x.numel() // b,
                                         seed, stream_id, L.stream()), "p_sample_update")
    return x


def randn(shape, device, seed, step, stream_id=0):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    L.check(L.load().ddk_randn(L.ptr(out), out.numel(), seed, step, stream_id, L.stream()), "randn")
    return out


def vlb_terms(x, x_t, eps_hat, t, c_recip, c_recipm1, c1, c2, post_logvar, eps=None):
    """Per-sample VLB term in bits/dim (KL for t > 0, discretised NLL at t == 0) and, with eps, sum (eps - eps_hat)^2."""
    b = x.shape[0]
    vlb = torch.empty((b,), device=x.device, dtype=torch.float32)
    sq = torch.empty((b,), device=x.device, dtype=torch.float32) if eps is not None else None
    lib = L.load()
    nbytes = lib.ddk_vlb_terms_workspace_bytes(b, x.numel() // b)
    ws = _ws(x.device, nbytes, "vlb")
    L.check(lib.ddk_vlb_terms(L.ptr(_f32(x)), L.ptr(_f32(x_t)), L.ptr(_f32(eps_hat)), L.ptr(eps), L.ptr(t), L.ptr(c_recip),
                              L.ptr(c_recipm1), L.ptr(c1), L.ptr(c2), L.ptr(post_logvar), L.ptr(vlb), L.ptr(sq), b,
                              x.numel() // b, L.ptr(ws), nbytes, L.stream()), "vlb_terms")
    return vlb, sq


def sq_err_sum(a, b):
    bsz = a.shape[0]
    out = torch.empty((bsz,), device=a.device, dtype=torch.float32)
    L.check(L.load().ddk_sq_err_sum(L.ptr(_f32(a)), L.ptr(_f32(b)), L.ptr(out), bsz, a.numel() // bsz, L.stream()),
            "sq_err_sum")
    return out


# ================================================================== training path (backward kernels)
CONV4X4_S2 = 4
_scratch = {}


def _ws(device, nbytes, tag="ws"):
    """Reusable scratch buffer per (device, tag): backward kernels need short-lived workspaces.

    Under stream capture the buffer comes from the capturing graph's PRIVATE memory pool instead (a plain allocation while
    capturing): the graph then owns every scratch address its kernels were recorded with, for as long as it lives.  Handing
    out the shared buffer would bake its address into the graph while a later, larger eager request replaces (and frees)
    it -- the replay would write into memory that belongs to someone else."""
    n = max(nbytes, 16) // 4 + 4
    if device is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return torch.empty(n, device=device, dtype=torch.float32)
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream)     # per stream: two branches of a step may run side by side
    buf = _scratch.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, device=device, dtype=torch.float32)
        _scratch[key] = buf
    return buf


def pack_conv_weight_dgrad(w, i_pad=None, o_pad=None):
    """OIHW -> [I_pad][KH*KW][O_pad] with flipped taps: operand of the input-gradient conv."""
    o, i, kh, kw = w.shape
    i_pad = pad32(i) if i_pad is None else i_pad
    o_pad = o if o_pad is None else o_pad
    out = torch.empty((i_pad, kh * kw, o_pad), device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv_weight_dgrad(L.ptr(_f32(w.contiguous())), L.ptr(out), o, i, kh, kw, i_pad, o_pad, L.stream()),
            "pack_conv_weight_dgrad")
    return out


def zero_stuff2(dy, h_out, w_out):
    b, h, w, c = dy.shape
    out = torch.empty((b, h_out, w_out, c), device=dy.device, dtype=torch.float32)
    L.check(L.load().ddk_zero_stuff2(L.ptr(_f32(dy)), L.ptr(out), b, h, w, h_out, w_out, c, L.stream()), "zero_stuff2")
    return out


# The slab reduces of a backward pass in one launch.  Inside `with deferred_wgrad():` every conv_wgrad_ keeps its slabs in a buffer of
# its own and only records the reduce; leaving the block (or flush_wgrad()) runs them all with ddk_wgrad_reduce_jobs -- 232 reduce
# launches of 4-7 us per cfg3 optimiser step before.  Nothing may read the gradients in between (the trainers wrap exactly one
# forward + backward); outside the block every call reduces at once, as before.
_wgrad_pending = None         # None: immediate reduces; list of (job struct, slab tensor) while deferring
_rows_pending = []            # (job struct, rows tensor): the GroupNorm / LayerNorm parameter-gradient row sums deferred alongside


# The deferred weight-gradient launches of a backward pass write only their own slabs, which nothing reads before the reduce at the end of
# the pass: they run on a SIDE stream, forked from the caller's stream after the tensors they read were produced and joined in front
# of the reduce -- beside the input-gradient / GroupNorm chain instead of inside it (a captured step keeps the two branches: most of
# these launches are small and latency-bound, and so is the chain).  The tensors a side launch reads are kept alive until the join.
WGRAD_SIDE_STREAM = True
_wgrad_side = {}              # device index -> torch.cuda.Stream
_wgrad_keep = []              # tensors read by side-stream launches since the last join
_wgrad_forked = None          # the side stream with launches the caller's stream has not waited for yet


def _side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _wgrad_side.get(idx)
    if st is None:
        st = _wgrad_side[idx] = torch.cuda.Stream(device=device)
    return st


WGRAD_SIDE_BATCH = 16            # 8 .. 64 measure alike (16.2 - 16.6 ms per cfg3 step against 16.7 - 16.8 on the caller's stream)
WGRAD_SIDE_MAX_PIXELS = 65536    # larger maps fill the chip by themselves: nothing to overlap, and the forks cost (cfg5 115.4 -> 117.1 ms)
_wgrad_queue = []             # deferred weight-gradient launches not yet issued


def _launch_side_queue():
    """issue the queued launches on the side stream, behind one fork from the caller's stream"""
    global _wgrad_forked
    if not _wgrad_queue:
        return
    lib = L.load()
    side = _side_stream(_wgrad_queue[0][1].device)
    seen = set()
    for entry in _wgrad_queue:                                 # operands and slab buffers are ready on the stream(s) that queued them
        st = entry[-1]
        if st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            side.wait_stream(st)
    with torch.cuda.stream(side):
        for kind, x, dy, grad_w, grad_b, dims, slab, nbytes, job, _ in _wgrad_queue:
            L.check(lib.ddk_conv_wgrad_defer(kind, L.ptr(x), L.ptr(dy), L.ptr(grad_w), L.ptr(grad_b), *dims, L.ptr(slab), nbytes,
                                             C.byref(job), L.stream()), "conv_wgrad_defer")
            _wgrad_keep.append((x, dy))
    del _wgrad_queue[:]
    _wgrad_forked = side


def _join_side():
    global _wgrad_forked
    _launch_side_queue()
    if _wgrad_forked is not None:
        torch.cuda.current_stream().wait_stream(_wgrad_forked)
        _wgrad_forked = None


class deferred_wgrad:
    def __enter__(self):
        global _wgrad_pending
        self._outer = _wgrad_pending
        if _wgrad_pending is None:
            _wgrad_pending = []
        return self

    def __exit__(self, exc_type, exc, tb):
        global _wgrad_pending
        if self._outer is None:
            try:
                if exc_type is None:
                    flush_wgrad()
            finally:
                if exc_type is not None:
                    del _wgrad_queue[:]
                _join_side()
                del _wgrad_keep[:]
                _wgrad_pending = None
                del _rows_pending[:]
        return False


def _rows_sum_targets(rows, nbatch, batch_stride, nrows, row_stride, targets, n):
    """targets[k][:n] += sum_r rows[k * batch_stride + r * row_stride + :n] (ddk_rows_sum_targets); inside deferred_wgrad() the sum is
    only recorded (`rows` is kept alive) and runs with the others when the block ends"""
    if _wgrad_pending is not None:
        job = L.RowsSumJob()
        job.rows = rows.data_ptr()
        for k in range(4):
            job.out[k] = targets[k].data_ptr() if k < len(targets) and targets[k] is not None else None
        job.batch_stride, job.row_stride, job.nbatch, job.nrows, job.n = batch_stride, row_stride, nbatch, nrows, n
        _rows_pending.append((job, rows))
        return
    t = [L.ptr(x) for x in (list(targets) + [None] * 4)[:4]]
    L.check(L.load().ddk_rows_sum_targets(L.ptr(rows), nbatch, batch_stride, nrows, row_stride, t[0], t[1], t[2], t[3], n, 1, L.stream()),
            "rows_sum_targets")


def _flush_rows():
    global _rows_pending
    pending, _rows_pending = _rows_pending, []
    lib = L.load()
    while pending:
        seen, batch, rest = set(), [], []
        for job, rows in pending:            # two sums into the same gradient must not share a launch
            keys = [p for p in job.out if p]
            if any(k in seen for k in keys):
                rest.append((job, rows))
            else:
                batch.append((job, rows))
                seen.update(keys)
        jobs = (L.RowsSumJob * len(batch))(*[j for j, _ in batch])
        L.check(lib.ddk_rows_sum_jobs(jobs, len(batch), L.stream()), "rows_sum_jobs")
        pending = rest


def flush_wgrad():
    """run the recorded reduces: one launch per set of jobs with pairwise distinct targets (normally one)"""
    global _wgrad_pending
    if _rows_pending:
        _flush_rows()
    _join_side()
    if not _wgrad_pending:
        del _wgrad_keep[:]
        return 0
    pending, _wgrad_pending = _wgrad_pending, []
    lib = L.load()
    launches = 0
    while pending:
        seen, batch, rest = set(), [], []
        for job, slab in pending:        # two reduces into the same gradient elements must not share a launch (no such pair in this model)
            keys = [(job.grad, job.c_off)] + ([("b", job.grad_b)] if job.grad_b else [])
            if any(k in seen for k in keys):
                rest.append((job, slab))
            else:
                batch.append((job, slab))
                seen.update(keys)
        jobs = (L.WgradReduceJob * len(batch))(*[j for j, _ in batch])
        L.check(lib.ddk_wgrad_reduce_jobs(jobs, len(batch), L.stream()), "wgrad_reduce_jobs")     # records travel as kernel arguments
        launches += 1
        pending = rest
    del _wgrad_keep[:]          # (freed on the caller's stream, which has waited for the side stream)
    return launches


def conv_wgrad_(kind, x, dy, grad_w, c_real, cw, c_off, grad_b=None, persistent=False):
    """grad_w (canonical layout, contiguous) += weight gradient of `kind` for the source x (see csrc/conv_wgrad.hip);
    grad_b [N] += column sums of dy when given (the bias gradient, produced by the same launches).
    persistent: grad_w / grad_b are buffers that outlive the backward pass (the parameters' `.grad` slots) -- only then may the
    reduce be deferred inside deferred_wgrad(); a temporary handed back to autograd is reduced at once."""
    b, h, w, cx = x.shape
    n = dy.shape[-1]
    lib = L.load()
    nbytes = lib.ddk_conv_wgrad_workspace_bytes(kind, b, h, w, cx, n)
    if _wgrad_pending is not None and persistent:
        global _wgrad_forked
        slab = torch.empty(max(nbytes, 16) // 4, device=x.device, dtype=torch.float32)
        job = L.WgradReduceJob()
        x, dy = _f32(x), _f32(dy)
        if WGRAD_SIDE_STREAM and b * h * w <= WGRAD_SIDE_MAX_PIXELS:
            # queued; every WGRAD_SIDE_BATCH of them leave together behind ONE fork (a cross-stream edge per launch cost more than the
            # overlap gave: 16.7 -> 18.2 ms per cfg3 step)
            _wgrad_queue.append((kind, x, dy, grad_w, grad_b, (b, h, w, cx, c_real, cw, c_off, n), slab, nbytes, job,
                                 torch.cuda.current_stream()))
            if len(_wgrad_queue) >= WGRAD_SIDE_BATCH:
                _launch_side_queue()
        else:
            L.check(lib.ddk_conv_wgrad_defer(kind, L.ptr(x), L.ptr(dy), L.ptr(grad_w), L.ptr(grad_b), b, h, w, cx, c_real, cw, c_off, n,
                                             L.ptr(slab), nbytes, C.byref(job), L.stream()), "conv_wgrad_defer")
        _wgrad_pending.append((job, slab))
        return grad_w
    ws = _ws(x.device, nbytes, "wgrad")
    L.check(lib.ddk_conv_wgrad_bias(kind, L.ptr(_f32(x)), L.ptr(_f32(dy)), L.ptr(grad_w), L.ptr(grad_b), b, h, w, cx, c_real, cw, c_off, n,
                                    L.ptr(ws), nbytes, L.stream()), "conv_wgrad")
    return grad_w


def bias_grad(dy, accumulate_into=None):
    n = dy.shape[-1]
    out = accumulate_into if accumulate_into is not None else torch.empty(n, device=dy.device, dtype=torch.float32)
    ws = _ws(dy.device, 256 * n * 4, "bias")
    L.check(L.load().ddk_bias_grad(L.ptr(_f32(dy)), L.ptr(out), dy.numel() // n, n, int(accumulate_into is not None), L.ptr(ws),
                                   256 * n * 4, L.stream()), "bias_grad")
    return out


def gn_train_resident(b, hw, c, groups=GN_GROUPS):
    """the training GroupNorm keeps a (sample, group) slab in registers (and so takes the slab forms) when this holds"""
    return L.load().ddk_groupnorm_train_workspace_bytes(b, hw, c, groups) == 0


def groupnorm_mish_train(x, gamma, beta, temb=None, addend=None, drop_p=0.0, seed=0, layer=0, groups=GN_GROUPS, eps=GN_EPS, slabs=None,
                         conv_bias=None):
    """slabs [S,B,H,W,C] (ops.conv(leave_slabs=True)): x = sum(slabs) + conv_bias is formed while loading and WRITTEN into `x`"""
    b, h, w, c = x.shape
    out = torch.empty_like(x)
    stride = temb.stride(0) if temb is not None else 0
    lib = L.load()
    if slabs is not None:
        L.check(lib.ddk_groupnorm_mish_train_fwd_slabs(L.ptr(slabs), slabs.shape[0], slabs.stride(0), L.ptr(conv_bias), L.ptr(x), L.ptr(gamma),
                                                       L.ptr(beta), temb.data_ptr() if temb is not None else None, stride, L.ptr(addend),
                                                       float(drop_p), seed, layer, L.ptr(out), b, h * w, c, groups, eps, L.stream()),
                "groupnorm_mish_train_fwd_slabs")
        return out
    nbytes = lib.ddk_groupnorm_train_workspace_bytes(b, h * w, c, groups)
    ws = _ws(x.device, nbytes, "gn_train") if nbytes else None
    L.check(lib.ddk_groupnorm_mish_train_fwd(L.ptr(_f32(x)), L.ptr(gamma), L.ptr(beta),
                                             temb.data_ptr() if temb is not None else None, stride, L.ptr(addend),
                                             float(drop_p), seed, layer, L.ptr(out), b, h * w, c, groups, eps, L.ptr(ws), nbytes,
                                             L.stream()),
            "groupnorm_mish_train_fwd")
    return out


def groupnorm_mish_bwd(x, gamma, beta, dy, drop_p=0.0, seed=0, layer=0, groups=GN_GROUPS, eps=GN_EPS, acc=None, dy_slabs=None):
    """-> dx, dtemb [B,C], sums [3][C] = (dgamma, dbeta, sum over all pixels of dx -- the bias gradient of the conv that
    produced x).  `acc`: optional (gamma.grad, beta.grad, conv_bias.grad) accumulated into directly (entries may be None).
    dy_slabs [S,B,H,W,C]: dy = sum(dy_slabs), formed while loading (`dy` itself is not read)."""
    b, h, w, c = x.shape
    dx = torch.empty_like(x)
    part = torch.empty((4, b, c), device=x.device, dtype=torch.float32)
    lib = L.load()
    if dy_slabs is not None:
        L.check(lib.ddk_groupnorm_mish_bwd_slabs(L.ptr(_f32(x)), L.ptr(gamma), L.ptr(beta), float(drop_p), seed, layer, L.ptr(dy_slabs),
                                                 dy_slabs.shape[0], dy_slabs.stride(0), L.ptr(dx), L.ptr(part), b, h * w, c, groups, eps,
                                                 L.stream()), "groupnorm_mish_bwd_slabs")
    else:
        nbytes = lib.ddk_groupnorm_train_workspace_bytes(b, h * w, c, groups)
        ws = _ws(x.device, nbytes, "gn_train") if nbytes else None
        L.check(lib.ddk_groupnorm_mish_bwd(L.ptr(_f32(x)), L.ptr(gamma), L.ptr(beta), float(drop_p), seed, layer, L.ptr(_f32(dy)),
                                           L.ptr(dx), L.ptr(part), b, h * w, c, groups, eps, L.ptr(ws), nbytes, L.stream()),
                "groupnorm_mish_bwd")
    sums = [None, None, None]
    missing = [k for k in range(3) if acc is None or acc[k] is None]
    if acc is not None and any(t is not None for t in acc):
        # straight into the parameters' gradients (fixed-order row sums, then +=), one launch for the three of them
        _rows_sum_targets(part[1], 3, b * c, b, c, [acc[0], acc[1], acc[2]], c)
    if missing:
        out = torch.empty((3, c), device=x.device, dtype=torch.float32)
        L.check(lib.ddk_rows_sum_batched(L.ptr(part[1]), 3, b * c, b, c, L.ptr(out), c, 0, L.stream()), "rows_sum_batched")
        for k in missing:
            sums[k] = out[k]
    return dx, part[0], sums


# ---------------------------------------------------------------- widths that are not multiples of 32 (training)
def groupnorm_mish_generic_train(x, gamma, beta, temb=None, addend=None, drop_p=0.0, seed=0, layer=0, groups=GN_GROUPS, eps=GN_EPS):
    """x [B,H,W,CP] with zero padding behind the C = gamma.numel() real channels -> Dropout(Mish(GN(x)) + temb[b]) + addend, padding zero"""
    b, h, w, cp = x.shape
    c = gamma.numel()
    out = torch.empty_like(x)
    stride = temb.stride(0) if temb is not None else 0
    L.check(L.load().ddk_groupnorm_mish_generic_train_fwd(L.ptr(_f32(x)), L.ptr(gamma), L.ptr(beta), temb.data_ptr() if temb is not None else None,
                                                          stride, L.ptr(addend), float(drop_p), seed, layer, L.ptr(out), b, h * w, cp, c, groups,
                                                          eps, L.stream()), "groupnorm_mish_generic_train_fwd")
    return out


def groupnorm_mish_generic_bwd(x, gamma, beta, dy, drop_p=0.0, seed=0, layer=0, groups=GN_GROUPS, eps=GN_EPS):
    """-> dx [B,H,W,CP], dtemb [B,C], (dgamma, dbeta, sum of dx over batch and pixels) [3][C]"""
    b, h, w, cp = x.shape
    c = gamma.numel()
    dx = torch.empty_like(x)
    part = torch.empty((4, b, c), device=x.device, dtype=torch.float32)
    lib = L.load()
    L.check(lib.ddk_groupnorm_mish_generic_bwd(L.ptr(_f32(x)), L.ptr(gamma), L.ptr(beta), float(drop_p), seed, layer, L.ptr(_f32(dy)), L.ptr(dx),
                                               L.ptr(part), b, h * w, cp, c, groups, eps, L.stream()), "groupnorm_mish_generic_bwd")
    sums = torch.empty((3, c), device=x.device, dtype=torch.float32)
    L.check(lib.ddk_rows_sum_batched(L.ptr(part[1]), 3, b * c, b, c, L.ptr(sums), c, 0, L.stream()), "rows_sum_batched")
    return dx, part[0], sums


def chan_layernorm_generic(x, g, b, eps=LN_EPS):
    """LayerNorm over the C = g.numel() real channels of CP-pitched rows; padding written as zero"""
    cp, c = x.shape[-1], g.numel()
    out = torch.empty_like(x)
    L.check(L.load().ddk_chan_layernorm_generic(L.ptr(_f32(x)), L.ptr(g.reshape(-1)), L.ptr(b.reshape(-1)), L.ptr(out), x.numel() // cp, cp, c,
                                                eps, L.stream()), "chan_layernorm_generic")
    return out


def chan_layernorm_generic_bwd(x, g, dy, eps=LN_EPS, addend=None):
    """-> dx (+ addend), dg [C], db [C]"""
    cp, c = x.shape[-1], g.numel()
    m = x.numel() // cp
    dx = torch.empty_like(x)
    max_parts = 512
    part = torch.empty((2, max_parts, c), device=x.device, dtype=torch.float32)
    n = C.c_int(0)
    lib = L.load()
    L.check(lib.ddk_chan_layernorm_generic_bwd(L.ptr(_f32(x)), L.ptr(g.reshape(-1)), L.ptr(_f32(dy)), L.ptr(addend), L.ptr(dx), L.ptr(part),
                                               max_parts, C.byref(n), m, cp, c, eps, L.stream()), "chan_layernorm_generic_bwd")
    # the kernel wrote rows [0, n) of dg and rows [n, 2n) of db, contiguously
    flat = part.view(-1)
    dg = rows_sum(flat[:n.value * c], n.value, c, c)
    db = rows_sum(flat[n.value * c:2 * n.value * c], n.value, c, c)
    return dx, dg, db


def rows_sum(rows, nrows, row_stride, n, out=None):
    if out is None:
        out = torch.empty(n, device=rows.device, dtype=torch.float32)
    L.check(L.load().ddk_rows_sum(rows.data_ptr(), nrows, row_stride, out.data_ptr(), n, 0, L.stream()), "rows_sum")
    return out


_multi_add_tables = {}


def multi_add_(src, segments):
    """ddk_multi_add: dst_k += src[off_k : off_k + n_k] for segments [(offset, dst tensor), ...] in ONE launch.  The device table is
    cached per (offsets, destination addresses): the destinations are views of the optimiser's flat gradient buffer, stable for its
    lifetime."""
    key = (str(src.device),) + tuple((int(o), d.data_ptr(), d.numel()) for o, d in segments)
    tab = _multi_add_tables.get(key)
    if tab is None:
        if len(_multi_add_tables) > 64:
            _multi_add_tables.clear()
        host = torch.tensor([[int(o), d.data_ptr(), d.numel()] for o, d in segments], dtype=torch.int64)
        tab = _multi_add_tables[key] = host.to(src.device)
    if torch.cuda.is_current_stream_capturing():
        _graph_keep[-1].append((tab, [d for _, d in segments]))
    L.check(L.load().ddk_multi_add(L.ptr(src), tab.data_ptr(), len(segments), max(d.numel() for _, d in segments), L.stream()), "multi_add")


def chan_layernorm_bwd(x, g, dy, eps=LN_EPS, acc=None, addend=None):
    """acc = (g.grad slot, b.grad slot): the row sums are ADDED into them (one launch) and (dx, None, None) is returned;
    addend: added to dx by the same launch"""
    c = x.shape[-1]
    m = x.numel() // c
    dx = torch.empty_like(x)
    max_parts = 512
    part = torch.empty((2, max_parts, c), device=x.device, dtype=torch.float32)
    n = C.c_int(0)
    lib = L.load()
    L.check(lib.ddk_chan_layernorm_bwd_add(L.ptr(_f32(x)), L.ptr(g.reshape(-1)), L.ptr(_f32(dy)), L.ptr(addend), L.ptr(dx), L.ptr(part),
                                           max_parts, C.byref(n), m, c, eps, L.stream()), "chan_layernorm_bwd")
    # the kernel laid the rows out as [2][nparts][C] with nparts = n.value
    if acc is not None and acc[0] is not None and acc[1] is not None:
        _rows_sum_targets(part, 2, n.value * c, n.value, c, [acc[0], acc[1]], c)
        return dx, None, None
    out = torch.empty((2, c), device=x.device, dtype=torch.float32)
    L.check(lib.ddk_rows_sum_batched(L.ptr(part), 2, n.value * c, n.value, c, L.ptr(out), c, 0, L.stream()), "rows_sum_batched")
    return dx, out[0], out[1]


def linattn_small_from_x(x, w_qkv, ln_g, ln_b, heads=4, eps=LN_EPS):
    """PreNorm(LinearAttention) up to (not including) to_out on a small map (H*W <= 64) in ONE launch: x [B,H,W,C]; w_qkv the
    canonical to_qkv weight [3*heads*32, C(,1,1)]; ln_g / ln_b the channel LayerNorm's g, b (blocks.py:57-60, 123-134).
    -> (out [B,H,W,heads*32], ctx [B,heads,32,32]).  The folded weights are derived here (the UNet plan caches them)."""
    b, h, w, c = x.shape
    wq = w_qkv.reshape(w_qkv.shape[0], -1).to(torch.float32)
    g, bb = ln_g.reshape(-1).to(torch.float32), ln_b.reshape(-1).to(torch.float32)
    cp = pad32(c)
    lnw = torch.zeros((wq.shape[0], cp), device=x.device, dtype=torch.float32)
    lnw[:, :c] = wq * g[None, :]
    c1 = (wq * g[None, :]).sum(dim=1).contiguous()
    c2 = (wq * bb[None, :]).sum(dim=1).contiguous()
    wop = torch.empty_like(lnw)
    lib = L.load()
    L.check(lib.ddk_pack_qkv_operand(L.ptr(lnw), L.ptr(wop), heads, cp, L.stream()), "pack_qkv_operand")
    ctx = torch.empty((b, heads, 32, 32), device=x.device, dtype=torch.float32)
    out = torch.empty((b, h, w, heads * 32), device=x.device, dtype=torch.float32)
    L.check(lib.ddk_linattn_small_from_x(L.ptr(_f32(x)), L.ptr(wop), L.ptr(c1), L.ptr(c2), eps, L.ptr(ctx), L.ptr(out), b, h * w, c, heads,
                                         L.stream()), "linattn_small_from_x")
    return out, ctx


def linattn_train(qkv, heads=4):
    out, ctx = linattn(qkv, heads)
    b, h, w, _ = qkv.shape
    stats = torch.empty((b, heads, 2, 32), device=qkv.device, dtype=torch.float32)
    lib = L.load()
    nbytes = lib.ddk_linattn_train_workspace_bytes(b, h * w, heads)
    ws = _ws(qkv.device, nbytes, "linattn_train") if nbytes else None
    L.check(lib.ddk_linattn_stats(L.ptr(qkv), L.ptr(stats), b, h * w, heads, L.ptr(ws), nbytes, L.stream()), "linattn_stats")
    return out, ctx, stats


def linattn_bwd(qkv, dout, ctx, stats, heads=4):
    """stats None: the softmax statistics of k are recomputed from qkv by the backward's own launches (ddk_linattn_bwd_recompute)"""
    b, h, w, _ = qkv.shape
    dctx = torch.empty_like(ctx)
    dqkv = torch.empty_like(qkv)
    lib = L.load()
    nbytes = lib.ddk_linattn_train_workspace_bytes(b, h * w, heads)
    ws = _ws(qkv.device, nbytes, "linattn_train") if nbytes else None
    if stats is None:
        scratch = torch.empty((b, heads, 2, 32), device=qkv.device, dtype=torch.float32)
        L.check(lib.ddk_linattn_bwd_recompute(L.ptr(qkv), L.ptr(_f32(dout)), L.ptr(ctx), L.ptr(scratch), L.ptr(dctx), L.ptr(dqkv), b, h * w,
                                              heads, L.ptr(ws), nbytes, L.stream()), "linattn_bwd_recompute")
        return dqkv
    L.check(lib.ddk_linattn_bwd(L.ptr(qkv), L.ptr(_f32(dout)), L.ptr(ctx), L.ptr(stats), L.ptr(dctx), L.ptr(dqkv), b, h * w,
                                heads, L.ptr(ws), nbytes, L.stream()), "linattn_bwd")
    return dqkv


def ae_objective(l_ddpm, l_rec, t, t_rec_max):
    """-> [3] = (objective, latent, recon) of the dDDPM autoencoder's 'simple' loss from the per-sample losses (ddk_ae_objective)"""
    out = torch.empty(3, device=l_ddpm.device, dtype=torch.float32)
    L.check(L.load().ddk_ae_objective(L.ptr(_f32(l_ddpm)), L.ptr(_f32(l_rec)), t.data_ptr(), int(t_rec_max), l_ddpm.numel(), L.ptr(out),
                                      L.stream()), "ae_objective")
    return out


def ae_objective_bwd(g, t, t_rec_max):
    d1 = torch.empty(t.numel(), device=g.device, dtype=torch.float32)
    d2 = torch.empty_like(d1)
    L.check(L.load().ddk_ae_objective_bwd(L.ptr(_f32(g)), t.data_ptr(), int(t_rec_max), t.numel(), L.ptr(d1), L.ptr(d2), L.stream()),
            "ae_objective_bwd")
    return d1, d2


def mish_bwd(x, dy):
    dx = torch.empty_like(x)
    L.check(L.load().ddk_mish_bwd(L.ptr(_f32(x)), L.ptr(_f32(dy)), L.ptr(dx), x.numel(), L.stream()), "mish_bwd")
    return dx


def tanh_bwd(y, dy):
    dx = torch.empty_like(y)
    L.check(L.load().ddk_tanh_bwd(L.ptr(_f32(y)), L.ptr(_f32(dy)), L.ptr(dx), y.numel(), L.stream()), "tanh_bwd")
    return dx


def avgpool2_bwd(dy):
    b, h2, w2, c = dy.shape
    dx = torch.empty((b, 2 * h2, 2 * w2, c), device=dy.device, dtype=torch.float32)
    L.check(L.load().ddk_avgpool2_bwd(L.ptr(_f32(dy)), L.ptr(dx), b, 2 * h2, 2 * w2, c, L.stream()), "avgpool2_bwd")
    return dx


def upsample_nearest2_bwd(dy):
    b, h2, w2, c = dy.shape
    dx = torch.empty((b, h2 // 2, w2 // 2, c), device=dy.device, dtype=torch.float32)
    L.check(L.load().ddk_upsample_nearest2_bwd(L.ptr(_f32(dy)), L.ptr(dx), b, h2 // 2, w2 // 2, c, L.stream()), "upsample_nearest2_bwd")
    return dx


def sq_err_grad(a, b, scale):
    out = torch.empty_like(b)
    bsz = a.shape[0]
    L.check(L.load().ddk_sq_err_grad(L.ptr(_f32(a)), L.ptr(_f32(b)), L.ptr(_f32(scale)), L.ptr(out), bsz, a.numel() // bsz, L.stream()),
            "sq_err_grad")
    return out


def scale_per_sample(x, scale):
    out = torch.empty_like(x)
    bsz = x.shape[0]
    L.check(L.load().ddk_scale_per_sample(L.ptr(_f32(x)), L.ptr(_f32(scale)), L.ptr(out), bsz, x.numel() // bsz, L.stream()),
            "scale_per_sample")
    return out


def conv1x1_small_n_bwd(a, w, dy, acc=None):
    """-> da, dw [n_out, C], db [n_out].  acc = (weight.grad slot, bias.grad slot): the partial rows are summed straight INTO them
    (two row-sum jobs, deferred with the others inside deferred_wgrad()) and (da, None, None) is returned"""
    c = a.shape[-1]
    n_out = w.shape[0]
    m = a.numel() // c
    da = torch.empty_like(a)
    max_rows = 1024
    rowlen = n_out * c + n_out
    part = torch.empty((max_rows, rowlen), device=a.device, dtype=torch.float32)
    n = C.c_int(0)
    L.check(L.load().ddk_conv1x1_small_n_bwd(L.ptr(_f32(a)), L.ptr(w.reshape(n_out, c).contiguous()), L.ptr(_f32(dy)), L.ptr(da),
                                             L.ptr(part), max_rows, C.byref(n), m, c, n_out, L.stream()), "conv1x1_small_n_bwd")
    if acc is not None and acc[0] is not None and acc[1] is not None:
        flat = part.view(-1)
        _rows_sum_targets(flat, 1, 0, n.value, rowlen, [acc[0]], n_out * c)
        _rows_sum_targets(flat[n_out * c:], 1, 0, n.value, rowlen, [acc[1]], n_out)
        return da, None, None
    tot = rows_sum(part, n.value, rowlen, rowlen)
    return da, tot[:n_out * c].reshape(n_out, c), tot[n_out * c:]


def small_gemm(mode, a, b, out, m, n, k, lda, ldb, ldc, accumulate=False):
    lib = L.load()
    nbytes = lib.ddk_small_gemm_workspace_bytes(m, n, k, ldc)
    ws = _ws(out.device, nbytes, "small_gemm") if nbytes else None
    L.check(lib.ddk_small_gemm(mode, a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, lda, ldb, ldc, int(accumulate), L.ptr(ws), nbytes,
                               L.stream()), "small_gemm")
    return out


def sincos_embed(t, freqs):
    bsz, dim = t.shape[0], freqs.numel() * 2
    e = torch.empty((bsz, dim), device=t.device, dtype=torch.float32)
    L.check(L.load().ddk_sincos_embed(L.ptr(t), L.ptr(freqs), L.ptr(e), bsz, dim, L.stream()), "sincos_embed")
    return e


def bias_act_(y, bias, want_act):
    act = torch.empty_like(y) if want_act else None
    L.check(L.load().ddk_bias_act(L.ptr(y), L.ptr(bias), L.ptr(act), y.shape[0], y.shape[1], L.stream()), "bias_act")
    return act


# ------------------------------------------------------------------ optimiser (flat buffers)
def grad_norm_clip(flat_grad, max_norm):
    """-> device tensor [norm, clip_coef] (no host sync)"""
    out = torch.empty(2, device=flat_grad.device, dtype=torch.float32)
    ws = _ws(flat_grad.device, 4096, "norm")
    L.check(L.load().ddk_grad_norm_clip(L.ptr(flat_grad), flat_grad.numel(), float(max_norm), L.ptr(out), L.ptr(ws), 4096, L.stream()),
            "grad_norm_clip")
    return out


def adam_step_(p, g, m, v, lr, step, clip=None, betas=(0.9, 0.999), eps=1e-8):
    L.check(L.load().ddk_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                                   float(eps), int(step), L.ptr(clip), L.stream()), "adam_step")


def ema_update_(p_ema, p, decay):
    L.check(L.load().ddk_ema_update(L.ptr(p_ema), L.ptr(p), p.numel(), float(decay), L.stream()), "ema_update")
