"""The level-chain kernels (csrc/level_chain.hip) op by op: each op kind of the 4x4 and the 8x8 kernel, alone and in short chains with
hand-offs, against plain torch formulations of the reference modules (models/unet/blocks.py:74-84 Block, :105-115 ResnetBlock,
:57-71, 116-134 PreNorm(LinearAttention) + to_out) in fp64.  The whole-UNet / sampler tests of the chains are in test_step_edges_gpu.py."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err
from ddk import lib as L, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"

CONV3, CONV1, ATTN = 0, 1, 2
WAIT, SIGNAL, ADD_KEEP, SAVE_KEEP, ADD_KEEP2, SAVE_KEEP2, KEEP_FROM_SRC, NO_OUT = 1, 2, 4, 8, 16, 32, 64, 128


class ChainOp(C.Structure):
    _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("out", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int), ("kind", C.c_int), ("flags", C.c_int),
                ("temb_off", C.c_int), ("n_out", C.c_int)]


def _p(t):
    return None if t is None else t.data_ptr()


def run_chain(op_list, hw, B, temb=None):
    lib = L.load()
    lib.ddk_debug_level_chain.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.ddk_debug_level_chain.restype = C.c_int
    arr = (ChainOp * len(op_list))(*op_list)
    cnt = torch.zeros(64 * B + 16, device=DEV, dtype=torch.int32)
    L.check(lib.ddk_debug_level_chain(C.cast(arr, C.c_void_p), len(op_list), hw, B, _p(temb), temb.shape[1] if temb is not None else 0,
                                      cnt.data_ptr(), L.stream()), "debug_level_chain")
    torch.cuda.synchronize()
    c = cnt.cpu()
    assert int(c[64 * B]) == 0, "a wait timed out"
    assert int(c[:64 * B].abs().sum()) == 0, "the counters did not re-arm"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def block_ref(x, w, b, g, be, shift=None, groups=8):
    """Block (blocks.py:74-84) + the ResnetBlock's time shift, fp64, NCHW"""
    y = F.conv2d(x.double(), w.double(), b.double(), padding=w.shape[-1] // 2)
    if g is not None:
        y = F.group_norm(y, groups, g.double(), be.double(), eps=1e-5)
        y = y * torch.tanh(F.softplus(y))
    if shift is not None:
        y = y + shift.double()[:, :, None, None]
    return y


def pack3(w, hw):
    return ops.pack_conv_weight_wino_local(w) if hw == 64 else ops.pack_conv_weight_local(w)


def pack1(w):
    o, i = w.shape[0], w.shape[1]
    out = torch.empty(o * i, device=w.device, dtype=torch.float32)
    L.check(L.load().ddk_pack_conv1x1_weight_local(L.ptr(w.reshape(o, i).contiguous()), L.ptr(out), o, i, i, L.stream()), "pack1")
    return out


def gen(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("hw", [16, 64])
@pytest.mark.parametrize("cin", [256, 512])
def test_chain_conv3_single_op(hw, cin):
    """CH_CONV3 alone: conv3x3 + GroupNorm + Mish + time shift + residual from the source, 256 channels and a 256 + 256 concat"""
    B, S = 6, (4 if hw == 16 else 8)
    x = gen((B, cin, S, S), 1)
    w = gen((256, cin, 3, 3), 2, (cin * 9) ** -0.5)
    b, g, be = gen((256,), 3, 0.1), 1 + gen((256,), 4, 0.1), gen((256,), 5, 0.1)
    temb = gen((B, 300), 6)
    xa, xb = nhwc(x[:, :256]), (nhwc(x[:, 256:]) if cin == 512 else None)
    out = torch.empty((B, S, S, 256), device=DEV)
    wp = pack3(w, hw)
    flags = KEEP_FROM_SRC | ADD_KEEP
    run_chain([ChainOp(_p(xa), _p(xb), _p(wp), _p(b), _p(g), _p(be), _p(out), 256, cin - 256, CONV3, flags, 44, 256)], hw, B, temb)
    ref = block_ref(x, w, b, g, be, temb[:, 44:300]) + x[:, :256].double()
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref.cpu()) < 2e-5


@pytest.mark.parametrize("hw", [16, 64])
@pytest.mark.parametrize("cin", [128, 512])
def test_chain_conv1_single_op(hw, cin):
    """CH_CONV1 alone: 1x1 conv + bias (to_out: 128 -> 256; res_conv: 256 + 256 -> 256)"""
    B, S = 5, (4 if hw == 16 else 8)
    x = gen((B, cin, S, S), 11)
    w = gen((256, cin, 1, 1), 12, cin ** -0.5)
    b = gen((256,), 13, 0.1)
    xa = nhwc(x[:, :256]) if cin == 512 else nhwc(x)
    xb = nhwc(x[:, 256:]) if cin == 512 else None
    out = torch.empty((B, S, S, 256), device=DEV)
    wp = pack1(w)                       # (kept in a variable: a temporary would be freed, and reused, before the kernel reads it)
    run_chain([ChainOp(_p(xa), _p(xb), _p(wp), _p(b), None, None, _p(out), min(cin, 256), cin - min(cin, 256), CONV1, 0, -1, 256)], hw, B)
    ref = F.conv2d(x.double(), w.double(), b.double())
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref.cpu()) < 2e-5


def attn_ref(x, w_qkv, g, b, w_out, b_out, heads=4):
    """Residual(PreNorm(LinearAttention)) (blocks.py:8-14, 57-71, 116-134), fp64 NCHW"""
    x = x.double()
    B, Cc, H, W = x.shape
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    xn = (x - mean) / (var.sqrt() + 1e-5) * g.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    qkv = F.conv2d(xn, w_qkv.double())
    q, k, v = qkv.reshape(B, 3, heads, 32, H * W).unbind(1)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, heads * 32, H, W)
    return F.conv2d(out, w_out.double(), b_out.double()) + x, out


@pytest.mark.parametrize("B", [7, 32, 37])
@pytest.mark.parametrize("hw", [16, 64])
def test_chain_attention_block(hw, B):
    """a producer conv1x1 (hand-off) -> CH_ATTN (heads on workgroups 0..3, hand-off) -> to_out + the kept residual: the attention block
    of the level chains, with its two waits"""
    S = 4 if hw == 16 else 8
    x0 = gen((B, 256, S, S), 21)
    w0, b0 = gen((256, 256, 1, 1), 22, 256 ** -0.5), gen((256,), 23, 0.1)
    w_qkv = gen((384, 256, 1, 1), 24, 256 ** -0.5)
    g, b = 1 + gen((256,), 25, 0.1), gen((256,), 26, 0.1)
    w_out, b_out = gen((256, 128, 1, 1), 27, 128 ** -0.5), gen((256,), 28, 0.1)
    wq = w_qkv.reshape(384, 256)
    lnw = (wq * g[None, :]).contiguous()
    c1, c2 = lnw.sum(dim=1).contiguous(), (wq * b[None, :]).sum(dim=1).contiguous()
    wop = torch.empty_like(lnw)
    L.check(L.load().ddk_pack_qkv_operand(L.ptr(lnw), L.ptr(wop), 4, 256, L.stream()), "pack_qkv_operand")
    xin = nhwc(x0)
    x1 = torch.empty((B, S, S, 256), device=DEV)
    heads = torch.empty((B, S, S, 128), device=DEV)
    out = torch.empty((B, S, S, 256), device=DEV)
    wp0, wpo = pack1(w0), pack1(w_out)
    run_chain([ChainOp(_p(xin), None, _p(wp0), _p(b0), None, None, _p(x1), 256, 0, CONV1, SIGNAL | SAVE_KEEP, -1, 256),
               ChainOp(_p(x1), None, _p(wop), None, _p(c1), _p(c2), _p(heads), 256, 0, ATTN, WAIT | SIGNAL, -1, 128),
               ChainOp(_p(heads), None, _p(wpo), _p(b_out), None, None, _p(out), 128, 0, CONV1, WAIT | ADD_KEEP, -1, 256)], hw, B)
    x1_ref = F.conv2d(x0.double(), w0.double(), b0.double())
    ref, heads_ref = attn_ref(x1_ref, w_qkv, g, b, w_out, b_out)
    assert rel_err(x1.permute(0, 3, 1, 2).cpu(), x1_ref.cpu()) < 2e-5
    assert rel_err(heads.permute(0, 3, 1, 2).cpu(), heads_ref.cpu()) < 5e-5
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref.cpu()) < 5e-5


@pytest.mark.parametrize("hw", [16, 64])
@pytest.mark.parametrize("B", [3, 32, 37])
def test_chain_resnet_block_pair(hw, B):
    """two ResnetBlocks back to back (four CH_CONV3 ops, three hand-offs, the residual carried in registers), run twice on different
    data through the same buffers (a stale hand-off would show) -- batch 37: images walked in two rounds"""
    S = 4 if hw == 16 else 8
    ws = [gen((256, 256, 3, 3), 30 + i, (256 * 9) ** -0.5) for i in range(4)]
    bs = [gen((256,), 40 + i, 0.1) for i in range(4)]
    gs = [1 + gen((256,), 50 + i, 0.1) for i in range(4)]
    bes = [gen((256,), 60 + i, 0.1) for i in range(4)]
    temb = gen((B, 512), 70)
    wps = [pack3(w, hw) for w in ws]
    bufs = [torch.empty((B, S, S, 256), device=DEV) for _ in range(4)]
    outs = []
    for seed in (80, 81, 80):
        x = gen((B, 256, S, S), seed)
        xin = nhwc(x)
        for t in bufs:
            t.fill_(float("nan"))
        chain = [ChainOp(_p(xin), None, _p(wps[0]), _p(bs[0]), _p(gs[0]), _p(bes[0]), _p(bufs[0]), 256, 0, CONV3, SIGNAL | KEEP_FROM_SRC, 0, 256),
                 ChainOp(_p(bufs[0]), None, _p(wps[1]), _p(bs[1]), _p(gs[1]), _p(bes[1]), _p(bufs[1]), 256, 0, CONV3,
                         WAIT | SIGNAL | ADD_KEEP | SAVE_KEEP, -1, 256),
                 ChainOp(_p(bufs[1]), None, _p(wps[2]), _p(bs[2]), _p(gs[2]), _p(bes[2]), _p(bufs[2]), 256, 0, CONV3, WAIT | SIGNAL, 256, 256),
                 ChainOp(_p(bufs[2]), None, _p(wps[3]), _p(bs[3]), _p(gs[3]), _p(bes[3]), _p(bufs[3]), 256, 0, CONV3, WAIT | ADD_KEEP, -1, 256)]
        run_chain(chain, hw, B, temb)
        h = block_ref(x, ws[0], bs[0], gs[0], bes[0], temb[:, 0:256])
        r1 = block_ref(h, ws[1], bs[1], gs[1], bes[1]) + x.double()
        h = block_ref(r1, ws[2], bs[2], gs[2], bes[2], temb[:, 256:512])
        r2 = block_ref(h, ws[3], bs[3], gs[3], bes[3]) + r1
        assert rel_err(bufs[1].permute(0, 3, 1, 2).cpu(), r1.cpu()) < 3e-5
        assert rel_err(bufs[3].permute(0, 3, 1, 2).cpu(), r2.cpu()) < 5e-5
        outs.append(bufs[3].clone())
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])


DOWN, NO_GN = 512, 256
UPT = 3


@pytest.mark.parametrize("B", [4, 33])
def test_chain_downsample_and_transpose_conv_ops(B):
    """the level chain's edges: the Downsample conv (3x3, stride 2, padding 1: 8x8 -> 4x4, blocks.py:41-47) as its first op, handed to a
    conv3x3 + GroupNorm op with the residual kept from it, and the Upsample transpose conv (4x4, stride 2, padding 1: 4x4 -> 8x8,
    blocks.py:32-38) as its last -- against torch's conv2d / conv_transpose2d in fp64"""
    x = gen((B, 256, 8, 8), 91)
    wd, bd = gen((256, 256, 3, 3), 92, (256 * 9) ** -0.5), gen((256,), 93, 0.1)
    wc, bc, g, be = gen((256, 256, 3, 3), 94, (256 * 9) ** -0.5), gen((256,), 95, 0.1), 1 + gen((256,), 96, 0.1), gen((256,), 97, 0.1)
    wt, bt = gen((256, 256, 4, 4), 98, (256 * 4) ** -0.5), gen((256,), 99, 0.1)          # ConvTranspose2d weight: [in][out][4][4]
    wdp, wcp = ops.pack_conv_weight_local(wd), ops.pack_conv_weight_local(wc)
    wtp = torch.empty(256 * 16 * 256, device=DEV)
    L.check(L.load().ddk_pack_convT_weight_local(L.ptr(wt.contiguous()), L.ptr(wtp), 256, 256, L.stream()), "pack_convT_weight_local")
    xin = nhwc(x)
    d = torch.empty((B, 4, 4, 256), device=DEV)
    h = torch.empty((B, 4, 4, 256), device=DEV)
    up = torch.empty((B, 8, 8, 256), device=DEV)
    run_chain([ChainOp(_p(xin), None, _p(wdp), _p(bd), None, None, _p(d), 256, 0, CONV3, DOWN | NO_GN | SIGNAL | SAVE_KEEP, -1, 256),
               ChainOp(_p(d), None, _p(wcp), _p(bc), _p(g), _p(be), _p(h), 256, 0, CONV3, WAIT | SIGNAL | ADD_KEEP, -1, 256),
               ChainOp(_p(h), None, _p(wtp), _p(bt), None, None, _p(up), 256, 0, UPT, WAIT, -1, 256)], 16, B)
    d_ref = F.conv2d(x.double(), wd.double(), bd.double(), stride=2, padding=1)
    h_ref = block_ref(d_ref, wc, bc, g, be) + d_ref
    up_ref = F.conv_transpose2d(h_ref, wt.double(), bt.double(), stride=2, padding=1)
    assert rel_err(d.permute(0, 3, 1, 2).cpu(), d_ref.cpu()) < 2e-5
    assert rel_err(h.permute(0, 3, 1, 2).cpu(), h_ref.cpu()) < 3e-5
    assert rel_err(up.permute(0, 3, 1, 2).cpu(), up_ref.cpu()) < 3e-5
