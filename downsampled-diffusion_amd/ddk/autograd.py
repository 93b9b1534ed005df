"""torch.autograd.Function wrappers: HIP forward + HIP backward for every op of the training path.

torch's autograd engine only ORDERS the backward calls and accumulates ``.grad``; every tensor it passes around is
produced by a kernel of libddk.so (csrc/conv_igemm.hip, conv_wgrad.hip, backward.hip).  Activations are NHWC.
Counterpart of the graph torch records for reference models/unet/blocks.py + models/diffusion/ddpm.py:275-315 when
trainers/trainer_ddpm.py:124-128 calls ``objective.backward()``.
"""
import torch

from . import ops


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


def _grad_slot(p):
    """The tensor a parameter gradient can be accumulated into in place: ``p.grad`` when it already exists as a dense fp32
    buffer (trainers/optim.py gives every parameter a view of ONE flat gradient buffer).  The backward kernels then add
    into it directly and the Function returns None for that input -- no zero-fill, no temporary, no torch add."""
    g = getattr(p, "grad", None)
    if p is not None and p.is_leaf and g is not None and g.dtype == torch.float32 and g.is_contiguous():
        return g
    return None


class GradHandoff:
    """A gradient that would reach a tensor over a SECOND path -- the skip of a residual block -- handed from the Function that
    produces it to the Function whose input-gradient conv can add it in its epilogue (ddk_conv_args.resid), instead of being
    returned to autograd, which would sum the two paths with an elementwise add launch (12 of them on 17-67 MB tensors per cfg3
    micro-batch, ~30 small ones).  Order-safe: a consumer that runs first computes alone and says so; the producer then returns
    its gradient to autograd as before."""
    __slots__ = ("g", "g2", "consumer_done")

    def __init__(self):
        self.g = self.g2 = None
        self.consumer_done = False

    def give(self, g, g2=None):
        """producer side: True when the gradient was taken over (return None to autograd)"""
        if self.consumer_done:
            self.consumer_done = False
            return False
        self.g, self.g2 = g, g2
        return True

    def take(self):
        """consumer side: (g, g2) to add, or (None, None)"""
        if self.g is None and self.g2 is None:
            self.consumer_done = True
            return None, None
        g, g2 = self.g, self.g2
        self.g = self.g2 = None
        return g, g2


# The split-K slabs of a conv whose only consumer is a GroupNorm stay unreduced and the GroupNorm sums them while it loads (forward:
# ConvGNMishFn; backward: SlabLink).  The tests switch this off to compare with the separate reduce launches, bit for bit.
FOLD_SLABS = True
# The gradient of a UNet skip tensor from the up path travels to the Downsample conv's input-gradient launch (its other consumer) and
# is added in that epilogue instead of by an autograd add (same switch for the tests).
SKIP_HANDOFF = True


class SlabLink:
    """Joins the input-gradient conv of one Block to the GroupNorm backward of the Block before it when the tensor between them has
    exactly that one consumer (h of a ResnetBlock): the conv leaves its split-K partial slabs unreduced, the GroupNorm backward sums
    them while it loads dy -- one reduce launch less per ResnetBlock.  The tensor autograd carries between the two Functions is
    then only a placeholder (slab 0); the consumer checks it received that very tensor and fails loudly otherwise."""
    __slots__ = ("slabs", "placeholder_ptr")

    def __init__(self):
        self.slabs = None
        self.placeholder_ptr = 0

    def put(self, slabs):
        self.slabs, self.placeholder_ptr = slabs, slabs.data_ptr()
        return slabs[0]

    def take(self, dy):
        """the slabs that stand for dy, or None when dy is an ordinary tensor"""
        slabs, self.slabs = self.slabs, None
        if slabs is None:
            return None
        if dy.data_ptr() != self.placeholder_ptr or dy.shape != slabs.shape[1:]:
            raise RuntimeError("SlabLink: the gradient that reached the GroupNorm backward is not the placeholder its producer returned "
                               "(the tensor between the two Blocks has a second consumer?)")
        return slabs


def _conv_backward(kind, x, x2, weight, bias, dy, needs, gb_ready=False, dmish_src=None, dx_resid=None, dx2_resid=None, dx_link=None):
    """Shared backward of the conv family.  needs = (x, x2, weight, bias).  Returns (dx, dx2, gw, gb); gw / gb are None
    when they were accumulated straight into ``.grad``.  gb_ready: the bias gradient was already produced elsewhere
    (by the GroupNorm backward that follows the conv).  dmish_src: x is Mish(dmish_src) and dx is wanted with respect to
    dmish_src -- the input-gradient conv multiplies by Mish' in its epilogue (single-source convs)."""
    need_x, need_x2, need_w, need_b = needs
    c0 = x.shape[-1]
    c1 = 0 if x2 is None else x2.shape[-1]
    dx = dx2 = gw = gb = None
    if need_w:
        slot = _grad_slot(weight)
        gw_t = slot if slot is not None else torch.zeros_like(weight, memory_format=torch.contiguous_format)
        if kind == ops.CONVT4X4_S2:
            # dW[i][o][ky][kx] = sum X[i] * dY[o] shifted: wgrad of the 4x4 stride-2 conv with the roles swapped
            ops.conv_wgrad_(ops.CONV4X4_S2, dy, x, gw_t, c_real=dy.shape[-1], cw=dy.shape[-1], c_off=0, persistent=slot is not None)
        else:
            cin = weight.shape[1]
            gb_t = None
            keeps = slot is not None              # both targets are `.grad` slots: the reduce may wait for the end of the backward pass
            if bias is not None and need_b and not gb_ready:
                # the bias gradient (column sums of dy) rides on the weight-gradient launches of the first source
                bslot = _grad_slot(bias)
                gb_t = bslot if bslot is not None else torch.zeros_like(bias, memory_format=torch.contiguous_format)
                gb = None if bslot is not None else gb_t
                gb_ready = True
                keeps = keeps and bslot is not None
            ops.conv_wgrad_(kind, x, dy, gw_t, c_real=min(c0, cin), cw=cin, c_off=0, grad_b=gb_t, persistent=keeps)
            if x2 is not None:
                ops.conv_wgrad_(kind, x2, dy, gw_t, c_real=c1, cw=cin, c_off=c0, persistent=slot is not None)
        gw = None if slot is not None else gw_t
    if bias is not None and need_b and not gb_ready:
        slot = _grad_slot(bias)
        gb = ops.bias_grad(dy, accumulate_into=slot)
        if slot is not None:
            gb = None
    if need_x or need_x2:
        if kind == ops.CONVT4X4_S2:
            dx = ops.conv(ops.CONV4X4_S2, dy, ops.cached_pack("fwd", weight, ops.pack_conv_weight))   # (I,O,4,4) read as OIHW
        else:
            def wd(lo, hi):        # [c0+c1][taps][N], made only for a launch the Winograd kernel does not take
                return ops.cached_pack(("dgrad", c0 + c1), weight, lambda w: ops.pack_conv_weight_dgrad(w, i_pad=c0 + c1))[lo:hi]
            src = dy
            k = ops.CONV1X1 if kind == ops.CONV1X1 else ops.CONV3X3_S1
            if kind == ops.CONV3X3_S2:
                src = ops.zero_stuff2(dy, x.shape[1], x.shape[2])
            # 3x3 input gradients (also the zero-stuffed stride-2 one) run as Winograd F(2x2,3x3) where the shape allows
            wino = k == ops.CONV3X3_S1 and dmish_src is None
            if need_x:
                ww = ops.wino_weight(weight, src.shape, 0, min(c0, weight.shape[1]), dgrad=True) if wino and c0 <= weight.shape[1] else None
                if (FOLD_SLABS and dx_link is not None and dx_resid is None and dmish_src is None and x2 is None
                        and ops.gn_train_resident(x.shape[0], x.shape[1] * x.shape[2], c0)):
                    dx, slabs = ops.conv(k, src, None if ww is not None else wd(0, c0), n_out=c0, w_wino=ww, leave_slabs=True)
                    if slabs is not None:
                        dx = dx_link.put(slabs)
                else:
                    dx = ops.conv(k, src, None if ww is not None else wd(0, c0), n_out=c0, dmish_src=dmish_src, resid=dx_resid, w_wino=ww)
                dx_resid = None
            if need_x2:
                ww2 = ops.wino_weight(weight, src.shape, c0, c0 + c1, dgrad=True) if wino else None
                dx2 = ops.conv(k, src, None if ww2 is not None else wd(c0, c0 + c1), n_out=c1, resid=dx2_resid, w_wino=ww2)
                dx2_resid = None
    if dx_resid is not None:          # a handed-off gradient that no conv epilogue took (transpose conv, or no input gradient wanted)
        dx = dx_resid if dx is None else ops.add(dx, dx_resid)
    if dx2_resid is not None:
        dx2 = dx2_resid if dx2 is None else ops.add(dx2, dx2_resid)
    return dx, dx2, gw, gb


class ConvFn(torch.autograd.Function):
    """conv family on NHWC x (optionally channel-concatenated with x2), canonical (OIHW / (I,O,4,4)) weight."""

    @staticmethod
    def forward(ctx, kind, x, x2, weight, bias, resid, handoff=None, resid_handoff=None, take=None):
        ctx.handoff = handoff
        ctx.resid_handoff = resid_handoff     # GradHandoff that takes the residual's gradient (dy itself) to another Function's kernel
        ctx.take = take                       # GradHandoff whose gradient (x's second consumer) the input-gradient conv adds in its epilogue
        wu = None
        if kind == ops.CONV3X3_S1:
            wu = ops.wino_weight(weight, (x.shape[0], x.shape[1], x.shape[2], x.shape[3] + (0 if x2 is None else x2.shape[3])))
        if kind == ops.CONVT4X4_S2:
            wp = ops.cached_pack("fwdT", weight, ops.pack_convT_weight)
            n = weight.shape[1]
        else:
            wp = None if wu is not None else ops.cached_pack("fwd", weight, ops.pack_conv_weight)
            n = weight.shape[0]
        out = ops.conv(kind, x, wp, None if bias is None else bias.detach(), n_out=n, x2=x2, resid=resid, w_wino=wu)
        ctx.kind = kind
        ctx.save_for_backward(x, x2, weight, bias)
        ctx.has_resid = resid is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, x2, weight, bias = ctx.saved_tensors
        dy = _c(dy)
        extra = ctx.take.take()[0] if (ctx.take is not None and SKIP_HANDOFF) else None
        dx, dx2, gw, gb = _conv_backward(ctx.kind, x, x2, weight, bias, dy, ctx.needs_input_grad[1:5], dx_resid=extra)
        if ctx.handoff is not None and ctx.handoff.give(dx, dx2):       # the skip conv of a ResnetBlock: Block1's dgrad conv adds these
            dx = dx2 = None
        dres = dy if ctx.has_resid else None
        if dres is not None and ctx.resid_handoff is not None and ctx.resid_handoff.give(dres):
            dres = None
        return None, dx, dx2, gw, gb, dres, None, None, None


def conv(kind, x, weight, bias=None, x2=None, resid=None, handoff=None, resid_handoff=None, take=None):
    return ConvFn.apply(kind, x, x2, weight, bias, resid, handoff, resid_handoff, take)


class PreActConvFn(torch.autograd.Function):
    """out = conv(Mish(h)) (+ resid) as a function of the PRE-activation h, for chains conv -> Mish -> conv (the dDDPM encoder /
    decoder blocks, convblocks.py:112-130).  a = Mish(h) is handed in already materialised -- by the producing conv's epilogue
    (ddk_conv_args.mish_out), which this Function's forward also fills for ITS consumer when want_act -- and the backward's
    input-gradient conv multiplies by Mish'(h) in its epilogue (ddk_conv_args.dmish_src): no Mish forward / backward launches."""

    @staticmethod
    def forward(ctx, kind, h, a, weight, bias, resid, want_act, handoff=None):
        ctx.handoff = handoff                       # with resid: gives dy away; without: takes it into the dgrad epilogue
        wp = ops.cached_pack("fwd", weight, ops.pack_conv_weight)
        n = weight.shape[0]
        a_out = torch.empty((a.shape[0], a.shape[1], a.shape[2], n), device=a.device, dtype=torch.float32) if want_act else None
        out = ops.conv(kind, a, wp, None if bias is None else bias.detach(), n_out=n, resid=resid, mish_out=a_out)
        ctx.kind = kind
        ctx.save_for_backward(h, a, weight, bias)
        ctx.has_resid = resid is not None
        # the second output never carries a gradient: without this autograd fills a zero tensor of its shape for every backward call
        # (36 fills of 2-33 MB per cfg3 micro-batch)
        ctx.set_materialize_grads(False)
        if want_act:
            ctx.mark_non_differentiable(a_out)
            return out, a_out
        return out, out.new_empty(0)

    @staticmethod
    def backward(ctx, dy, _unused):
        h, a, weight, bias = ctx.saved_tensors
        if dy is None:
            return None, None, None, None, None, None, None, None
        dy = _c(dy)
        need = ctx.needs_input_grad
        skip = None
        if ctx.handoff is not None and not ctx.has_resid:
            skip, _ = ctx.handoff.take()
        dh, _, gw, gb = _conv_backward(ctx.kind, a, None, weight, bias, dy, (need[1], False, need[3], need[4]), dmish_src=h, dx_resid=skip)
        dres = dy if ctx.has_resid else None
        if dres is not None and ctx.handoff is not None and ctx.handoff.give(dres):
            dres = None
        return None, dh, None, gw, gb, dres, None, None


def preact_conv(kind, h, a, weight, bias=None, resid=None, want_act=True, handoff=None):
    """-> (out, Mish(out) or None); see PreActConvFn"""
    out, a_out = PreActConvFn.apply(kind, h, a, weight, bias, resid, want_act, handoff)
    return out, (a_out if want_act else None)


class GNMishFn(torch.autograd.Function):
    """y = dropout_p(mish(groupnorm(x)) + temb) + addend  (blocks.py:79-80,106-111)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, temb, addend, drop_p, seed, layer, groups, eps):
        ctx.save_for_backward(x, gamma, beta)
        ctx.cfg = (drop_p, seed, layer, groups, eps, temb is not None, addend is not None)
        return ops.groupnorm_mish_train(x, gamma.detach(), beta.detach(), temb=temb, addend=addend, drop_p=drop_p, seed=seed,
                                        layer=layer, groups=groups, eps=eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        drop_p, seed, layer, groups, eps, has_temb, has_add = ctx.cfg
        dy = _c(dy)
        acc = (_grad_slot(gamma), _grad_slot(beta), None)
        dx, dtemb, sums = ops.groupnorm_mish_bwd(x, gamma.detach(), beta.detach(), dy, drop_p, seed, layer, groups, eps, acc=acc)
        return dx, sums[0], sums[1], (dtemb if has_temb else None), (dy if has_add else None), None, None, None, None, None


class ConvGNMishFn(torch.autograd.Function):
    """One `Block` (blocks.py:74-84): y = dropout_p(mish(groupnorm(conv3x3(cat(x, x2)) + bias)) + temb) + addend.
    Fused so that the backward of the GroupNorm hands the conv its bias gradient (the per-channel sum of dx it already
    has in registers) -- no column-sum pass over dY."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, gamma, beta, temb, addend, drop_p, seed, layer, groups, eps, give=None, take=None, dy_link=None,
                dx_link=None, give2=None):
        ctx.give, ctx.take = give, take           # GradHandoff: `give` the addend's gradient away / `take` one into the dgrad epilogue
        ctx.give2 = give2                         # GradHandoff that carries x2's gradient (the UNet skip) to x2's other consumer's dgrad conv
        ctx.dy_link, ctx.dx_link = dy_link, dx_link     # SlabLink: this Block's dy arrives as slabs / its dx leaves as slabs
        wu = ops.wino_weight(weight, (x.shape[0], x.shape[1], x.shape[2], x.shape[3] + (0 if x2 is None else x2.shape[3])))
        # a conv that splits k leaves its partial slabs for the GroupNorm to sum while it loads (which also writes `raw`)
        fold = FOLD_SLABS and ops.gn_train_resident(x.shape[0], x.shape[1] * x.shape[2], weight.shape[0], groups)
        # (the im2col copy of the filter is made -- and refreshed every step -- only where the Winograd kernel does not take the shape)
        raw = ops.conv(ops.CONV3X3_S1, x, None if wu is not None else ops.cached_pack("fwd", weight, ops.pack_conv_weight), bias.detach(),
                       n_out=weight.shape[0], x2=x2, w_wino=wu, leave_slabs=fold)
        slabs = None
        if fold:
            raw, slabs = raw
        ctx.save_for_backward(x, x2, weight, bias, raw, gamma, beta)
        ctx.cfg = (drop_p, seed, layer, groups, eps, temb is not None, addend is not None)
        return ops.groupnorm_mish_train(raw, gamma.detach(), beta.detach(), temb=temb, addend=addend, drop_p=drop_p, seed=seed,
                                        layer=layer, groups=groups, eps=eps, slabs=slabs, conv_bias=bias.detach())

    @staticmethod
    def backward(ctx, dy):
        x, x2, weight, bias, raw, gamma, beta = ctx.saved_tensors
        drop_p, seed, layer, groups, eps, has_temb, has_add = ctx.cfg
        dy = _c(dy)
        need = ctx.needs_input_grad
        acc = (_grad_slot(gamma), _grad_slot(beta), _grad_slot(bias) if need[3] else None)
        dy_slabs = ctx.dy_link.take(dy) if ctx.dy_link is not None else None
        if dy_slabs is not None and has_add:
            raise RuntimeError("ConvGNMishFn: a Block with an addend cannot take its output gradient as slabs")
        draw, dtemb, sums = ops.groupnorm_mish_bwd(raw, gamma.detach(), beta.detach(), dy, drop_p, seed, layer, groups, eps, acc=acc,
                                                   dy_slabs=dy_slabs)
        r1 = r2 = None
        if ctx.take is not None:
            r1, r2 = ctx.take.take()
        dx, dx2, gw, _ = _conv_backward(ops.CONV3X3_S1, x, x2, weight, bias, draw, need[0:4], gb_ready=True, dx_resid=r1, dx2_resid=r2,
                                        dx_link=ctx.dx_link)
        gb = sums[2] if need[3] else None
        dadd = dy if has_add else None
        if dadd is not None and ctx.give is not None and ctx.give.give(dadd):
            dadd = None
        if dx2 is not None and ctx.give2 is not None and SKIP_HANDOFF and ctx.give2.give(dx2):
            dx2 = None
        return (dx, dx2, gw, gb, sums[0], sums[1], (dtemb if has_temb else None), dadd, None, None, None, None, None, None, None, None, None,
                None)


def conv_groupnorm_mish(x, weight, bias, gamma, beta, x2=None, temb=None, addend=None, drop_p=0.0, seed=0, layer=0, groups=8, eps=1e-5,
                        give=None, take=None, dy_link=None, dx_link=None, give2=None):
    return ConvGNMishFn.apply(x, x2, weight, bias, gamma, beta, temb, addend, float(drop_p), int(seed), int(layer), groups, eps, give, take,
                              dy_link, dx_link, give2)


def groupnorm_mish(x, gamma, beta, temb=None, addend=None, drop_p=0.0, seed=0, layer=0, groups=8, eps=1e-5):
    return GNMishFn.apply(x, gamma, beta, temb, addend, float(drop_p), int(seed), int(layer), groups, eps)


# ---------------------------------------------------------------- widths that are not multiples of 32
class PadParamFn(torch.autograd.Function):
    """A parameter zero-padded into the channel pitch the kernels want.  dims: for each of the (first two) axes a list of
    (dst_offset, src_offset, length) blocks -- one block for a plain axis, two for the input axis of a conv that reads a concat
    (each source keeps its own padded pitch).  Forward: zeros + block copies; backward: the blocks of the gradient copied back.
    Memory plumbing only (no arithmetic); the padded copy is what the conv / weight-gradient kernels see."""

    @staticmethod
    def forward(ctx, w, shape, dims):
        ctx.dims, ctx.wshape = dims, tuple(w.shape)
        wp = torch.zeros(shape, device=w.device, dtype=torch.float32)
        src = w.detach()
        if len(dims) == 1:
            for (d0, s0, n0) in dims[0]:
                wp[d0:d0 + n0] = src[s0:s0 + n0]
        else:
            for (d0, s0, n0) in dims[0]:
                for (d1, s1, n1) in dims[1]:
                    wp[d0:d0 + n0, d1:d1 + n1] = src[s0:s0 + n0, s1:s1 + n1]
        return wp

    @staticmethod
    def backward(ctx, g):
        dims = ctx.dims
        gw = torch.empty(ctx.wshape, device=g.device, dtype=torch.float32)
        if len(dims) == 1:
            for (d0, s0, n0) in dims[0]:
                gw[s0:s0 + n0] = g[d0:d0 + n0]
        else:
            for (d0, s0, n0) in dims[0]:
                for (d1, s1, n1) in dims[1]:
                    gw[s0:s0 + n0, s1:s1 + n1] = g[d0:d0 + n0, d1:d1 + n1]
        return gw, None, None


def pad_param(w, out_real=None, in_real=None):
    """w [O, I, ...] (or [O]) -> zero-padded copy with O -> pad32(O) and the input axis laid out as the padded segments of
    in_real = [c_a, c_b, ...] (default: one segment).  out_real / in_real None: that axis is left as it is."""
    o = w.shape[0]
    d0 = [(0, 0, o)]
    shape = [ops.pad32(o) if out_real is not None else o] + list(w.shape[1:])
    if w.dim() == 1:
        return PadParamFn.apply(w, tuple(shape), (d0,))
    d1, dst, src = [], 0, 0
    if in_real is None:
        d1 = [(0, 0, w.shape[1])]
        dst = w.shape[1]
    else:
        for n in in_real:
            d1.append((dst, src, n))
            dst += ops.pad32(n)
            src += n
        assert src == w.shape[1], (in_real, tuple(w.shape))
    shape[1] = dst
    return PadParamFn.apply(w, tuple(shape), (d0, d1))


class GNMishGenericFn(torch.autograd.Function):
    """GNMishFn on CP-pitched rows with C = gamma.numel() real channels (zero padding behind them): blocks.py:79-80,106-111"""

    @staticmethod
    def forward(ctx, x, gamma, beta, temb, addend, drop_p, seed, layer, groups, eps):
        ctx.save_for_backward(x, gamma, beta)
        ctx.cfg = (drop_p, seed, layer, groups, eps, temb is not None, addend is not None)
        return ops.groupnorm_mish_generic_train(x, gamma.detach(), beta.detach(), temb=temb, addend=addend, drop_p=drop_p, seed=seed,
                                                layer=layer, groups=groups, eps=eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        drop_p, seed, layer, groups, eps, has_temb, has_add = ctx.cfg
        dy = _c(dy)
        dx, dtemb, sums = ops.groupnorm_mish_generic_bwd(x, gamma.detach(), beta.detach(), dy, drop_p, seed, layer, groups, eps)
        return dx, sums[0], sums[1], (dtemb if has_temb else None), (dy if has_add else None), None, None, None, None, None


class ChanLayerNormGenericFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return ops.chan_layernorm_generic(x, g.detach(), b.detach(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dx, dg, db = ops.chan_layernorm_generic_bwd(x, g.detach(), _c(dy), ctx.eps)
        return dx, dg.reshape(g.shape), db.reshape(g.shape), None


class ChanLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps, take=None):
        ctx.save_for_backward(x, g, b)
        ctx.eps = eps
        ctx.take = take               # GradHandoff: the gradient that reaches x over the Residual around this PreNorm, added by the kernel
        return ops.chan_layernorm(x, g.detach(), b.detach(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, g, b = ctx.saved_tensors
        sg, sb = _grad_slot(g), _grad_slot(b)
        extra = ctx.take.take()[0] if ctx.take is not None else None
        dx, dg, db = ops.chan_layernorm_bwd(x, g.detach(), _c(dy), ctx.eps, acc=(sg, sb) if sg is not None and sb is not None else None,
                                            addend=extra)
        if dg is None:
            return dx, None, None, None, None
        return dx, dg.reshape(g.shape), db.reshape(g.shape), None, None


class LinAttnFn(torch.autograd.Function):
    """qkv [B,H,W,384] -> attention output [B,H,W,128] (blocks.py:128-133)."""

    @staticmethod
    def forward(ctx, qkv, heads):
        out, cx = ops.linattn(qkv, heads)           # (the softmax statistics of k are the backward's to recompute: no launch for them here)
        ctx.save_for_backward(qkv, cx)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, cx = ctx.saved_tensors
        return ops.linattn_bwd(qkv, _c(dout), cx, None, ctx.heads), None


class SmallNConvFn(torch.autograd.Function):
    """final 1x1 to in_channels (unet.py:71) and the resamplers' last 1x1."""

    @staticmethod
    def forward(ctx, a, weight, bias):
        ctx.save_for_backward(a, weight)
        ctx.bias_ref = bias                    # only to find its `.grad` slot in the backward
        return ops.conv1x1_small_n(a, weight.detach(), bias.detach())

    @staticmethod
    def backward(ctx, dy):
        a, weight = ctx.saved_tensors
        sw, sb = _grad_slot(weight), _grad_slot(ctx.bias_ref)
        da, dw, db = ops.conv1x1_small_n_bwd(a, weight.detach(), _c(dy), acc=(sw, sb) if sw is not None and sb is not None else None)
        if dw is None:
            return da, None, None
        return da, dw.reshape(weight.shape), db


class SqErrSumFn(torch.autograd.Function):
    """per-sample sum((a - b)^2); gradient flows to b (the model output)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.sq_err_sum(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gb = ops.sq_err_grad(a, b, _c(g)) if ctx.needs_input_grad[1] else None
        ga = None
        if ctx.needs_input_grad[0]:
            ga = ops.sq_err_grad(b, a, _c(g))
        return ga, gb


class AEObjectiveFn(torch.autograd.Function):
    """(objective, latent, recon) of DownsampleDDPMAutoencoder.losses for the 'simple' loss from the two per-sample losses [B]; the
    objective carries the gradient, the two report values do not (the trainers only log them)."""

    @staticmethod
    def forward(ctx, l_ddpm, l_rec, t, t_rec_max):
        out = ops.ae_objective(l_ddpm, l_rec, t, t_rec_max)
        ctx.save_for_backward(t)
        ctx.t_rec_max = t_rec_max
        obj, latent, recon = out[0], out[1], out[2]
        ctx.mark_non_differentiable(latent, recon)
        ctx.set_materialize_grads(False)
        return obj, latent, recon

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        (t,) = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        d1, d2 = ops.ae_objective_bwd(_c(g).reshape(1), t, ctx.t_rec_max)
        return d1, d2, None, None


class MishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.mish(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.mish_bwd(ctx.saved_tensors[0], _c(dy))


class TanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.tanh(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        return ops.tanh_bwd(ctx.saved_tensors[0], _c(dy))


class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.avgpool2(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool2_bwd(_c(dy))


class UpNearest2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample_nearest2(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample_nearest2_bwd(_c(dy))


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class NchwToNhwcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c_pad):
        ctx.c = x.shape[1]
        return ops.nchw_to_nhwc(x, c_pad)

    @staticmethod
    def backward(ctx, dy):
        return ops.nhwc_to_nchw(_c(dy), ctx.c), None


class NhwcToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        ctx.cs = x.shape[-1]
        return ops.nhwc_to_nchw(x, c)

    @staticmethod
    def backward(ctx, dy):
        return ops.nchw_to_nhwc(_c(dy), ctx.cs), None


class QSampleFn(torch.autograd.Function):
    """x_t = sqrt_acp[t] x + sqrt(1-acp)[t] eps; gradient to x only (needed by the non-autoencoder dDDPM loss)."""

    @staticmethod
    def forward(ctx, x, eps, t, sqrt_acp, sqrt_1m_acp):
        ctx.save_for_backward(t, sqrt_acp)
        return ops.q_sample(x, eps, t, sqrt_acp, sqrt_1m_acp)

    @staticmethod
    def backward(ctx, dy):
        t, sqrt_acp = ctx.saved_tensors
        return ops.scale_per_sample(_c(dy), sqrt_acp[t].contiguous()), None, None, None, None


class TimeEmbedFn(torch.autograd.Function):
    """t -> [B, sum C_out] time shifts of every ResnetBlock (blocks.py:22-29,92-95; unet.py:30-35).

    inputs: t, freqs, W1 [4d,d], b1, W2 [d,4d], b2, then the blocks' (weight [Co,d], bias [Co]) pairs.
    """

    @staticmethod
    def forward(ctx, t, freqs, w1, b1, w2, b2, *mlps):
        bsz, d = t.shape[0], w2.shape[0]
        dev = t.device
        e = ops.sincos_embed(t, freqs)
        u1 = torch.empty((bsz, 4 * d), device=dev, dtype=torch.float32)
        ops.small_gemm(1, e, w1.detach(), u1, bsz, 4 * d, d, d, d, 4 * d)
        h1 = ops.bias_act_(u1, b1.detach(), True)
        tv = torch.empty((bsz, d), device=dev, dtype=torch.float32)
        ops.small_gemm(1, h1, w2.detach(), tv, bsz, d, 4 * d, 4 * d, 4 * d, d)
        act = ops.bias_act_(tv, b2.detach(), True)
        ws, bs = mlps[0::2], mlps[1::2]
        ctot = sum(w.shape[0] for w in ws)
        # all per-block Linears as ONE product against the concatenated weight (3 launches instead of 17 + 1)
        wcat = torch.cat([w.detach() for w in ws])                  # [ctot, d]
        out = torch.empty((bsz, ctot), device=dev, dtype=torch.float32)
        ops.small_gemm(1, act, wcat, out, bsz, ctot, d, d, d, ctot)
        bcat = torch.cat([b.detach() for b in bs])
        ops.bias_act_(out, bcat, False)
        ctx.save_for_backward(e, u1, h1, tv, act, w1, w2, wcat)
        ctx.couts = [w.shape[0] for w in ws]
        ctx.params = (w1, b1, w2, b2) + tuple(mlps)          # the leaves themselves: their .grad slots are looked up in the backward
        # one output per block: column ranges (views) of the single [B, sum C_out] product.  Separate outputs so that the
        # backward gets one gradient per block -- slicing a single output would make autograd build a zero-filled
        # [B, sum C_out] tensor per block and add them up.
        views, off = [], 0
        for co in ctx.couts:
            views.append(out[:, off:off + co])
            off += co
        return tuple(views)

    @staticmethod
    def backward(ctx, *douts):
        saved = ctx.saved_tensors
        e, u1, h1, tv, act, w1, w2, wcat = saved
        bsz = act.shape[0]
        dout = torch.cat([d if d is not None else torch.zeros((bsz, co), device=act.device, dtype=torch.float32)
                          for d, co in zip(douts, ctx.couts)], dim=1)
        bsz, ctot = dout.shape
        d = w2.shape[0]
        dev = dout.device
        # all 2 x 17 + 4 parameter gradients side by side in ONE buffer; when every parameter has its place in the flat gradient bucket
        # they are added there by one launch (ddk_multi_add) instead of 38 autograd accumulation adds
        params = ctx.params
        slots = [_grad_slot(p) for p in params]
        n_w, n_b = wcat.numel(), ctot
        sizes = [n_w, n_b, w1.numel(), 4 * d, w2.numel(), d]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 3) // 4 * 4)
        G = torch.empty(offs[-1], device=dev, dtype=torch.float32)
        gwcat, gw1, gw2 = G[offs[0]:offs[0] + n_w].view(ctot, d), G[offs[2]:offs[2] + sizes[2]].view_as(w1), G[offs[4]:offs[4] + sizes[4]].view_as(w2)
        ops.small_gemm(2, dout, act, gwcat, ctot, d, bsz, ctot, d, d)        # dWcat[ctot][d] = dout^T act
        ops.rows_sum(dout, bsz, ctot, ctot, out=G[offs[1]:offs[1] + n_b])
        gbcat = G[offs[1]:offs[1] + n_b]
        dact = torch.empty((bsz, d), device=dev, dtype=torch.float32)
        ops.small_gemm(0, dout, wcat, dact, bsz, d, ctot, ctot, d, d)        # dact = dout Wcat
        grads = []
        off = 0
        for co in ctx.couts:                                                 # per-parameter gradients are row ranges (views)
            grads += [gwcat[off:off + co], gbcat[off:off + co]]
            off += co
        dtv = ops.mish_bwd(tv, dact)
        ops.small_gemm(2, dtv, h1, gw2, d, 4 * d, bsz, d, 4 * d, 4 * d)
        gb2 = ops.rows_sum(dtv, bsz, d, d, out=G[offs[5]:offs[5] + d])
        dh1 = torch.empty((bsz, 4 * d), device=dev, dtype=torch.float32)
        ops.small_gemm(0, dtv, w2.detach(), dh1, bsz, 4 * d, d, d, 4 * d, 4 * d)
        du1 = ops.mish_bwd(u1, dh1)
        ops.small_gemm(2, du1, e, gw1, 4 * d, d, bsz, 4 * d, d, d)
        gb1 = ops.rows_sum(du1, bsz, 4 * d, 4 * d, out=G[offs[3]:offs[3] + 4 * d])
        if all(sl is not None for sl in slots):
            segs = [(offs[2], slots[0]), (offs[3], slots[1]), (offs[4], slots[2]), (offs[5], slots[3])]
            off = 0
            for k, co in enumerate(ctx.couts):
                segs.append((offs[0] + off * d, slots[4 + 2 * k]))
                segs.append((offs[1] + off, slots[5 + 2 * k]))
                off += co
            ops.multi_add_(G, segs)
            return (None,) * (6 + len(grads))
        return (None, None, gw1, gb1, gw2, gb2, *grads)
