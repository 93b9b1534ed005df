// unet_plan.hip -- native sequencing of the eval-mode UNet forward and of the T-step sampler loop.
//
// Reference: models/unet/unet.py:10-72 (module tree), :74-104 (forward), models/unet/blocks.py:105-115
// (ResnetBlock), :8-14,63-71,126-134 (Residual(PreNorm(LinearAttention))), models/diffusion/ddpm.py:203-249
// (p_sample, p_sample_loop).
//
// The plan owns (a) the list of weight slots -- one per state_dict tensor, with the packed layout each
// kernel wants -- and (b) the order of kernel launches for one forward.  Nothing here computes: every
// arithmetic step is a HIP kernel from the other translation units.  The sampler captures one reverse step
// (t bookkeeping + UNet + x update) into a hipGraph and replays it, so the T-step loop costs one
// hipGraphLaunch per step on the host.
#include <mutex>
#include <string>
#include <vector>

#include "ddk_internal.h"
#include "level_chain.h"

namespace ddk {

constexpr int HEADS = 4;
constexpr int HIDDEN = HEADS * 32;  // LinearAttention hidden width is 128 whatever unet_chan is (blocks.py:119-122)
constexpr int GROUPS = 8;
constexpr float GN_EPS = 1e-5f, LN_EPS = 1e-5f;

static inline int pad32(int c) { return (c + 31) / 32 * 32; }

enum PackKind { PK_COPY = 0, PK_CONV = 1, PK_CONVT = 2, PK_LINEAR_T = 3, PK_WINO = 4, PK_LOCAL = 5, PK_WLOCAL = 6, PK_FIRST = 7, PK_CONVT_WINO = 8,
                PK_ROWS = 9 /* [O][I] rows into rows of pitch ld */, PK_LOCAL1 = 10 /* a 1x1 filter in conv_local.hip's operand order */,
                PK_CONVT_LOCAL = 11 /* a transpose-conv filter in the level chain's operand order */ };

struct Slot {
    std::string name;
    long long numel;
    int kind;
    size_t off;  // float offset into the packed arena
    int O, I, KH, KW, i_pad, ld, col0;
    int split = 0, split_pad = 0;   // PK_CONV over two separately padded sources (generic widths): see ddk_pack_conv_weight_split
    int attn = -1;  // >= 0: the slot feeds the folded LayerNorm of that attention site (packing it re-derives W o g, W g, W b)
};

struct ConvW {
    size_t w = 0, b = 0;
    bool has_bias = false;
    int cin = 0, cin_pad = 0, cout = 0;
    size_t wu = 0;          // Winograd-domain copy of a 3x3 filter ([cin_pad/32][16][cout][32]); has_wu says whether it exists
    bool has_wu = false;
    size_t wl = 0;          // conv_local.hip's operand-order copy (3x3 convs that feed a GroupNorm); has_wl says whether it exists
    bool has_wl = false;
    size_t wwl = 0;         // and its Winograd-domain form for the 8x8 maps (conv3x3_gn_wlocal_kernel); has_wwl
    bool has_wwl = false;
    size_t wf = 0;          // conv_first.hip's operand-order copy (the network's first conv, C_in <= 8); has_wf
    bool has_wf = false;
    size_t wl1 = 0;         // a 1x1 filter (to_out, res_conv) in conv_local.hip's operand order, one tap: the level chain's 1x1 ops
    bool has_wl1 = false;
    size_t wtl = 0;         // a transpose-conv filter in the level chain's operand order (CH_UPT); has_wtl
    bool has_wtl = false;
};
struct NormW { size_t g = 0, b = 0; int c_real = 0; };
struct ResW {
    // ci / co: channel counts as the kernels see them (co is padded to 32 at generic widths); *_real: the reference's
    int ci = 0, ci_pad = 0, co = 0, temb_off = 0, ci_real = 0, co_real = 0;
    ConvW c1, c2, res;
    NormW n1, n2;
    bool has_res = false;
};
struct AttnW {
    int c = 0, c_real = 0;
    NormW ln;
    ConvW qkv, out;
    // LayerNorm folded into to_qkv (derived at pack time by ddk_unet_finalize_pack): W o g, W g, W b
    size_t qkv_lnw = 0, ln_c1 = 0, ln_c2 = 0;
    size_t qkv_op = 0;      // qkv_lnw in the operand order of linattn_small_qkv_kernel (derived with it)
};

}  // namespace ddk

using namespace ddk;

// One captured reverse step.  A hipGraph bakes in every pointer its kernels were launched with, so an entry is valid for
// exactly this set of buffers and this shape; t, the Philox seed / stream id and the injected-noise step index are read
// from device memory by the kernels, so the same graph serves every step of every chain on those buffers.
struct SamplerGraph {
    const void *packed, *x, *noise, *ws, *c_recip, *c_recipm1, *c1, *c2, *sigma;
    int B, H, W, t_start, device;
    unsigned long long pack_epoch;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph_multi = nullptr;        // SAMPLER_MULTI consecutive steps in one graph (captured when a chain is long enough)
    hipGraphExec_t exec_multi = nullptr;
    hipEvent_t done = nullptr;               // recorded behind the entry's latest launches: what an eviction has to wait for
    unsigned long long last_use = 0;
};
constexpr int SAMPLER_MULTI = 16;

struct ddk_unet {
    // sampler state cached across ddk_sampler_run calls (guarded by `mu`; see ddk_sampler_invalidate)
    std::mutex mu;
    std::vector<SamplerGraph> graphs;
    unsigned long long use_clock = 0;
    unsigned long long pack_epoch = 0;       // bumped by every pack / finalize: weights changed
    struct { const void* ws = nullptr; int t_start = -1, B = 0, H = 0, W = 0; unsigned long long pack_epoch = 0; } table;
    ddk_unet_config cfg;
    int cluster_gn = 1;                      // GroupNorm finished inside the Winograd conv launch where eligible (ddk_unet_set_option):
                                             // 0 never, 1 in ddk_sampler_run (whose caller checks ddk_unet_cluster_check at the chain's
                                             // sync point), 2 in ddk_unet_forward too
    int cluster_limit = 1 << 30;             // diagnostic: only the first so many eligible launches of a forward take that path
    int cluster_np_max = 8;                  // largest cluster (workgroups per image and n tile) that takes the path
    bool cluster_split = true;               // ... also on shapes whose channel chunks are split over 2-4 workgroups (conv_wino_cluster_split_np):
                                             // a tile's first workgroup sums its partners' partial tiles in the launch (diagnostic option 11)
    bool attn_kvctx = true;                  // folded attention block: k, v projection + context in one launch (no kv tensor)
    bool fold_down_reduce = false;           // Downsample conv's split-K slabs summed by the image-local ResnetBlock behind it (no reduce launch).
                                             // OFF by default: measured 6 us per step SLOWER (each of an image's eight workgroups re-sums the slabs:
                                             // +6.3 / +5.9 us on the two consumers against reduce launches of 5.2 / 4.9 us; tools/fold_ab.py)
    bool first_gn = true;                    // the first Block's GroupNorm + Mish + shift inside conv_first_kernel's launch (DDK_OPT_FIRST_GROUPNORM),
                                             // wherever the in-launch GroupNorm of the Winograd convs may run
    int level_chain = 9;                     // bit 0: the whole 4x4 level (ResnetBlocks + attention of downs[-1], mid, ups[0]) as ONE persistent launch
                                             // (level_chain.hip) wherever the in-launch GroupNorm may run (its workgroups wait for each other too);
                                             // bit 3: ... with the Downsample conv in front of it and the Upsample transpose conv behind it inside
                                             // the same launch (8x8 -> 4x4 -> ... -> 4x4 -> 8x8: two igemm + slab-reduce pairs less);
                                             // bits 1, 2: the two 8x8 levels (downs[-2]; ups[1]) likewise, one launch each -- built, tested,
                                             // OFF by default: measured 100 us per step SLOWER (150 + 170 us against 104 + 125 for the 16
                                             // launches: a hop moves a 64 KB image to each of 8 workgroups, 5 us of staging and 7 us of
                                             // tail per op where the launches pay 2.9 + 2.4 + 2.0; profiles/r06_chain8_clock.txt, r06_chain8_ab.txt)
    int level_chain_max_batch = 32;          // the chain walks images in rounds of 32 (one workgroup per CU): measured at cfg4 it wins up to ONE round
                                             // (batch 16: 0.984 -> 0.947 ms, 32: 1.282 -> 1.232) and loses from the second on (48: 2.084 -> 2.142,
                                             // 64: 2.071 -> 2.095, 192: 5.753 -> 5.976; tools/chain_batch_ab.py) -- the launches it replaces amortise
                                             // their boundaries over the larger batch, the chain's hops do not (diagnostic option 10 moves the limit)
    int attn_fold_min_hw = 256;              // ... on maps with MORE pixels than this (diagnostic option 9: 255 lets the 16x16, C = 128 site take it)
    bool attn_fold = true;                   // attention on maps with HW > 256, C = 128: q projection + apply + to_out as ONE per-image C x C conv
    // unet_chan % 8 == 0 but not % 32 (reference blocks.py:75 takes any GroupNorm(8, C)): every tensor keeps a pitch of pad32(C) channels
    // with zero padding, the convs run on the generic im2col kernels over zero-padded weights and the normalisations on their
    // generic forms (norm_act.hip) -- correct, untuned, none of the fused paths
    bool generic = false;
    std::vector<int> dimp;                   // dims padded to 32 (== dims unless generic)
    std::vector<Slot> slots;
    size_t packed_floats = 0;
    int L = 0;
    std::vector<int> dims;  // [in, chan*m0, ...]
    int time_dim = 0, temb_total = 0;
    size_t freqs = 0, w1t = 0, b1 = 0, w2t = 0, b2 = 0, temb_wt = 0, temb_bias = 0;
    std::vector<ResW> down_res, up_res;  // 2 per level
    std::vector<AttnW> down_attn, up_attn;
    std::vector<ConvW> down_conv, up_conv;
    ResW mid1, mid2;
    AttnW mid_attn;
    ConvW final_conv;
    NormW final_norm;
    size_t final_w = 0, final_b = 0;

    size_t alloc(size_t n) {
        const size_t o = packed_floats;
        packed_floats += (n + 3) / 4 * 4;
        return o;
    }
    size_t add_copy(const std::string& name, long long n) {
        const size_t o = alloc((size_t)n);
        slots.push_back(Slot{name, n, PK_COPY, o, 0, 0, 0, 0, 0, 0, 0});
        return o;
    }
    void add_copy_at(const std::string& name, long long n, size_t off) {
        slots.push_back(Slot{name, n, PK_COPY, off, 0, 0, 0, 0, 0, 0, 0});
    }
    // split > 0: the cin input channels are the concat of two sources of `split` and cin - split channels, each padded to 32 on its own
    ConvW add_conv(const std::string& prefix, int cout, int cin, int k, bool bias, bool gn_follows = false, int split = 0, bool local = false) {
        ConvW c;
        c.cin = cin; c.cin_pad = split > 0 ? pad32(split) + pad32(cin - split) : pad32(cin); c.cout = cout;
        const int cout_rows = pad32(cout);          // rows beyond cout stay zero (the arena is zero-filled): N as the kernels see it
        c.w = alloc((size_t)cout_rows * k * k * c.cin_pad);
        Slot sl{prefix + "weight", (long long)cout * cin * k * k, PK_CONV, c.w, cout, cin, k, k, c.cin_pad, 0, 0};
        if (split > 0 && c.cin_pad != cin) { sl.split = split; sl.split_pad = pad32(split); }
        slots.push_back(sl);
        c.has_bias = bias;
        if (bias) {
            c.b = alloc((size_t)cout_rows);
            slots.push_back(Slot{prefix + "bias", cout, PK_COPY, c.b, 0, 0, 0, 0, 0, 0, 0});
        }
        if (generic) return c;                      // no Winograd / image-local / first-layer copies: the generic kernels only
        if (k == 1 && bias && cout % 32 == 0 && c.cin_pad == cin && cin % 32 == 0) {
            // to_out / res_conv: also in the operand order of the image-local kernels (level_chain.hip's 1x1 ops)
            c.wl1 = alloc((size_t)cout * c.cin_pad);
            c.has_wl1 = true;
            slots.push_back(Slot{prefix + "weight", (long long)cout * cin, PK_LOCAL1, c.wl1, cout, cin, 1, 1, c.cin_pad, 0, 0});
        }
        if (k == 3 && cout % 64 == 0) {
            // the same state_dict tensor feeds a second slot: its Winograd-domain form G g G^T for conv3x3_wino_kernel
            c.wu = alloc((size_t)16 * cout * c.cin_pad);
            c.has_wu = true;
            slots.push_back(Slot{prefix + "weight", (long long)cout * cin * k * k, PK_WINO, c.wu, cout, cin, k, k, c.cin_pad, 0, 0});
        }
        if (k == 3 && (gn_follows || local) && cout % 32 == 0) {
            // and a third for the one-launch conv + GroupNorm kernel of the 4x4 maps (its MFMA operand order; `local`: the Downsample conv
            // the level chain absorbs)
            c.wl = alloc((size_t)9 * cout * c.cin_pad);
            c.has_wl = true;
            slots.push_back(Slot{prefix + "weight", (long long)cout * cin * k * k, PK_LOCAL, c.wl, cout, cin, k, k, c.cin_pad, 0, 0});
            if (gn_follows && (c.cin_pad <= 320 || c.cin_pad == 512)) {     // (512: the 8x8 level chain walks the concat input as two staged halves)
                c.wwl = alloc((size_t)16 * cout * c.cin_pad);
                c.has_wwl = true;
                slots.push_back(Slot{prefix + "weight", (long long)cout * cin * k * k, PK_WLOCAL, c.wwl, cout, cin, k, k, c.cin_pad, 0, 0});
            }
        }
        return c;
    }
    ConvW add_convT(const std::string& prefix, int ch, bool local = false) {
        ConvW c;
        c.cin = ch; c.cin_pad = pad32(ch); c.cout = ch;
        c.w = alloc((size_t)16 * pad32(ch) * pad32(ch));
        slots.push_back(Slot{prefix + "weight", (long long)16 * ch * ch, PK_CONVT, c.w, ch, ch, 4, 4, pad32(ch), 0, 0});
        c.has_bias = true;
        c.b = alloc((size_t)pad32(ch));
        slots.push_back(Slot{prefix + "bias", ch, PK_COPY, c.b, 0, 0, 0, 0, 0, 0, 0});
        if (local && ch % 32 == 0 && !generic) {
            // ... in the level chain's operand order (four phases x 2 x 2 taps: level_chain.hip, CH_UPT)
            c.wtl = alloc((size_t)16 * ch * ch);
            c.has_wtl = true;
            slots.push_back(Slot{prefix + "weight", (long long)16 * ch * ch, PK_CONVT_LOCAL, c.wtl, ch, ch, 4, 4, ch, 0, 0});
        }
        if (ch % 128 == 0) {
            // the same tensor's Winograd F(2x2, 2x2) form, four phases x 9 positions (conv_winoT_kernel.inc)
            c.wu = alloc((size_t)36 * ch * ch);
            c.has_wu = true;
            slots.push_back(Slot{prefix + "weight", (long long)16 * ch * ch, PK_CONVT_WINO, c.wu, ch, ch, 4, 4, ch, 0, 0});
        }
        return c;
    }
    NormW add_norm(const std::string& wname, const std::string& bname, int c) {
        NormW n;
        n.c_real = c;
        n.g = alloc((size_t)pad32(c));              // padding stays zero
        slots.push_back(Slot{wname, c, PK_COPY, n.g, 0, 0, 0, 0, 0, 0, 0});
        n.b = alloc((size_t)pad32(c));
        slots.push_back(Slot{bname, c, PK_COPY, n.b, 0, 0, 0, 0, 0, 0, 0});
        return n;
    }
    // ci, co: the reference's channel counts; split > 0: the input is the concat (unet.py:97) of `split` + (ci - split) channels
    ResW add_res(const std::string& p, int ci, int co, int& temb_cursor, int split = 0) {
        ResW r;
        r.ci_real = ci; r.co_real = co;
        r.ci = ci; r.ci_pad = split > 0 ? pad32(split) + pad32(ci - split) : pad32(ci); r.co = pad32(co);
        if (!generic) r.co = co;
        r.temb_off = temb_cursor;
        temb_cursor += co;
        slots.push_back(Slot{p + "mlp.1.weight", (long long)co * time_dim, PK_LINEAR_T, temb_wt, co, time_dim, 0, 0, 0, temb_total, r.temb_off});
        add_copy_at(p + "mlp.1.bias", co, temb_bias + r.temb_off);
        r.c1 = add_conv(p + "block1.block.0.", co, ci, 3, true, true, split);
        r.n1 = add_norm(p + "block1.block.1.weight", p + "block1.block.1.bias", co);
        r.c2 = add_conv(p + "block2.block.0.", co, co, 3, true, true);
        r.n2 = add_norm(p + "block2.block.1.weight", p + "block2.block.1.bias", co);
        r.has_res = ci != co;
        if (r.has_res) r.res = add_conv(p + "res_conv.", co, ci, 1, true, false, split);
        return r;
    }
    int n_attn = 0;
    std::vector<AttnW> attn_all;   // copies in creation order (slot.attn indexes this)
    AttnW add_attn(const std::string& p, int c) {
        AttnW a;
        a.c = generic ? pad32(c) : c;
        a.c_real = c;
        const size_t s0 = slots.size();
        a.qkv = add_conv(p + "fn.fn.to_qkv.", 3 * HIDDEN, c, 1, false);
        const size_t s1 = slots.size();
        a.out = add_conv(p + "fn.fn.to_out.", c, HIDDEN, 1, true);
        const size_t s2 = slots.size();
        a.ln = add_norm(p + "fn.norm.g", p + "fn.norm.b", c);
        for (size_t i = s0; i < s1; ++i) slots[i].attn = n_attn;               // to_qkv weight
        for (size_t i = s2; i < slots.size(); ++i) slots[i].attn = n_attn;     // LayerNorm g, b
        a.qkv_lnw = alloc((size_t)3 * HIDDEN * pad32(c));
        a.ln_c1 = alloc((size_t)3 * HIDDEN);
        a.ln_c2 = alloc((size_t)3 * HIDDEN);
        a.qkv_op = alloc((size_t)3 * HIDDEN * pad32(c));
        attn_all.push_back(a);
        ++n_attn;
        return a;
    }
};

static int total_temb(const ddk_unet& u) {
    // 2 ResnetBlocks per down level, 2 mid, 2 per up level; C_out per unet.py:43-66
    int tot = 0;
    for (int l = 0; l < u.L; ++l) tot += 2 * u.dims[l + 1];
    tot += 2 * u.dims[u.L];
    for (int i = 0; i < u.L - 1; ++i) tot += 2 * u.dims[u.L - 1 - i];
    return tot;
}

extern "C" ddk_unet* ddk_unet_create(const ddk_unet_config* cfg) {
    if (!cfg || cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->in_ch < 1 || cfg->chan < 8 || cfg->chan % 8 ||
        cfg->chan > 512) {
        // (GroupNorm(8, C) of blocks.py:75 needs C % 8 == 0 in the reference too; multiples of 32 take the tuned kernels, other
        //  multiples of 8 the generic ones)
        set_error("unet_create: need 1 <= n_levels <= 8, in_ch >= 1, unet_chan a multiple of 8 in [8, 512]");
        return nullptr;
    }
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->mults[i] < 1) { set_error("unet_create: unet_dims entries must be >= 1"); return nullptr; }
    if (cfg->mults[0] != 1) {
        // final_conv is Block(dim, dim) applied to the last up level's dim * unet_dims[0] channels (unet.py:59-72): the reference
        // itself cannot run a forward with unet_dims[0] != 1 (its Conv2d raises on the channel mismatch), so this is not a narrowing
        set_error("unet_create: unet_dims[0] must be 1 (the reference's final Block(dim, dim) meets dim * unet_dims[0] channels "
                  "otherwise and raises, unet.py:69-72)");
        return nullptr;
    }
    {   // best effort: a plan can be created on a host without a GPU (shape / FLOP queries); launches re-check
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && ensure_device_init() != DDK_OK) return nullptr;
        (void)hipGetLastError();
    }
    ddk_unet* u = new ddk_unet();
    u->cfg = *cfg;
    u->L = cfg->n_levels;
    u->generic = cfg->chan % 32 != 0;
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->chan * cfg->mults[i] > 512) { set_error("unet_create: level widths above 512 channels are not supported"); delete u; return nullptr; }
    u->dims.push_back(cfg->in_ch);
    for (int i = 0; i < u->L; ++i) u->dims.push_back(cfg->chan * cfg->mults[i]);
    for (int d : u->dims) u->dimp.push_back(u->generic ? pad32(d) : d);
    u->time_dim = cfg->chan;
    u->temb_total = total_temb(*u);
    const int td = u->time_dim;

    u->freqs = u->add_copy("@sinusoidal_freqs", td / 2);
    u->w1t = u->alloc((size_t)td * 4 * td);
    u->slots.push_back(Slot{"time_mlp.1.weight", (long long)4 * td * td, PK_LINEAR_T, u->w1t, 4 * td, td, 0, 0, 0, 4 * td, 0});
    u->b1 = u->add_copy("time_mlp.1.bias", 4 * td);
    u->w2t = u->alloc((size_t)4 * td * td);
    u->slots.push_back(Slot{"time_mlp.3.weight", (long long)4 * td * td, PK_LINEAR_T, u->w2t, td, 4 * td, 0, 0, 0, td, 0});
    u->b2 = u->add_copy("time_mlp.3.bias", td);
    u->temb_wt = u->alloc((size_t)td * u->temb_total);
    u->temb_bias = u->alloc((size_t)u->temb_total);

    int cur = 0;
    for (int l = 0; l < u->L; ++l) {
        const std::string p = "downs." + std::to_string(l) + ".";
        const int ci = u->dims[l], co = u->dims[l + 1];
        u->down_res.push_back(u->add_res(p + "0.", ci, co, cur));
        if (l == 0 && ci <= 8 && co <= 256 && !u->generic) {
            // the network's first conv also gets conv_first.hip's layout (K = 9 * C_in exactly, unpadded input)
            ConvW& c1 = u->down_res[0].c1;
            const int K2 = (9 * ci + 1) / 2;
            c1.wf = u->alloc((size_t)(co / 32) * K2 * 64);
            c1.has_wf = true;
            u->slots.push_back(Slot{p + "0.block1.block.0.weight", (long long)co * ci * 9, PK_FIRST, c1.wf, co, ci, 3, 3, 0, 0, 0});
        }
        u->down_res.push_back(u->add_res(p + "1.", co, co, cur));
        u->down_attn.push_back(u->add_attn(p + "2.", co));
        if (l < u->L - 1) u->down_conv.push_back(u->add_conv(p + "3.conv.", co, co, 3, true, false, 0, /*local=*/l == u->L - 2));
    }
    const int mid = u->dims[u->L];
    u->mid1 = u->add_res("mid_block1.", mid, mid, cur);
    u->mid_attn = u->add_attn("mid_attn.", mid);
    u->mid2 = u->add_res("mid_block2.", mid, mid, cur);
    for (int i = 0; i < u->L - 1; ++i) {
        const std::string p = "ups." + std::to_string(i) + ".";
        const int din = u->dims[u->L - 1 - i], dout = u->dims[u->L - i];  // reversed(in_out[1:])[i]
        u->up_res.push_back(u->add_res(p + "0.", 2 * dout, din, cur, dout));
        u->up_res.push_back(u->add_res(p + "1.", din, din, cur));
        u->up_attn.push_back(u->add_attn(p + "2.", din));
        u->up_conv.push_back(u->add_convT(p + "3.conv.", din, /*local=*/i == 0));
    }
    u->final_conv = u->add_conv("final_conv.0.block.0.", cfg->chan, cfg->chan, 3, true, true);
    u->final_norm = u->add_norm("final_conv.0.block.1.weight", "final_conv.0.block.1.bias", cfg->chan);
    if (u->generic) {        // rows of pitch pad32(chan): the projection reads the padded activation
        u->final_w = u->alloc((size_t)cfg->in_ch * pad32(cfg->chan));
        u->slots.push_back(Slot{"final_conv.1.weight", (long long)cfg->in_ch * cfg->chan, PK_ROWS, u->final_w, cfg->in_ch, cfg->chan, 0, 0, 0,
                                pad32(cfg->chan), 0});
    } else {
        u->final_w = u->add_copy("final_conv.1.weight", (long long)cfg->in_ch * cfg->chan);
    }
    u->final_b = u->add_copy("final_conv.1.bias", cfg->in_ch);
    if (cur != u->temb_total) {
        set_error("unet_create: internal temb accounting mismatch");
        delete u;
        return nullptr;
    }
    return u;
}

static void drop_graphs(ddk_unet* u) {   // caller holds u->mu (or owns u exclusively)
    for (SamplerGraph& g : u->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        if (g.exec_multi) (void)hipGraphExecDestroy(g.exec_multi);
        if (g.graph_multi) (void)hipGraphDestroy(g.graph_multi);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    u->graphs.clear();
    u->table.ws = nullptr;
}

/* Destroying a plan that has captured sampler graphs waits for the device first (an executable graph must outlive its
   in-flight launches). */
extern "C" void ddk_unet_destroy(ddk_unet* u) {
    if (!u) return;
    if (!u->graphs.empty()) (void)hipDeviceSynchronize();
    drop_graphs(u);
    delete u;
}

extern "C" int ddk_sampler_invalidate(ddk_unet* u) {
    DDK_REQUIRE(u, "sampler_invalidate: null plan");
    std::lock_guard<std::mutex> lock(u->mu);
    if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
    drop_graphs(u);
    return DDK_OK;
}
static void destroy_entry(SamplerGraph& g) {
    if (g.done) { (void)hipEventSynchronize(g.done); (void)hipEventDestroy(g.done); }
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    if (g.exec_multi) (void)hipGraphExecDestroy(g.exec_multi);
    if (g.graph_multi) (void)hipGraphDestroy(g.graph_multi);
}

/* Drops only what points into `workspace` (the caller is about to free or reuse it): the cached graphs captured on it, after
   waiting for THEIR launches (the per-entry event), and the shift table if it lives there.  Other buffer sets keep their graphs. */
extern "C" int ddk_sampler_release_workspace(ddk_unet* u, const void* workspace) {
    DDK_REQUIRE(u && workspace, "sampler_release_workspace: arguments");
    std::lock_guard<std::mutex> lock(u->mu);
    for (size_t i = 0; i < u->graphs.size();) {
        if (u->graphs[i].ws == workspace) {
            destroy_entry(u->graphs[i]);
            u->graphs.erase(u->graphs.begin() + (long)i);
        } else {
            ++i;
        }
    }
    if (u->table.ws == workspace) u->table.ws = nullptr;
    return DDK_OK;
}

extern "C" int ddk_unet_set_option(ddk_unet* u, int option, int value) {
    DDK_REQUIRE(u, "unet_set_option: null plan");
    if (option == DDK_OPT_CLUSTER_GROUPNORM) {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);                          // a captured step bakes the choice in
        u->cluster_gn = value < 0 ? 0 : value > 2 ? 2 : value;
        return DDK_OK;
    }
    if (option == DDK_OPT_ATTENTION_FOLD) {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->attn_fold = value != 0;
        return DDK_OK;
    }
    if (option == DDK_OPT_ATTENTION_KV_CONTEXT) {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->attn_kvctx = value != 0;
        return DDK_OK;
    }
    if (option == DDK_OPT_FIRST_GROUPNORM) {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->first_gn = value != 0;
        return DDK_OK;
    }
    if (option == DDK_OPT_LEVEL_CHAIN) {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        // 0 off, 1 (default): the 4x4 level with its Downsample / Upsample convs, 2: the 4x4 level alone, 3: the 8x8 levels as well,
        // 4: the 8x8 levels only, 8 / 16: only downs[-2] / only ups[1]
        u->level_chain = value == 1 ? 9 : value == 2 ? 1 : value == 3 ? 15 : value == 4 ? 6 : value == 8 ? 2 : value == 16 ? 4 : value != 0 ? 9 : 0;
        return DDK_OK;
    }
    if (option == DDK_OPT_FOLD_DOWNSAMPLE_REDUCE) {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->fold_down_reduce = value != 0;
        return DDK_OK;
    }
    if (option == 10) {  // diagnostic (not in ddk.h): largest batch the level chain takes
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->level_chain_max_batch = value;
        return DDK_OK;
    }
    if (option == 9) {   // diagnostic (not in ddk.h): the folded attention block from this many pixels up
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->attn_fold_min_hw = value;
        return DDK_OK;
    }
    if (option == 2) {   // diagnostic (not in ddk.h): cap on the cluster launches per forward
        std::lock_guard<std::mutex> lock(u->mu);
        u->cluster_limit = value;
        return DDK_OK;
    }
    if (option == 4) {   // diagnostic (not in ddk.h): largest cluster size that takes the in-launch GroupNorm
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->cluster_np_max = value;
        return DDK_OK;
    }
    if (option == 11) {  // diagnostic (not in ddk.h): the in-launch GroupNorm on channel-chunk-split shapes
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->graphs.empty()) DDK_HIP(hipDeviceSynchronize());
        drop_graphs(u);
        u->cluster_split = value != 0;
        return DDK_OK;
    }
    return fail_arg("unet_set_option: unknown option");
}
extern "C" unsigned ddk_debug_cluster_timeouts(void) { return ddk::conv_wino_cluster_timeouts(); }
extern "C" int ddk_unet_num_slots(const ddk_unet* u) { return u ? (int)u->slots.size() : 0; }
extern "C" const char* ddk_unet_slot_name(const ddk_unet* u, int slot) {
    return (u && slot >= 0 && slot < (int)u->slots.size()) ? u->slots[slot].name.c_str() : nullptr;
}
extern "C" long long ddk_unet_slot_numel(const ddk_unet* u, int slot) {
    return (u && slot >= 0 && slot < (int)u->slots.size()) ? u->slots[slot].numel : -1;
}
extern "C" size_t ddk_unet_packed_bytes(const ddk_unet* u) { return u ? u->packed_floats * sizeof(float) : 0; }

// w[n][c] (packed 1x1 weight, row pitch cp) -> wg[n][c] = w*g[c], c1[n] = sum_c w*g, c2[n] = sum_c w*b   (one wave per n)
__global__ __launch_bounds__(64) void ln_fold_kernel(const float* __restrict__ w, const float* __restrict__ g, const float* __restrict__ b,
                                                     float* __restrict__ wg, float* __restrict__ c1, float* __restrict__ c2, int C, int cp) {
    const int n = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < cp; c += 64) {
        const float wv = w[(long long)n * cp + c];
        const float gv = c < C ? g[c] : 0.f, bv = c < C ? b[c] : 0.f;
        wg[(long long)n * cp + c] = wv * gv;
        s1 += wv * gv;
        s2 += wv * bv;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x == 0) { c1[n] = s1; c2[n] = s2; }
}

static int fold_attn(const AttnW& a, float* P, hipStream_t st) {
    hipLaunchKernelGGL(ln_fold_kernel, dim3(3 * HIDDEN), dim3(64), 0, st, P + a.qkv.w, P + a.ln.g, P + a.ln.b, P + a.qkv_lnw, P + a.ln_c1,
                       P + a.ln_c2, a.c, pad32(a.c));
    DDK_TRY(check_launch("ln_fold_kernel"));
    return qkv_operand_pack(P + a.qkv_lnw, P + a.qkv_op, HEADS, pad32(a.c), st);
}

// Derived weights of every attention site.  ddk_unet_pack_slot already re-derives a site's weights whenever one of its three
// source slots is packed (so a caller that packs every slot, in any order, ends up consistent); this entry point re-derives
// all of them explicitly, e.g. after writing into the packed arena by other means.
static void weights_changed(const ddk_unet* u) {
    ddk_unet* m = const_cast<ddk_unet*>(u);      // the cache bookkeeping is logically mutable
    std::lock_guard<std::mutex> lock(m->mu);
    ++m->pack_epoch;
}

extern "C" int ddk_unet_finalize_pack(const ddk_unet* u, void* packed, ddk_stream_t s) {
    DDK_REQUIRE(u && packed, "unet_finalize_pack: arguments");
    weights_changed(u);
    for (const AttnW& a : u->attn_all) DDK_TRY(fold_attn(a, static_cast<float*>(packed), as_stream(s)));
    return DDK_OK;
}

extern "C" int ddk_unet_pack_slot(const ddk_unet* u, int slot, const float* canonical, void* packed, ddk_stream_t s) {
    DDK_REQUIRE(u && canonical && packed && slot >= 0 && slot < (int)u->slots.size(), "unet_pack_slot: arguments");
    const Slot& sl = u->slots[slot];
    float* dst = static_cast<float*>(packed) + sl.off;
    weights_changed(u);
    int rc;
    switch (sl.kind) {
        case PK_COPY:
            DDK_HIP(hipMemcpyAsync(dst, canonical, (size_t)sl.numel * sizeof(float), hipMemcpyDeviceToDevice, as_stream(s)));
            rc = DDK_OK;
            break;
        case PK_CONV:
            rc = sl.split > 0 ? ddk_pack_conv_weight_split(canonical, dst, sl.O, sl.I, sl.KH, sl.KW, sl.i_pad, sl.split, sl.split_pad, s)
                              : ddk_pack_conv_weight(canonical, dst, sl.O, sl.I, sl.KH, sl.KW, sl.i_pad, s);
            break;
        case PK_CONVT: rc = sl.i_pad == sl.I ? ddk_pack_convT_weight(canonical, dst, sl.I, sl.O, s)
                                             : ddk_pack_convT_weight_padded(canonical, dst, sl.I, sl.O, sl.i_pad, s); break;
        case PK_ROWS:
            DDK_HIP(hipMemcpy2DAsync(dst, (size_t)sl.ld * sizeof(float), canonical, (size_t)sl.I * sizeof(float), (size_t)sl.I * sizeof(float),
                                     (size_t)sl.O, hipMemcpyDeviceToDevice, as_stream(s)));
            rc = DDK_OK;
            break;
        case PK_LINEAR_T: rc = ddk_pack_linear_T(canonical, dst, sl.O, sl.I, sl.ld, sl.col0, s); break;
        case PK_WINO: rc = ddk_pack_conv_weight_wino(canonical, dst, sl.O, sl.I, sl.i_pad, s); break;
        case PK_LOCAL: rc = ddk_pack_conv_weight_local(canonical, dst, sl.O, sl.I, sl.i_pad, s); break;
        case PK_LOCAL1: rc = ddk_pack_conv1x1_weight_local(canonical, dst, sl.O, sl.I, sl.i_pad, s); break;
        case PK_CONVT_LOCAL: rc = ddk_pack_convT_weight_local(canonical, dst, sl.I, sl.O, s); break;
        case PK_WLOCAL: rc = ddk_pack_conv_weight_wino_local(canonical, dst, sl.O, sl.I, sl.i_pad, s); break;
        case PK_FIRST: rc = ddk_pack_conv_weight_first(canonical, dst, sl.O, sl.I, s); break;
        case PK_CONVT_WINO: rc = ddk_pack_convT_weight_wino(canonical, dst, sl.I, sl.O, sl.i_pad, s); break;
        default: return fail_arg("unet_pack_slot: slot kind");
    }
    DDK_TRY(rc);
    // to_qkv weight / LayerNorm g / b of an attention site: re-derive its folded weights (stream-ordered after the pack above;
    // whichever of the three is packed last leaves them consistent)
    if (sl.attn >= 0) DDK_TRY(fold_attn(u->attn_all[sl.attn], static_cast<float*>(packed), as_stream(s)));
    return DDK_OK;
}

// ------------------------------------------------------------------------------------------------
namespace ddk {

// Workspace carve-up (all offsets in floats, 16-byte aligned).
struct Layout {
    size_t act = 0;      // one general activation buffer (max over layers of M*C)
    size_t qkv = 0, o = 0, ctx = 0, splitk = 0, gn_ws = 0, cl = 0, chain = 0;
    std::vector<size_t> skip;  // per level
    size_t xpad = 0, temb = 0, tact = 0;
    // offsets
    size_t off_A = 0, off_B = 0, off_C = 0, off_raw = 0, off_a1 = 0, off_res = 0, off_xn = 0, off_qkv = 0, off_o = 0, off_ctx = 0,
           off_splitk = 0, off_gn = 0, off_xpad = 0, off_temb = 0, off_tact = 0, off_cl = 0, off_chain = 0;
    std::vector<size_t> off_skip;
    size_t total = 0;
};

static size_t al4(size_t n) { return (n + 3) / 4 * 4; }

static void upd(size_t& m, size_t v) { if (v > m) m = v; }

// The 3x3 stride-1 convs run as Winograd F(2x2,3x3) wherever the shape allows (conv_wino.hip); their split count and slab
// workspace follow that kernel's plan.
static bool convT_wino_use(int B, int H, int W, int cin, int N) {
    return convT_wino_ok(H, W, cin, N) && convT_wino_splits(B, H, W, cin, N) <= 2;
}
static bool use_wino(const ConvW& cw, int H, int W, int cin, int N) {
    return cw.has_wu && conv_wino_ok(DDK_CONV3X3_S1, H, W, cin, N);
}
static int conv3_splits(const ConvW& cw, int B, int H, int W, int cin, int N) {
    return use_wino(cw, H, W, cin, N) ? conv_wino_splits(B, H, W, cin, N) : conv_splits(DDK_CONV3X3_S1, B, H, W, cin, N);
}
static size_t conv3_ws_floats(const ConvW& cw, int B, int H, int W, int cin, int N) {
    if (!use_wino(cw, H, W, cin, N)) return conv_workspace_bytes(DDK_CONV3X3_S1, B, H, W, cin, N) / 4;
    const int s = conv_wino_splits(B, H, W, cin, N);
    return s > 1 ? (size_t)s * B * H * W * N : 0;
}

// counters of the cluster GroupNorm: fixed place (16 words per (image, n tile), N <= 512) so every layer re-arms the same words
// ... and behind them one line for the sticky give-up count of this workspace (ddk_unet_cluster_check)
// ... and behind that the level chain's arrival and departure counters, one 128-byte line per image each
// ... and the pair counters of the channel-chunk-split in-launch GroupNorm: <= 128 (m tile, n tile) pairs (256 workgroups, >= 2 splits)
constexpr size_t CL_PAIR_WORDS = 128 * 16;
static size_t cl_counter_floats(int B) { return (size_t)B * 8 * 16 + 16 + (size_t)B * 64 + CL_PAIR_WORDS; }
static size_t cl_fail_offset(int B) { return (size_t)B * 8 * 16; }
static size_t cl_chain_offset(int B) { return (size_t)B * 8 * 16 + 16; }
static size_t cl_pair_offset(int B) { return (size_t)B * 8 * 16 + 16 + (size_t)B * 64; }
constexpr int CHAIN_BUFS = 18;                // activations that cross workgroups inside the level chain, [B][16][256] each
constexpr int CHAIN8_BUFS = 5;                // ... inside an 8x8 chain, [B][64][256] each

static void res_sizes(const ResW& r, int B, int H, int W, Layout& ly) {
    const size_t M = (size_t)B * H * W;
    upd(ly.act, M * r.co);
    upd(ly.splitk, conv3_ws_floats(r.c1, B, H, W, r.ci_pad, r.co));
    upd(ly.splitk, conv3_ws_floats(r.c2, B, H, W, r.co, r.co));
    if (r.has_res) upd(ly.splitk, conv_workspace_bytes(DDK_CONV1X1, B, H, W, r.ci_pad, r.co) / 4);
    upd(ly.gn_ws, groupnorm_workspace_bytes(B, H * W, r.co, GROUPS) / 4);
    upd(ly.gn_ws, (size_t)2 * GROUPS * (M / 128 + 1));      // {mean, M2} partials the Winograd conv leaves for GroupNorm
    upd(ly.cl, cl_counter_floats(B) + conv_wino_cluster_ws_floats(B, H, W, r.co));   // cluster GroupNorm: counters, then records
    if (H * W % 128 == 0 && H * W / 128 <= 8) upd(ly.cl, cl_counter_floats(B) + conv_first_gn_ws_floats(B, H, W));   // ... of the first Block
}

static void attn_sizes(const AttnW& a, int B, int H, int W, Layout& ly) {
    const size_t M = (size_t)B * H * W;
    upd(ly.act, M * a.c);
    upd(ly.qkv, M * 3 * HIDDEN);
    upd(ly.o, M * HIDDEN);
    upd(ly.ctx, (size_t)B * HEADS * 32 * 32);
    upd(ly.splitk, linattn_context_workspace_bytes(B, H * W, HEADS) / 4);   // the context partials reuse the split-K slab area
    if (attn_kvctx_ok(B, H * W, a.c, HEADS)) upd(ly.splitk, attn_kvctx_workspace_bytes(B, H * W) / 4);
    upd(ly.splitk, conv_workspace_bytes(DDK_CONV1X1, B, H, W, a.c, 3 * HIDDEN) / 4);
    upd(ly.splitk, conv_workspace_bytes(DDK_CONV1X1, B, H, W, HIDDEN, a.c) / 4);
}

// The level chain (level_chain.hip) takes the last level when it is a 4x4 map of 256 channels whose neighbours are 256 wide too (cfg1,
// cfg2, cfg4): ResnetBlocks without a skip conv on the way down and in the middle, one 512 -> 256 ResnetBlock on the way up.
static bool chain_plain(const ResW& r, bool wino) {
    return r.ci == 256 && r.co == 256 && !r.has_res && r.c1.has_bias && r.c2.has_bias && (wino ? r.c1.has_wwl && r.c2.has_wwl : r.c1.has_wl && r.c2.has_wl);
}
static bool chain_att(const AttnW& a) { return a.c == 256 && a.out.has_wl1 && a.out.has_bias; }
static bool chain_cat(const ResW& r, bool wino) {
    return r.ci == 512 && r.ci_pad == 512 && r.co == 256 && r.has_res && r.res.has_wl1 && r.res.has_bias && r.c1.cin_pad == 512 && r.c1.has_bias &&
           r.c2.has_bias && (wino ? r.c1.has_wwl && r.c2.has_wwl : r.c1.has_wl && r.c2.has_wl);
}
static bool level_chain_shape_ok(const ddk_unet& u, int H0, int W0) {
    if (u.generic || u.L < 2) return false;
    const int sh = u.L - 1;
    if ((H0 >> sh) != 4 || (W0 >> sh) != 4 || (H0 & ((1 << sh) - 1)) || (W0 & ((1 << sh) - 1))) return false;
    if (u.dimp[u.L] != 256 || u.dimp[u.L - 1] != 256 || GROUPS != 8) return false;
    return chain_plain(u.down_res[2 * sh], false) && chain_plain(u.down_res[2 * sh + 1], false) && chain_plain(u.mid1, false) &&
           chain_plain(u.mid2, false) && chain_plain(u.up_res[1], false) && chain_att(u.down_attn[sh]) && chain_att(u.mid_attn) &&
           chain_att(u.up_attn[0]) && chain_cat(u.up_res[0], false);
}
// ... and the level above it when that is an 8x8 map of 256 channels between 256-wide neighbours (cfg4: downs[2] and ups[1])
static bool level8_chain_shape_ok(const ddk_unet& u, int H0, int W0) {
    if (u.generic || u.L < 3) return false;
    const int sh = u.L - 2;
    if ((H0 >> sh) != 8 || (W0 >> sh) != 8 || (H0 & ((1 << sh) - 1)) || (W0 & ((1 << sh) - 1))) return false;
    if (u.dimp[sh + 1] != 256 || u.dimp[sh] != 256 || u.dimp[sh + 2] != 256 || GROUPS != 8) return false;
    return chain_plain(u.down_res[2 * sh], true) && chain_plain(u.down_res[2 * sh + 1], true) && chain_att(u.down_attn[sh]) &&
           chain_cat(u.up_res[2], true) && chain_plain(u.up_res[3], true) && chain_att(u.up_attn[1]);
}

static Layout make_layout(const ddk_unet& u, int B, int H0, int W0) {
    Layout ly;
    int H = H0, W = W0;
    ly.skip.resize(u.L);
    for (int l = 0; l < u.L; ++l) {
        res_sizes(u.down_res[2 * l], B, H, W, ly);
        res_sizes(u.down_res[2 * l + 1], B, H, W, ly);
        attn_sizes(u.down_attn[l], B, H, W, ly);
        ly.skip[l] = al4((size_t)B * H * W * u.dimp[l + 1]);
        if (l < u.L - 1) {
            upd(ly.splitk, conv_workspace_bytes(DDK_CONV3X3_S2, B, H, W, u.dimp[l + 1], u.dimp[l + 1]) / 4);
            H /= 2; W /= 2;
            upd(ly.act, (size_t)B * H * W * u.dimp[l + 1]);
        }
    }
    res_sizes(u.mid1, B, H, W, ly);
    attn_sizes(u.mid_attn, B, H, W, ly);
    res_sizes(u.mid2, B, H, W, ly);
    for (int i = 0; i < u.L - 1; ++i) {
        res_sizes(u.up_res[2 * i], B, H, W, ly);
        res_sizes(u.up_res[2 * i + 1], B, H, W, ly);
        attn_sizes(u.up_attn[i], B, H, W, ly);
        const int c = pad32(u.up_conv[i].cout);
        upd(ly.splitk, conv_workspace_bytes(DDK_CONVT4X4_S2, B, H, W, c, c) / 4);
        H *= 2; W *= 2;
        upd(ly.act, (size_t)B * H * W * c);
    }
    upd(ly.act, (size_t)B * H * W * pad32(u.cfg.chan));
    upd(ly.splitk, conv3_ws_floats(u.final_conv, B, H, W, pad32(u.cfg.chan), pad32(u.cfg.chan)));
    upd(ly.gn_ws, groupnorm_workspace_bytes(B, H * W, pad32(u.cfg.chan), GROUPS) / 4);
    upd(ly.gn_ws, (size_t)2 * GROUPS * ((size_t)B * H * W / 128 + 1));
    ly.xpad = al4((size_t)B * H0 * W0 * pad32(u.cfg.in_ch));
    ly.temb = al4((size_t)B * u.temb_total);
    ly.tact = al4((size_t)B * u.time_dim);
    ly.act = al4(ly.act); ly.qkv = al4(ly.qkv); ly.o = al4(ly.o); ly.ctx = al4(ly.ctx);
    ly.splitk = al4(ly.splitk); ly.gn_ws = al4(ly.gn_ws); ly.cl = al4(ly.cl);
    if (level_chain_shape_ok(u, H0, W0)) ly.chain = (size_t)CHAIN_BUFS * B * 16 * 256;
    if (level8_chain_shape_ok(u, H0, W0)) upd(ly.chain, (size_t)CHAIN8_BUFS * B * 64 * 256);     // the chains run one after the other

    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += n; return o; };
    ly.off_A = take(ly.act); ly.off_B = take(ly.act); ly.off_C = take(ly.act);
    ly.off_raw = take(ly.act); ly.off_a1 = take(ly.act); ly.off_res = take(ly.act); ly.off_xn = take(ly.act);
    ly.off_qkv = take(ly.qkv); ly.off_o = take(ly.o); ly.off_ctx = take(ly.ctx);
    ly.off_splitk = take(ly.splitk); ly.off_gn = take(ly.gn_ws); ly.off_cl = take(ly.cl); ly.off_chain = take(ly.chain);
    ly.off_xpad = take(ly.xpad); ly.off_temb = take(ly.temb); ly.off_tact = take(ly.tact);
    ly.off_skip.resize(u.L);
    for (int l = 0; l < u.L; ++l) ly.off_skip[l] = take(ly.skip[l]);
    ly.total = off;
    return ly;
}

struct Ctx {
    const ddk_unet& u;
    const float* P;  // packed weights
    float* W;        // workspace base
    const Layout& ly;
    int B;
    hipStream_t st;
    const float* temb;            // [B][temb_total], or the sampler's per-timestep table [t_start+1][temb_total]
    const long long* temb_rows;   // nullptr: row b;  sampler: row = t_cur[b] (device), so one table serves every step
    bool allow_cluster = false;   // this call's owner checks ddk_unet_cluster_check (sampler; forward only with option value 2)
    int n_cluster = 0;            // cluster launches issued so far in this forward
    const float* x_first = nullptr;   // the network input, unpadded, and its padded copy: the first conv may take the small-C_in
    const float* x_padded = nullptr;  // kernel on maps the one-launch first block does not cover (run_conv_gn)
};

static int run_conv(Ctx& c, int kind, const ConvW& cw, const float* src0, int c0, const float* src1, int c1, const float* resid,
                    float* out, int H, int W, int N, const float* weight_override = nullptr, const ConvLnFold* ln = nullptr) {
    ddk_conv_args a{};
    a.kind = kind;
    a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
    a.weight = weight_override ? weight_override : c.P + cw.w;
    a.bias = cw.has_bias ? c.P + cw.b : nullptr;
    a.resid = resid;
    a.out = out;
    a.B = c.B; a.H = H; a.W = W; a.N = N;
    a.pre_mish = 0;
    a.post_mish = 0;
    a.defer_reduce = 0;
    a.workspace = c.W + c.ly.off_splitk;
    a.workspace_bytes = c.ly.splitk * sizeof(float);
    if (kind == DDK_CONV3X3_S1 && !weight_override && use_wino(cw, H, W, c0 + c1, N)) a.weight_wino = c.P + cw.wu;
    // (measured, B = 32: 128 ch 16 -> 32: 47.8 -> 28.0 us; 256 ch 8 -> 16 (2 slabs): 49.4 -> 37.0; 256 ch 4 -> 8 would need 8 slabs of one
    //  chunk each and loses, 20.7 -> 21.9: the direct kernel keeps the maps that small)
    if (kind == DDK_CONVT4X4_S2 && !weight_override && cw.has_wu && !src1 && convT_wino_use(c.B, H, W, c0, N)) a.weight_wino = c.P + cw.wu;
    return conv_forward(a, c.st, ln);
}

// conv3x3 -> GroupNorm+Mish(+shift)(+residual).  When the conv splits k, its slabs stay in the workspace and the
// GroupNorm kernel sums them (plus the conv bias) while loading: one kernel and one HBM round trip fewer.
static bool conv_gn_is_local(const ConvW& cw, int B, int H, int W, int c0, int c1, int N) {
    return (cw.has_wl && (H * W == 16 || (H * W == 4 && B % 4 == 0)) && conv_gn_local_ok(H, W, c0 + c1, c0, N, GROUPS)) ||
           (cw.has_wwl && H * W == 64 && conv_gn_wlocal_ok(H, W, c0 + c1, c0, N, GROUPS));
}

static int run_conv_gn(Ctx& c, const ConvW& cw, const float* src0, int c0, const float* src1, int c1, float* raw, const NormW& n,
                       const float* temb, const float* addend, float* out, int H, int W, int N, const AddendSlabs& as = AddendSlabs(),
                       const AddendSlabs& ss = AddendSlabs()) {
    if (ss.n > 1 && !conv_gn_is_local(cw, c.B, H, W, c0, c1, N)) return fail_arg("run_conv_gn: slab source on a path that cannot sum it");
    if (c.u.generic) {
        // widths that are not multiples of 32: plain conv over the zero-padded weights, then GroupNorm over the REAL channels
        if (as.n > 1 || ss.n > 1) return fail_arg("run_conv_gn: slab addend / source on the generic path");
        DDK_TRY(run_conv(c, DDK_CONV3X3_S1, cw, src0, c0, src1, c1, nullptr, raw, H, W, N));
        return groupnorm_mish_generic(raw, c.P + n.g, c.P + n.b, temb, c.u.temb_total, addend, out, c.B, H * W, N, n.c_real, GROUPS, GN_EPS,
                                      c.st, c.temb_rows);
    }
    if (cw.has_wl && (H * W == 16 || (H * W == 4 && c.B % 4 == 0)) && conv_gn_local_ok(H, W, c0 + c1, c0, N, GROUPS))
        // 4x4 maps (and 2x2 maps, four images to a block): one image x 32 channels per workgroup, k reduced inside it -> statistics, Mish, shift and residual in the
        // conv's own epilogue, no slabs and no GroupNorm launch (conv_local.hip)
        return conv_gn_local(src0, c0, src1, c1, c.P + cw.wl, cw.has_bias ? c.P + cw.b : nullptr, c.P + n.g, c.P + n.b, temb,
                             c.u.temb_total, c.temb_rows, addend, out, c.B, H, W, N, GROUPS, GN_EPS, c.st, as, ss);
    if (cw.has_wwl && H * W == 64 && conv_gn_wlocal_ok(H, W, c0 + c1, c0, N, GROUPS))
        // 8x8 maps: the same image-local tiling in Winograd form
        return conv_gn_wlocal(src0, c0, src1, c1, c.P + cw.wwl, cw.has_bias ? c.P + cw.b : nullptr, c.P + n.g, c.P + n.b, temb,
                              c.u.temb_total, c.temb_rows, addend, out, c.B, H, W, N, GROUPS, GN_EPS, c.st, as, ss);
    if (as.n > 1) return fail_arg("run_conv_gn: slab addend on a path that cannot sum it");
    if (cw.has_wf && c.x_first && src0 == c.x_padded && !src1 && conv_first_ok(c.u.cfg.in_ch, N, H, W, GROUPS)) {
        // the network's first conv on a map too large for the one-launch first block (256x256: 512 statistics tiles per image):
        // still the K = 9 C_in kernel on the unpadded input instead of a 32-channel im2col conv (724 -> ~70 us at 8 x 256 x 256),
        // followed by the generic GroupNorm
        DDK_TRY(conv_first(c.x_first, c.P + cw.wf, cw.has_bias ? c.P + cw.b : nullptr, raw, nullptr, c.B, H, W, c.u.cfg.in_ch, N, GROUPS,
                           nullptr, nullptr, c.st));
        return groupnorm_mish(raw, c.P + n.g, c.P + n.b, temb, c.u.temb_total, addend, out, c.B, H * W, N, GROUPS, GN_EPS,
                              c.W + c.ly.off_gn, c.ly.gn_ws * sizeof(float), c.st, c.temb_rows);
    }
    const int cl_np = c.allow_cluster && cw.has_wu && conv_wino_cluster_device_ok() ? conv_wino_cluster_np(c.B, H, W, c0 + c1, N, GROUPS) : 0;
    if (cl_np > 0 && cl_np <= c.u.cluster_np_max && c.n_cluster++ < c.u.cluster_limit) {
        // one launch: the workgroups of an image exchange their tile statistics and normalise their own tile in registers
        ddk_conv_args a{};
        a.kind = DDK_CONV3X3_S1;
        a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
        a.weight = c.P + cw.w;
        a.weight_wino = c.P + cw.wu;
        a.bias = cw.has_bias ? c.P + cw.b : nullptr;
        a.resid = addend;
        a.out = out;
        a.B = c.B; a.H = H; a.W = W; a.N = N;
        float* cl = c.W + c.ly.off_cl;
        const WinoGnFuse f{c.P + n.g, c.P + n.b, temb, c.temb_rows, c.u.temb_total, GN_EPS, GROUPS, cl + cl_counter_floats(c.B),
                           reinterpret_cast<unsigned*>(cl), reinterpret_cast<unsigned*>(cl + cl_fail_offset(c.B))};
        return conv_forward(a, c.st, nullptr, &f);
    }
    int cls_splits = 1;
    const int cls_np = c.allow_cluster && c.u.cluster_split && cw.has_wu && conv_wino_cluster_device_ok()
                           ? conv_wino_cluster_split_np(c.B, H, W, c0 + c1, N, GROUPS, &cls_splits) : 0;
    if (cls_np > 0 && cls_np <= c.u.cluster_np_max && (size_t)cls_splits * c.B * H * W * N <= c.ly.splitk &&
        conv_wino_cluster_pair_words(c.B, H, W, N) <= CL_PAIR_WORDS && c.n_cluster++ < c.u.cluster_limit) {
        // the same on a shape whose channel chunks are split over workgroups (16x16 maps of 64-channel tiles at batch 32): a tile's first
        // workgroup sums its partners' partial tiles in the launch -- no slabs left for a GroupNorm launch to sum, no GroupNorm launch
        ddk_conv_args a{};
        a.kind = DDK_CONV3X3_S1;
        a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
        a.weight = c.P + cw.w;
        a.weight_wino = c.P + cw.wu;
        a.bias = cw.has_bias ? c.P + cw.b : nullptr;
        a.resid = addend;
        a.out = out;
        a.B = c.B; a.H = H; a.W = W; a.N = N;
        a.workspace = c.W + c.ly.off_splitk;
        a.workspace_bytes = c.ly.splitk * sizeof(float);
        float* cl = c.W + c.ly.off_cl;
        WinoGnFuse f{c.P + n.g, c.P + n.b, temb, c.temb_rows, c.u.temb_total, GN_EPS, GROUPS, cl + cl_counter_floats(c.B),
                     reinterpret_cast<unsigned*>(cl), reinterpret_cast<unsigned*>(cl + cl_fail_offset(c.B))};
        f.pairs = reinterpret_cast<unsigned*>(cl + cl_pair_offset(c.B));
        return conv_forward(a, c.st, nullptr, &f);
    }
    const int np = cw.has_wu ? conv_wino_stats_parts(c.B, H, W, c0 + c1, N, GROUPS) : 0;
    if (np > 0) {
        // one-pass Winograd conv: its epilogue leaves per-tile {mean, M2}; GroupNorm then is a single streaming read + write
        ddk_conv_args a{};
        a.kind = DDK_CONV3X3_S1;
        a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
        a.weight = c.P + cw.w;
        a.weight_wino = c.P + cw.wu;
        a.bias = cw.has_bias ? c.P + cw.b : nullptr;
        a.out = raw;
        a.B = c.B; a.H = H; a.W = W; a.N = N;
        a.gn_partials = c.W + c.ly.off_gn;
        a.gn_groups = GROUPS;
        DDK_TRY(conv_forward(a, c.st));
        return groupnorm_mish_parts(raw, c.W + c.ly.off_gn, np, c.P + n.g, c.P + n.b, temb, c.u.temb_total, addend, out, c.B, H * W, N,
                                    GROUPS, GN_EPS, c.st, c.temb_rows);
    }
    const int splits = conv3_splits(cw, c.B, H, W, c0 + c1, N);
    const bool resident = groupnorm_workspace_bytes(c.B, H * W, N, GROUPS) == 0;
    if (splits > 1 && resident) {
        ddk_conv_args a{};
        a.kind = DDK_CONV3X3_S1;
        a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
        a.weight = c.P + cw.w;
        if (use_wino(cw, H, W, c0 + c1, N)) a.weight_wino = c.P + cw.wu;
        a.out = raw;  // unused: the slabs are the result
        a.B = c.B; a.H = H; a.W = W; a.N = N;
        a.defer_reduce = 1;
        a.workspace = c.W + c.ly.off_splitk;
        a.workspace_bytes = c.ly.splitk * sizeof(float);
        DDK_TRY(conv_forward(a, c.st));
        return groupnorm_mish_ex(c.W + c.ly.off_splitk, splits, (long long)c.B * H * W * N, cw.has_bias ? c.P + cw.b : nullptr,
                                 c.P + n.g, c.P + n.b, temb, c.u.temb_total, addend, out, c.B, H * W, N, GROUPS, GN_EPS, nullptr, 0,
                                 c.st, c.temb_rows);
    }
    DDK_TRY(run_conv(c, DDK_CONV3X3_S1, cw, src0, c0, src1, c1, nullptr, raw, H, W, N));
    return groupnorm_mish(raw, c.P + n.g, c.P + n.b, temb, c.u.temb_total, addend, out, c.B, H * W, N, GROUPS, GN_EPS,
                          c.W + c.ly.off_gn, c.ly.gn_ws * sizeof(float), c.st, c.temb_rows);
}

static int run_gn(Ctx& c, const float* x, const NormW& n, const float* temb, const float* addend, float* out, int HW, int C) {
    return groupnorm_mish(x, c.P + n.g, c.P + n.b, temb, c.u.temb_total, addend, out, c.B, HW, C, GROUPS, GN_EPS,
                          c.W + c.ly.off_gn, c.ly.gn_ws * sizeof(float), c.st, c.temb_rows);
}

// blocks.py:105-115 (eval): out = Mish(GN(conv2(Mish(GN(conv1(x))) + temb))) + res(x)
// ss.n > 1 (res_takes_slab_source only): src0 is the split-K slab area the stride-2 conv in front left; both its readers -- the first
// Block's staging loop and the second Block's residual -- sum the slabs (+ that conv's bias) in splitk_reduce_kernel's order
static bool res_takes_slab_source(const ddk_unet& u, const ResW& r, int B, int H, int W, int c0) {
    return !u.generic && !r.has_res && conv_gn_is_local(r.c1, B, H, W, c0, 0, r.co) && conv_gn_is_local(r.c2, B, H, W, r.co, 0, r.co);
}

static int run_res(Ctx& c, const ResW& r, const float* src0, int c0, const float* src1, int c1, float* out, int H, int W,
                   const AddendSlabs& ss = AddendSlabs()) {
    float* raw = c.W + c.ly.off_raw;
    float* a1 = c.W + c.ly.off_a1;
    float* res = c.W + c.ly.off_res;
    // the 1x1 skip first: the second conv's split-K slabs and the skip conv's would otherwise share the workspace
    const float* addend = src0;
    AddendSlabs as;
    if (ss.n > 1) {
        if (r.has_res || src1) return fail_arg("run_res: slab source with a skip conv or a second source");
        as = ss;
    }
    if (r.has_res) {
        const int rs = conv_splits(DDK_CONV1X1, c.B, H, W, c0 + c1, r.co);
        if (rs > 1 && !conv1x1_sm_ok((long long)c.B * H * W, c0, c1, r.co) && conv_gn_is_local(r.c1, c.B, H, W, c0, c1, r.co) &&
            conv_gn_is_local(r.c2, c.B, H, W, r.co, 0, r.co)) {
            // neither Block conv touches the split-K workspace on these maps: the skip conv leaves its slabs there and the
            // second Block's epilogue sums them (+ bias) while it adds the residual -- no reduce launch
            ddk_conv_args a{};
            a.kind = DDK_CONV1X1;
            a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
            a.weight = c.P + r.res.w;
            a.out = res;                      // unused: the slabs are the result
            a.B = c.B; a.H = H; a.W = W; a.N = r.co;
            a.defer_reduce = 1;
            a.workspace = c.W + c.ly.off_splitk;
            a.workspace_bytes = c.ly.splitk * sizeof(float);
            DDK_TRY(conv_forward(a, c.st));
            addend = c.W + c.ly.off_splitk;
            as.n = rs;
            as.stride = (long long)c.B * H * W * r.co;
            as.bias = r.res.has_bias ? c.P + r.res.b : nullptr;
        } else {
            DDK_TRY(run_conv(c, DDK_CONV1X1, r.res, src0, c0, src1, c1, nullptr, res, H, W, r.co));
            addend = res;
        }
    }
    DDK_TRY(run_conv_gn(c, r.c1, src0, c0, src1, c1, raw, r.n1, c.temb + r.temb_off, nullptr, a1, H, W, r.co, AddendSlabs(), ss));
    return run_conv_gn(c, r.c2, a1, r.co, nullptr, 0, raw, r.n2, nullptr, addend, out, H, W, r.co, as);
}

// The folded form of the attention block (attention.hip, attn_fold_kernel): maps with many more pixels than channels, C = 128
static bool attn_fold_eligible(const ddk_unet& u, const AttnW& a, int B, int H, int W) {
    const long long M = (long long)B * H * W;
    return u.attn_fold && H * W > u.attn_fold_min_hw && (H * W) % 64 == 0 && attn_fold_ok(a.c, HEADS) && conv1x1_ws_ok(M, a.c, a.c) &&
           conv_ln_fold_ok(B, H, W, a.c, 2 * HIDDEN) && B <= 256;
}

// blocks.py:8-14,63-71,126-134: out = to_out(attn(to_qkv(LN(x)))) + x
static int run_attn(Ctx& c, const AttnW& a, const float* x, float* out, int H, int W) {
    float* xn = c.W + c.ly.off_xn;
    float* qkv = c.W + c.ly.off_qkv;
    float* ctx = c.W + c.ly.off_ctx;
    float* o = c.W + c.ly.off_o;
    const long long M = (long long)c.B * H * W;
    if (c.u.generic) {
        // LayerNorm over the real channels, then the plain projections over zero-padded weights
        DDK_TRY(chan_layernorm_generic(x, c.P + a.ln.g, c.P + a.ln.b, xn, M, a.c, a.c_real, LN_EPS, c.st));
        DDK_TRY(run_conv(c, DDK_CONV1X1, a.qkv, xn, a.c, nullptr, 0, nullptr, qkv, H, W, 3 * HIDDEN));
        if (H * W <= 256) {
            DDK_TRY(linattn_fused_small(qkv, ctx, o, c.B, H * W, HEADS, c.st));
        } else {
            DDK_TRY(linattn_context(qkv, ctx, c.B, H * W, HEADS, c.W + c.ly.off_splitk, c.ly.splitk * sizeof(float), c.st));
            DDK_TRY(linattn_apply(qkv, ctx, o, c.B, H * W, HEADS, c.st));
        }
        return run_conv(c, DDK_CONV1X1, a.out, o, HIDDEN, nullptr, 0, x, out, H, W, a.c);
    }
    if (H * W <= 16 && a.c % 32 == 0 && linattn_small_qkv_ok(H * W, a.c)) {
        // 4x4 maps: projection (LayerNorm folded), context and apply of one (image, head) in one workgroup -- no qkv tensor,
        // 14.6 us instead of 11.0 + 5.2.  (On 8x8 maps the projection is 4x the work on the same 128 workgroups -- half the
        // chip, one wave per SIMD: 25.5 us against 12.5 + 5.7 for the two launches, so those keep the im2col kernel.)
        DDK_TRY(linattn_small_qkv(x, c.P + a.qkv_op, c.P + a.ln_c1, c.P + a.ln_c2, LN_EPS, ctx, o, c.B, H * W, a.c, HEADS, c.st));
        return run_conv(c, DDK_CONV1X1, a.out, o, HIDDEN, nullptr, 0, x, out, H, W, a.c);
    }
    if (attn_fold_eligible(c.u, a, c.B, H, W)) {
        // q is linear in this attention: project k and v only, build the context, fold to_out . ctx^T . W_q (and the LayerNorm)
        // into one C x C matrix per image and apply it to x as a per-image 1x1 conv with the residual -- no q third of to_qkv, no
        // apply kernel, no separate to_out
        if (c.u.attn_kvctx && attn_kvctx_ok(c.B, H * W, a.c, HEADS) && c.ly.splitk * sizeof(float) >= attn_kvctx_workspace_bytes(c.B, H * W)) {
            // round 5: projection and context in one launch -- the 33.5 MB kv tensor of the 32x32 level is never written
            DDK_TRY(attn_kvctx(x, c.P + a.qkv_lnw + (size_t)HIDDEN * a.c, c.P + a.ln_c1 + HIDDEN, c.P + a.ln_c2 + HIDDEN, LN_EPS, ctx, c.B, H * W,
                               c.W + c.ly.off_splitk, c.ly.splitk * sizeof(float), c.st));
        } else {
            const ConvLnFold lnkv{c.P + a.ln_c1 + HIDDEN, c.P + a.ln_c2 + HIDDEN, LN_EPS};
            DDK_TRY(run_conv(c, DDK_CONV1X1, a.qkv, x, a.c, nullptr, 0, nullptr, qkv, H, W, 2 * HIDDEN, c.P + a.qkv_lnw + (size_t)HIDDEN * a.c,
                             &lnkv));
            // (round 4: merging the split context partials inside attn_fold_kernel instead of the 5.5 us merge launch made that kernel 13.0 ->
            //  25.6 us -- each of an image's four workgroups redoes the 4 x 8 partial reads -- so the merge launch stays)
            DDK_TRY(linattn_context(qkv, ctx, c.B, H * W, HEADS, c.W + c.ly.off_splitk, c.ly.splitk * sizeof(float), c.st, true));
        }
        float* A = o;                                   // [B][C][C], then a1, a2 [B][C] (the apply output buffer is free on this path)
        float* a1 = A + (size_t)c.B * a.c * a.c;
        float* a2 = a1 + (size_t)c.B * a.c;
        DDK_TRY(attn_fold(ctx, c.P + a.qkv_lnw, c.P + a.ln_c1, c.P + a.ln_c2, c.P + a.out.w, a.out.has_bias ? c.P + a.out.b : nullptr, A, a1,
                          a2, c.B, a.c, HEADS, c.st));
        const ConvLnFold lnA{a1, a2, LN_EPS};
        return conv1x1_ws(x, A, nullptr, x, out, M, a.c, &lnA, c.st, c.B);
    }
    if (conv_ln_fold_ok(c.B, H, W, a.c, 3 * HIDDEN)) {
        // LayerNorm folded into the projection: no LayerNorm launch, no normalised copy of x
        const ConvLnFold ln{c.P + a.ln_c1, c.P + a.ln_c2, LN_EPS};
        DDK_TRY(run_conv(c, DDK_CONV1X1, a.qkv, x, a.c, nullptr, 0, nullptr, qkv, H, W, 3 * HIDDEN, c.P + a.qkv_lnw, &ln));
    } else {
        DDK_TRY(chan_layernorm(x, c.P + a.ln.g, c.P + a.ln.b, xn, M, a.c, LN_EPS, c.st));
        DDK_TRY(run_conv(c, DDK_CONV1X1, a.qkv, xn, a.c, nullptr, 0, nullptr, qkv, H, W, 3 * HIDDEN));
    }
    if (H * W <= 256) {      // 16x16 and smaller: k, v, q of one (image, head) fit the LDS -- context, merge and apply in one launch
                             // (256 rows: both products on the matrix pipe; the VALU form was LDS-bound there, 23 us)
        DDK_TRY(linattn_fused_small(qkv, ctx, o, c.B, H * W, HEADS, c.st));
    } else {
        // (folding the split context's merge into the apply kernel's fragment build was tried: 73 vs 11 + 5.5 us at 32x32 --
        //  hundreds of dependent L2 loads per lane; the 5 us merge launch stays)
        DDK_TRY(linattn_context(qkv, ctx, c.B, H * W, HEADS, c.W + c.ly.off_splitk, c.ly.splitk * sizeof(float), c.st));
        DDK_TRY(linattn_apply(qkv, ctx, o, c.B, H * W, HEADS, c.st));
    }
    return run_conv(c, DDK_CONV1X1, a.out, o, HIDDEN, nullptr, 0, x, out, H, W, a.c);
}

// Levels as persistent launches (level_chain.hip; unet.py:83-101).  part 0: the last level whole -- downs[-1], mid, ups[0] -- on 4x4 maps:
// `in` = the Downsample conv's output, `skip` receives the level's skip tensor (unet.py:87), `out` the up attention block's output.
// part 1: downs[-2] on 8x8 maps (in -> skip).  part 2: ups[1] on 8x8 maps (cat(in, skip) -> out).
static bool level_chain_use(const Ctx& c, int H0, int W0) {
    return c.allow_cluster && (c.u.level_chain & 1) && c.B <= c.u.level_chain_max_batch && c.ly.chain > 0 && level_chain_shape_ok(c.u, H0, W0) &&
           level_chain_device_ok();
}
static bool level8_chain_use(const Ctx& c, int H0, int W0, int bit) {
    return c.allow_cluster && (c.u.level_chain & bit) && c.B <= c.u.level_chain_max_batch && c.ly.chain > 0 && level8_chain_shape_ok(c.u, H0, W0) && level_chain_device_ok();
}

// down_src != null (part 0): `in` is not used -- the chain starts with the Downsample conv on the 8x8 map down_src (blocks.py:41-47);
// up_out != null (part 0): the chain ends with the Upsample transpose conv (blocks.py:32-38) into the 8x8 map up_out, `out` is not used
static int run_level_chain(Ctx& c, int part, const float* in, float* skip, float* out, const float* down_src = nullptr, float* up_out = nullptr) {
    const ddk_unet& u = c.u;
    const int hw = part == 0 ? 16 : 64;
    const bool wino = hw == 64;
    ChainParams p{};
    int n = 0, e = 0;
    const size_t per = (size_t)c.B * hw * 256;
    auto buf = [&](int i) { return c.W + c.ly.off_chain + (size_t)i * per; };
    auto conv3 = [&](const ConvW& cw, const NormW& nw, const float* s0, int c0, const float* s1, int c1, int temb_off, int flags, float* o) {
        ChainOp& op = p.op[n++];
        op = ChainOp{s0, s1, c.P + (wino ? cw.wwl : cw.wl), c.P + cw.b, c.P + nw.g, c.P + nw.b, o, c0, c1, CH_CONV3, flags, temb_off, 256};
    };
    auto conv1 = [&](const ConvW& cw, const float* s0, int c0, const float* s1, int c1, int flags, float* o) {
        ChainOp& op = p.op[n++];
        op = ChainOp{s0, s1, c.P + cw.wl1, c.P + cw.b, nullptr, nullptr, o, c0, c1, CH_CONV1, flags, -1, 256};
    };
    auto attn = [&](const AttnW& a, const float* s0, float* o) {
        ChainOp& op = p.op[n++];
        op = ChainOp{s0, nullptr, c.P + a.qkv_op, nullptr, c.P + a.ln_c1, c.P + a.ln_c2, o, a.c, 0, CH_ATTN, CHF_WAIT | CHF_SIGNAL, -1, HIDDEN};
    };
    const int WS = CHF_WAIT | CHF_SIGNAL, RES = CHF_ADD_KEEP | CHF_SAVE_KEEP;
    // a ResnetBlock without a skip conv (blocks.py:105-115): x enters as `keep`
    auto res_plain = [&](const ResW& r, const float* x, bool external, float* o) {
        float* h = buf(e++);
        conv3(r.c1, r.n1, x, 256, nullptr, 0, r.temb_off, external ? (CHF_SIGNAL | CHF_KEEP_FROM_SRC) : WS, h);
        conv3(r.c2, r.n2, h, 256, nullptr, 0, -1, WS | RES, o);
    };
    // Residual(PreNorm(LinearAttention)) (blocks.py:8-14, 63-71, 116-134): x is `keep`
    auto attn_block = [&](const AttnW& a, const float* x, float* o, bool signals) {
        float* heads = buf(e++);
        attn(a, x, heads);
        conv1(a.out, heads, HIDDEN, nullptr, 0, (signals ? WS : CHF_WAIT) | RES, o);
    };
    // a ResnetBlock on cat(x, skip) -> 512 channels, res_conv is a 1x1 (its result lives in keep2); x_external: x comes from another kernel
    auto res_cat = [&](const ResW& r, const float* x, bool x_external, const float* sk, float* o) {
        conv1(r.res, x, 256, sk, 256, (x_external ? 0 : CHF_WAIT) | CHF_SAVE_KEEP2 | CHF_NO_OUT, nullptr);
        float* h = buf(e++);
        conv3(r.c1, r.n1, x, 256, sk, 256, r.temb_off, CHF_SIGNAL, h);
        conv3(r.c2, r.n2, h, 256, nullptr, 0, -1, WS | CHF_ADD_KEEP2 | CHF_SAVE_KEEP, o);
    };
    if (part == 0) {
        const int sh = u.L - 1;
        if (down_src) {
            const ConvW& dc = u.down_conv[sh - 1];
            float* dn = buf(e++);
            ChainOp& op = p.op[n++];
            op = ChainOp{down_src, nullptr, c.P + dc.wl, c.P + dc.b, nullptr, nullptr, dn, 256, 0, CH_CONV3,
                         CHF_DOWN | CHF_NO_GN | CHF_SIGNAL | CHF_SAVE_KEEP, -1, 256};
            in = dn;
        }
        float* d0 = buf(e++);
        res_plain(u.down_res[2 * sh], in, down_src == nullptr, d0);
        float* d1 = buf(e++);
        res_plain(u.down_res[2 * sh + 1], d0, false, d1);
        attn_block(u.down_attn[sh], d1, skip, true);
        float* m1 = buf(e++);
        res_plain(u.mid1, skip, false, m1);
        float* ma = buf(e++);
        attn_block(u.mid_attn, m1, ma, true);
        float* m2 = buf(e++);
        res_plain(u.mid2, ma, false, m2);
        float* u0 = buf(e++);
        res_cat(u.up_res[0], m2, false, skip, u0);
        float* u1 = buf(e++);
        res_plain(u.up_res[1], u0, false, u1);
        if (up_out) {
            float* ua = buf(e++);
            attn_block(u.up_attn[0], u1, ua, true);
            const ConvW& uc = u.up_conv[0];
            ChainOp& op = p.op[n++];
            op = ChainOp{ua, nullptr, c.P + uc.wtl, c.P + uc.b, nullptr, nullptr, up_out, 256, 0, CH_UPT, CHF_WAIT, -1, 256};
        } else {
            attn_block(u.up_attn[0], u1, out, false);
        }
    } else if (part == 1) {
        const int sh = u.L - 2;
        float* d0 = buf(e++);
        res_plain(u.down_res[2 * sh], in, true, d0);
        float* d1 = buf(e++);
        res_plain(u.down_res[2 * sh + 1], d0, false, d1);
        attn_block(u.down_attn[sh], d1, skip, false);
    } else {
        float* u0 = buf(e++);
        res_cat(u.up_res[2], in, true, skip, u0);
        float* u1 = buf(e++);
        res_plain(u.up_res[3], u0, false, u1);
        attn_block(u.up_attn[1], u1, out, false);
    }
    if (n > CH_MAX_OPS || e > (part == 0 ? CHAIN_BUFS : CHAIN8_BUFS)) return fail_arg("level_chain: internal op / buffer count");
    p.n_ops = n;
    p.B = c.B;
    p.hw = hw;
    p.temb = c.temb;
    p.temb_rows = c.temb_rows;
    p.temb_stride = u.temb_total;
    float* cl = c.W + c.ly.off_cl;
    p.cnt = reinterpret_cast<unsigned*>(cl + cl_chain_offset(c.B));
    p.done = p.cnt + (size_t)c.B * 32;
    p.fail = reinterpret_cast<unsigned*>(cl + cl_fail_offset(c.B));
    p.gn_eps = GN_EPS;
    p.ln_eps = LN_EPS;
    return level_chain_launch(p, c.st);
}

// What a reverse step of the sampler adds to a forward: the bookkeeping in front (t_cur[b] <- counter; counter -= 1) and the
// update of x behind it (ddpm.py:203-227).  Both ride on the forward's own first / last kernel where the shape allows.
struct StepArgs {
    int64_t* state;               // [0] step counter, [1] Philox seed, [2] stream id
    float* x;                     // chain state, updated in place
    float* eps_hat;               // scratch for the unfused tail
    const float* noise;
    long long noise_step_stride;
    int t_first;
    const float *c_recip, *c_recipm1, *c1, *c2, *sigma;
    long long per;
};

// t_cur[b] = counter for every sample, then counter -= 1; also zero-pads x into xpad.  First kernel of a step on shapes the
// first-layer kernel does not take: the previous step's kernels have all completed (stream order), nobody else reads the counter.
__global__ __launch_bounds__(256) void step_prepare_kernel(const float* __restrict__ x, float* __restrict__ xpad, long long total,
                                                           int C, int c_pad, int64_t* __restrict__ counter,
                                                           int64_t* __restrict__ t_cur, int B) {
    if (blockIdx.x == 0) {
        const int64_t v = *counter;
        for (int b = threadIdx.x; b < B; b += blockDim.x) t_cur[b] = v;
        __syncthreads();
        if (threadIdx.x == 0) *counter = v - 1;
    }
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % c_pad);
        const long long m = i / c_pad;
        xpad[i] = ch < C ? x[m * C + ch] : 0.f;
    }
}

// The first ResnetBlock on the unpadded input: conv_first.hip for its first conv, and its 1x1 res_conv evaluated inside the
// second GroupNorm launch (norm_act.hip, Res1x1) -- no pad kernel, no padded 32-channel conv, no res_conv launch.  Needs the
// second conv to leave GroupNorm partials (one-pass Winograd shape).
static bool first_fast(const ddk_unet& u, int B, int H, int W) {
    const ResW& r = u.down_res[0];
    if (!r.c1.has_wf || !r.has_res || !r.c2.has_wu) return false;
    if (!conv_first_ok(r.ci, r.co, H, W, GROUPS)) return false;
    const int upr = r.co / 4;
    if (upr > 256 || 256 % upr) return false;
    if ((long long)H * W / 128 * GROUPS > 1024) return false;
    return conv_wino_stats_parts(B, H, W, r.co, r.co, GROUPS) > 0;
}

// x: NHWC input, unpadded ([B][H0][W0][in_ch]).
// temb_table != nullptr (sampler): the per-block time shifts of EVERY timestep were computed once up front
// (build_temb_table); row t[b] of the table is read directly by the GroupNorm kernels and no time kernel runs per step.
// step != nullptr (sampler): t is the t_cur array the step's first kernel fills from step->state, and the forward ends with
// the update of step->x instead of writing eps_hat to `out`.
static int forward_core(const ddk_unet& u, const float* P, const float* x, int64_t* t, float* out, int B, int H0, int W0,
                        float* ws, const Layout& ly, hipStream_t st, bool allow_cluster, const float* temb_table = nullptr,
                        const StepArgs* step = nullptr) {
    float* tact = ws + ly.off_tact;
    float* temb = ws + ly.off_temb;
    float* xpad = ws + ly.off_xpad;
    const bool fast0 = first_fast(u, B, H0, W0);
    if (!fast0) {
        const int C = u.cfg.in_ch, cp = pad32(C);
        const long long padded = (long long)B * H0 * W0 * cp;
        if (step) {
            const int prep_blocks = (int)(ceil_div(padded, 256) < 1024 ? ceil_div(padded, 256) : 1024);
            hipLaunchKernelGGL(step_prepare_kernel, dim3(prep_blocks), dim3(256), 0, st, x, xpad, padded, C, cp, step->state, t, B);
            DDK_TRY(check_launch("step_prepare_kernel"));
        } else {
            DDK_TRY(ddk_pad_channels(x, xpad, (long long)B * H0 * W0, C, cp, st));
        }
    }
    if (!temb_table) {
        DDK_TRY(time_mlp(t, P + u.freqs, P + u.w1t, P + u.b1, P + u.w2t, P + u.b2, tact, nullptr, B, u.time_dim, st));
        DDK_TRY(time_proj(tact, P + u.temb_wt, P + u.temb_bias, temb, B, u.time_dim, u.temb_total, st));
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "timestep rows are read as long long");
    Ctx c{u, P, ws, ly, B, st, temb_table ? temb_table : temb, temb_table ? reinterpret_cast<const long long*>(t) : nullptr};
    if (!fast0) { c.x_first = x; c.x_padded = xpad; }
    c.allow_cluster = allow_cluster;
    float* bufA = ws + ly.off_A;
    float* bufB = ws + ly.off_B;
    float* bufC = ws + ly.off_C;
    float* raw = ws + ly.off_raw;
    float* a1 = ws + ly.off_a1;
    float* gnp = ws + ly.off_gn;

    int H = H0, W = W0;
    const float* cur = xpad;
    int cur_c = pad32(u.cfg.in_ch);
    AddendSlabs cur_ss;           // > 1 slab: `cur` is the split-K area a Downsample conv left for the next ResnetBlock to sum
    // the first Block with its GroupNorm in the conv's launch: in the sampler that kernel leaves the step counter alone (every workgroup of
    // it reads the counter) and the step's LAST kernel decrements it (dec_counter below)
    const bool first_fused = fast0 && c.allow_cluster && u.first_gn && conv_wino_cluster_device_ok() &&
                             conv_first_gn_ok(u.down_res[0].ci, u.down_res[0].co, H0, W0, GROUPS) && u.cluster_limit > 0;
    int64_t* dec_counter = (first_fused && step) ? step->state : nullptr;
    const bool chained = level_chain_use(c, H0, W0);    // the last level (4x4 maps) as one persistent launch
    const bool chain_edges = chained && (u.level_chain & 8) && u.L >= 2 && u.down_conv[u.L - 2].has_wl && u.down_conv[u.L - 2].has_bias &&
                             u.up_conv[0].has_wtl && u.up_conv[0].cin == 256 && u.down_conv[u.L - 2].cin == 256;
    const bool chained8 = level8_chain_use(c, H0, W0, 2);   // the level above it (8x8 maps): downs[-2] as one launch,
    const bool chained8u = level8_chain_use(c, H0, W0, 4);  // ups[1] as another
    for (int l = 0; l < u.L; ++l) {
        float* skip = ws + ly.off_skip[l];
        const int co = u.dimp[l + 1];
        if (chained && l == u.L - 1) {
            // downs[-1] (2 ResnetBlocks + attention), mid_block1, mid_attn, mid_block2, ups[0] (2 ResnetBlocks + attention): 19 launches in one;
            // with the Downsample conv in front and the Upsample transpose conv behind (chain_edges): 23 in one
            DDK_TRY(run_level_chain(c, 0, cur, skip, bufB, chain_edges ? ws + ly.off_skip[l - 1] : nullptr, chain_edges ? bufA : nullptr));
            cur_c = co;
            break;
        }
        const bool level_chained = chained8 && l == u.L - 2 && cur_ss.n == 1;
        if (level_chained) {
            // downs[-2] on 8x8 maps (2 ResnetBlocks + attention): 7 launches in one
            DDK_TRY(run_level_chain(c, 1, cur, skip, nullptr));
        } else if (l == 0 && fast0) {
            const ResW& r = u.down_res[0];
            if (first_fused) {
                // round 6: the first Block's GroupNorm + Mish + shift inside the conv's launch (the image's 8 tiles exchange their statistics):
                // no raw tensor, no GroupNorm-apply launch
                float* cl = ws + ly.off_cl;
                DDK_TRY(conv_first_gn(x, P + r.c1.wf, r.c1.has_bias ? P + r.c1.b : nullptr, P + r.n1.g, P + r.n1.b, c.temb + r.temb_off,
                                      u.temb_total, c.temb_rows, GN_EPS, a1, B, H, W, r.ci, r.co, GROUPS, cl + cl_counter_floats(B),
                                      reinterpret_cast<unsigned*>(cl), reinterpret_cast<unsigned*>(cl + cl_fail_offset(B)),
                                      step ? step->state : nullptr, step ? t : nullptr, st));
            } else {
                DDK_TRY(conv_first(x, P + r.c1.wf, r.c1.has_bias ? P + r.c1.b : nullptr, raw, gnp, B, H, W, r.ci, r.co, GROUPS,
                                   step ? step->state : nullptr, step ? t : nullptr, st));
                DDK_TRY(groupnorm_mish_parts(raw, gnp, H * W / 128, P + r.n1.g, P + r.n1.b, c.temb + r.temb_off, u.temb_total, nullptr, a1, B,
                                             H * W, r.co, GROUPS, GN_EPS, st, c.temb_rows));
            }
            ddk_conv_args a{};
            a.kind = DDK_CONV3X3_S1;
            a.src0 = a1; a.c0 = r.co;
            a.weight = P + r.c2.w;
            a.weight_wino = P + r.c2.wu;
            a.bias = r.c2.has_bias ? P + r.c2.b : nullptr;
            a.B = B; a.H = H; a.W = W; a.N = r.co;
            const int cl_np = c.allow_cluster && conv_wino_cluster_device_ok() && r.ci <= 8 ? conv_wino_cluster_np(B, H, W, r.co, r.co, GROUPS) : 0;
            if (cl_np > 0 && cl_np <= u.cluster_np_max && conv_wino_variant_new(B, H, W, r.co) && c.n_cluster++ < u.cluster_limit) {
                // round 4: GroupNorm + Mish + the 1x1 res_conv of the input finished in the conv's own launch (in-launch exchange of the
                // tile statistics; the epilogue evaluates the <= 8-channel 1x1 itself): no raw tensor, no GroupNorm-apply launch
                a.out = bufB;
                float* cl = ws + ly.off_cl;
                WinoGnFuse f{P + r.n2.g, P + r.n2.b, nullptr, nullptr, u.temb_total, GN_EPS, GROUPS, cl + cl_counter_floats(B),
                             reinterpret_cast<unsigned*>(cl), reinterpret_cast<unsigned*>(cl + cl_fail_offset(B))};
                f.res_x = x; f.res_w = P + r.res.w; f.res_b = r.res.has_bias ? P + r.res.b : nullptr; f.res_cin = r.ci; f.res_ld = r.res.cin_pad;
                DDK_TRY(conv_forward(a, st, nullptr, &f));
            } else {
                a.out = raw;
                a.gn_partials = gnp;
                a.gn_groups = GROUPS;
                DDK_TRY(conv_forward(a, st));
                DDK_TRY(groupnorm_mish_parts(raw, gnp, conv_wino_stats_parts(B, H, W, r.co, r.co, GROUPS), P + r.n2.g, P + r.n2.b, nullptr,
                                             u.temb_total, nullptr, bufB, B, H * W, r.co, GROUPS, GN_EPS, st, nullptr, x, P + r.res.w,
                                             r.res.has_bias ? P + r.res.b : nullptr, r.ci, r.res.cin_pad));
            }
        } else {
            DDK_TRY(run_res(c, u.down_res[2 * l], cur, cur_c, nullptr, 0, bufB, H, W, cur_ss));
            cur_ss = AddendSlabs();
        }
        if (!level_chained) {
            DDK_TRY(run_res(c, u.down_res[2 * l + 1], bufB, co, nullptr, 0, bufC, H, W));
            DDK_TRY(run_attn(c, u.down_attn[l], bufC, skip, H, W));
        }
        if (l < u.L - 1 && chain_edges && l + 1 == u.L - 1) {
            H /= 2; W /= 2;               // the level chain runs this Downsample conv itself, from `skip`
            cur = skip;
        } else if (l < u.L - 1) {
            const int s2 = conv_splits(DDK_CONV3X3_S2, B, H, W, co, co);
            if (u.fold_down_reduce && s2 > 1 && !(chained && l + 1 == u.L - 1) && !(chained8 && l + 1 == u.L - 2) &&
                res_takes_slab_source(u, u.down_res[2 * l + 2], B, H / 2, W / 2, co)) {
                // the Downsample conv splits k and the ResnetBlock behind it is image-local (8x8 / 4x4 maps, no skip conv): the conv leaves
                // its slabs in the split-K area and that block's two readers sum them -- no reduce launch, no reduced tensor
                ddk_conv_args a{};
                a.kind = DDK_CONV3X3_S2;
                a.src0 = skip; a.c0 = co;
                a.weight = P + u.down_conv[l].w;
                a.out = bufA;                 // unused: the slabs are the result
                a.B = B; a.H = H; a.W = W; a.N = co;
                a.defer_reduce = 1;
                a.workspace = ws + ly.off_splitk;
                a.workspace_bytes = ly.splitk * sizeof(float);
                DDK_TRY(conv_forward(a, st));
                cur_ss.n = s2;
                cur_ss.stride = (long long)B * (H / 2) * (W / 2) * co;
                cur_ss.bias = u.down_conv[l].has_bias ? P + u.down_conv[l].b : nullptr;
                H /= 2; W /= 2;
                cur = ws + ly.off_splitk;
            } else {
                DDK_TRY(run_conv(c, DDK_CONV3X3_S2, u.down_conv[l], skip, co, nullptr, 0, nullptr, bufA, H, W, co));
                H /= 2; W /= 2;
                cur = bufA;
            }
        } else {
            cur = skip;
        }
        cur_c = co;
    }
    if (!chained) {
        DDK_TRY(run_res(c, u.mid1, cur, cur_c, nullptr, 0, bufB, H, W));
        DDK_TRY(run_attn(c, u.mid_attn, bufB, bufC, H, W));
        DDK_TRY(run_res(c, u.mid2, bufC, cur_c, nullptr, 0, bufA, H, W));
        cur = bufA;
    }
    for (int i = 0; i < u.L - 1; ++i) {
        const int lvl = u.L - 1 - i;  // skips.pop(): the most recent skip first (unet.py:97)
        const float* skip = ws + ly.off_skip[lvl];
        const int dout = u.dimp[lvl + 1], din = u.dimp[lvl];
        if (chained8u && i == 1) {
            // ups[1] on 8x8 maps (ResnetBlock on the concat, ResnetBlock, attention): 9 launches in one
            DDK_TRY(run_level_chain(c, 2, cur, const_cast<float*>(skip), bufB));
        } else if (!(chained && i == 0)) {
            DDK_TRY(run_res(c, u.up_res[2 * i], cur, cur_c, skip, dout, bufB, H, W));
            DDK_TRY(run_res(c, u.up_res[2 * i + 1], bufB, din, nullptr, 0, bufC, H, W));
            DDK_TRY(run_attn(c, u.up_attn[i], bufC, bufB, H, W));
        }
        if (!(chain_edges && i == 0))     // (the level chain has written this transpose conv's output into bufA itself)
            DDK_TRY(run_conv(c, DDK_CONVT4X4_S2, u.up_conv[i], bufB, din, nullptr, 0, nullptr, bufA, H, W, din));
        H *= 2; W *= 2;
        cur = bufA;
        cur_c = din;
    }
    // final_conv: Block(dim, dim) then 1x1 to in_ch (unet.py:69-72)
    const int chan = u.generic ? pad32(u.cfg.chan) : u.cfg.chan, n_out = u.cfg.in_ch;
    const int npf = u.final_conv.has_wu ? conv_wino_stats_parts(B, H, W, cur_c, chan, GROUPS) : 0;
    if (npf > 0 && final_tail_ok(H * W, chan, GROUPS, n_out, npf) && (!step || step->per == (long long)H * W * n_out)) {
        // one-pass Winograd conv with statistics, then GroupNorm + Mish + projection (+ the update of x) in ONE launch
        ddk_conv_args a{};
        a.kind = DDK_CONV3X3_S1;
        a.src0 = cur; a.c0 = cur_c;
        a.weight = P + u.final_conv.w;
        a.weight_wino = P + u.final_conv.wu;
        a.bias = u.final_conv.has_bias ? P + u.final_conv.b : nullptr;
        a.out = raw;
        a.B = B; a.H = H; a.W = W; a.N = chan;
        a.gn_partials = gnp;
        a.gn_groups = GROUPS;
        DDK_TRY(conv_forward(a, st));
        if (step)
            return final_tail(raw, gnp, npf, P + u.final_norm.g, P + u.final_norm.b, GN_EPS, P + u.final_w, P + u.final_b, n_out, nullptr,
                              step->x, step->noise, step->noise_step_stride, step->t_first, t, step->c_recip, step->c_recipm1, step->c1,
                              step->c2, step->sigma, step->state, 0, 0, B, H * W, chan, GROUPS, st, dec_counter);
        return final_tail(raw, gnp, npf, P + u.final_norm.g, P + u.final_norm.b, GN_EPS, P + u.final_w, P + u.final_b, n_out, out, nullptr,
                          nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, B, H * W, chan, GROUPS, st);
    }
    DDK_TRY(run_conv_gn(c, u.final_conv, cur, cur_c, nullptr, 0, raw, u.final_norm, nullptr, nullptr, a1, H, W, chan));
    float* eps_hat = step ? step->eps_hat : out;
    DDK_TRY(conv1x1_small_n(a1, P + u.final_w, P + u.final_b, eps_hat, (long long)B * H * W, chan, n_out, st));
    if (!step) return DDK_OK;
    return p_sample_update(step->x, eps_hat, step->noise, step->noise_step_stride, step->t_first, t, step->c_recip, step->c_recipm1,
                           step->c1, step->c2, step->sigma, B, step->per, 0, 0, st, step->state, dec_counter);
}

static int check_shape(const ddk_unet* u, int B, int H, int W) {
    DDK_REQUIRE(u, "unet: null plan");
    DDK_REQUIRE(B > 0 && H > 0 && W > 0, "unet: B/H/W must be positive");
    const int f = 1 << (u->L - 1);
    if (H % f || W % f) {
        set_error("unet: spatial size %dx%d must be divisible by %d (%d stride-2 stages, unet.py:43-50)", H, W, f, u->L - 1);
        return DDK_ERR_ARG;
    }
    return DDK_OK;
}

// sampler bookkeeping kernels ----------------------------------------------------------------------
// per-chain state the captured step reads from memory: [0] step counter, [1] Philox seed, [2] Philox stream id
__global__ void set_chain_state_kernel(int64_t* state, int64_t t_start, uint64_t seed, uint32_t stream_id) {
    state[0] = t_start;
    state[1] = (int64_t)seed;
    state[2] = (int64_t)stream_id;
}

}  // namespace ddk

extern "C" size_t ddk_unet_workspace_bytes(const ddk_unet* u, int B, int H, int W) {
    if (check_shape(u, B, H, W) != DDK_OK) return 0;
    return make_layout(*u, B, H, W).total * sizeof(float);
}

extern "C" int ddk_unet_forward(const ddk_unet* u, const void* packed, const float* x, const int64_t* t, float* out, int B, int H,
                                int W, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_TRY(check_shape(u, B, H, W));
    DDK_REQUIRE(packed && x && t && out && workspace, "unet_forward: null pointer");
    DDK_REQUIRE(aligned16(packed) && aligned16(workspace) && aligned16(x) && aligned16(out), "unet_forward: alignment");
    const Layout ly = make_layout(*u, B, H, W);
    if (workspace_bytes < ly.total * sizeof(float)) {
        set_error("unet_forward: workspace too small (%zu < %zu)", workspace_bytes, ly.total * sizeof(float));
        return DDK_ERR_WORKSPACE;
    }
    float* ws = static_cast<float*>(workspace);
    DDK_TRY(ensure_device_init());
    // the cluster GroupNorm's arrival / departure counters start at zero (they re-arm themselves after every launch)
    DDK_HIP(hipMemsetAsync(ws + ly.off_cl, 0, cl_counter_floats(B) * sizeof(float), as_stream(s)));
    return forward_core(*u, static_cast<const float*>(packed), x, const_cast<int64_t*>(t), out, B, H, W, ws, ly, as_stream(s),
                        u->cluster_gn >= 2);
}

/* The in-launch GroupNorm's workgroups wait for each other; when the device could not host a whole cluster at once (a foreign
 * kernel, a CU mask the library could not see) a wait times out after 20 ms, the tile is written as NaN and the workspace's
 * sticky counter moves.  This call waits for `s`, reads and clears that counter: DDK_ERR_CLUSTER means the tensors produced on
 * `workspace` since the last check are not to be used -- rerun with DDK_OPT_CLUSTER_GROUPNORM = 0. */
extern "C" int ddk_unet_cluster_check(const ddk_unet* u, void* workspace, int B, int H, int W, ddk_stream_t s) {
    DDK_TRY(check_shape(u, B, H, W));
    DDK_REQUIRE(workspace, "unet_cluster_check: null workspace");
    const Layout ly = make_layout(*u, B, H, W);
    unsigned* word = reinterpret_cast<unsigned*>(static_cast<float*>(workspace) + ly.off_cl + cl_fail_offset(B));
    unsigned v = 0;
    DDK_HIP(hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, as_stream(s)));
    DDK_HIP(hipStreamSynchronize(as_stream(s)));
    if (v == 0) return DDK_OK;
    DDK_HIP(hipMemsetAsync(word, 0, sizeof(v), as_stream(s)));
    set_error("in-launch GroupNorm: %u workgroup(s) gave up waiting for their cluster (the GPU is shared, masked or partitioned); "
              "results on this workspace are invalid -- rerun with DDK_OPT_CLUSTER_GROUPNORM = 0", v);
    return DDK_ERR_CLUSTER;
}

extern "C" double ddk_unet_flops(const ddk_unet* u, int B, int H0, int W0) {
    if (check_shape(u, B, H0, W0) != DDK_OK) return 0;
    double f = 0;
    auto res = [&](const ResW& r, int H, int W) {
        f += conv_flops(DDK_CONV3X3_S1, B, H, W, r.ci_real, r.co_real) + conv_flops(DDK_CONV3X3_S1, B, H, W, r.co_real, r.co_real);
        if (r.has_res) f += conv_flops(DDK_CONV1X1, B, H, W, r.ci_real, r.co_real);
        f += 2.0 * B * u->time_dim * r.co_real;  // mlp Linear
    };
    auto attn = [&](const AttnW& a, int H, int W) {
        f += conv_flops(DDK_CONV1X1, B, H, W, a.c_real, 3 * HIDDEN) + conv_flops(DDK_CONV1X1, B, H, W, HIDDEN, a.c_real);
        f += 2.0 * 2.0 * B * HEADS * 32.0 * 32.0 * H * W;  // the two einsums
    };
    int H = H0, W = W0;
    f += 2.0 * B * (u->time_dim * 4.0 * u->time_dim) * 2;  // time_mlp Linears
    for (int l = 0; l < u->L; ++l) {
        res(u->down_res[2 * l], H, W); res(u->down_res[2 * l + 1], H, W); attn(u->down_attn[l], H, W);
        if (l < u->L - 1) { f += conv_flops(DDK_CONV3X3_S2, B, H, W, u->dims[l + 1], u->dims[l + 1]); H /= 2; W /= 2; }
    }
    res(u->mid1, H, W); attn(u->mid_attn, H, W); res(u->mid2, H, W);
    for (int i = 0; i < u->L - 1; ++i) {
        res(u->up_res[2 * i], H, W); res(u->up_res[2 * i + 1], H, W); attn(u->up_attn[i], H, W);
        f += conv_flops(DDK_CONVT4X4_S2, B, H, W, u->up_conv[i].cin, u->up_conv[i].cout);
        H *= 2; W *= 2;
    }
    f += conv_flops(DDK_CONV3X3_S1, B, H, W, u->cfg.chan, u->cfg.chan) + conv_flops(DDK_CONV1X1, B, H, W, u->cfg.chan, u->cfg.in_ch);
    return f;
}

// MFMA / FMA FLOPs the plan's kernels really issue for one forward: the same walk as ddk_unet_flops, but every 3x3 stride-1
// conv is priced by the kernel the plan dispatches it to -- Winograd F(2x2,3x3) forms issue 16 multiplies per 2x2 output tile
// instead of 36 (conv_wino.hip, conv3x3_gn_wlocal_kernel), m tiles are padded to 32 tiles, input channels to 32 -- so that
// executed / time / peak is a fraction of the matrix pipe's peak (<= 1 by construction).
extern "C" double ddk_unet_flops_executed(const ddk_unet* u, int B, int H0, int W0) {
    if (check_shape(u, B, H0, W0) != DDK_OK) return 0;
    double f = 0;
    auto conv3 = [&](const ConvW& cw, int H, int W, int c0, int c1, int N, bool gn) {
        const int cin = c0 + c1;
        if (gn && cw.has_wl && (H * W == 16 || (H * W == 4 && B % 4 == 0)) && conv_gn_local_ok(H, W, cin, c0, N, GROUPS))
            return 2.0 * B * H * W * 9.0 * cin * N;
        if (gn && cw.has_wwl && H * W == 64 && conv_gn_wlocal_ok(H, W, cin, c0, N, GROUPS)) return 2.0 * B * (H * W / 4) * 16.0 * cin * N;
        if (use_wino(cw, H, W, cin, N)) return 2.0 * (double)(ceil_div((long long)B * (H / 2) * (W / 2), 32) * 32) * 16.0 * cin * N;
        return 2.0 * B * H * W * 9.0 * cin * N;
    };
    auto res = [&](const ResW& r, int H, int W, int c0, int c1, bool fast) {
        if (fast) {
            f += 2.0 * B * H * W * (2.0 * ((9 * r.ci + 1) / 2)) * r.co;        // conv_first: K = 9 * C_in rounded up to even
            f += 2.0 * B * H * W * (double)r.ci * r.co;                        // res_conv inside the GroupNorm launch (FMA)
        } else {
            f += conv3(r.c1, H, W, c0, c1, r.co, true);
            if (r.has_res) f += conv_flops(DDK_CONV1X1, B, H, W, c0 + c1, r.co);
        }
        f += conv3(r.c2, H, W, r.co, 0, r.co, true);
    };
    auto attn = [&](const AttnW& a, int H, int W) {
        if (attn_fold_eligible(*u, a, B, H, W)) {
            // k, v projection + context + the per-image fold (T four times per image, A once) + the per-image C x C conv
            f += conv_flops(DDK_CONV1X1, B, H, W, a.c, 2 * HIDDEN) + 2.0 * B * HEADS * 32.0 * 32.0 * H * W;
            f += 2.0 * B * (4.0 * HIDDEN * 32.0 * a.c + (double)a.c * HIDDEN * a.c) + conv_flops(DDK_CONV1X1, B, H, W, a.c, a.c);
            return;
        }
        f += conv_flops(DDK_CONV1X1, B, H, W, a.c, 3 * HIDDEN) + conv_flops(DDK_CONV1X1, B, H, W, HIDDEN, a.c);
        f += 2.0 * 2.0 * B * HEADS * 32.0 * 32.0 * H * W;
    };
    int H = H0, W = W0, cur_c = pad32(u->cfg.in_ch);
    for (int l = 0; l < u->L; ++l) {
        const int co = u->dims[l + 1];
        res(u->down_res[2 * l], H, W, cur_c, 0, l == 0 && first_fast(*u, B, H, W));
        res(u->down_res[2 * l + 1], H, W, co, 0, false);
        attn(u->down_attn[l], H, W);
        if (l < u->L - 1) { f += conv_flops(DDK_CONV3X3_S2, B, H, W, co, co); H /= 2; W /= 2; }
        cur_c = co;
    }
    res(u->mid1, H, W, cur_c, 0, false); attn(u->mid_attn, H, W); res(u->mid2, H, W, cur_c, 0, false);
    for (int i = 0; i < u->L - 1; ++i) {
        const int lvl = u->L - 1 - i, dout = u->dims[lvl + 1], din = u->dims[lvl];
        res(u->up_res[2 * i], H, W, cur_c, dout, false);
        res(u->up_res[2 * i + 1], H, W, din, 0, false);
        attn(u->up_attn[i], H, W);
        if (u->up_conv[i].has_wu && convT_wino_use(B, H, W, din, din))  // Winograd F(2x2, 2x2) per phase: 36 position GEMMs per 2x2 input tile
            f += 2.0 * (double)(ceil_div((long long)B * (H / 2) * (W / 2), 32) * 32) * 36.0 * din * din;
        else
            f += conv_flops(DDK_CONVT4X4_S2, B, H, W, din, din);
        H *= 2; W *= 2;
        cur_c = din;
    }
    f += conv3(u->final_conv, H, W, cur_c, 0, u->cfg.chan, true) + 2.0 * B * H * W * (double)u->cfg.chan * u->cfg.in_ch;
    return f;
}

// ------------------------------------------------------------------------------------------------ sampler
namespace ddk {
struct SamplerLayout {
    size_t unet_floats, off_eps, off_t, off_table, off_tact_all, off_tall, total;
};
// rows = t_start + 1 timesteps: the time-shift table [rows][temb_total], its staging [rows][time_dim] and int64 t = 0..rows-1
static SamplerLayout sampler_layout(const ddk_unet& u, int B, int H, int W, int t_start) {
    SamplerLayout s;
    const size_t rows = (size_t)t_start + 1;
    s.unet_floats = make_layout(u, B, H, W).total;
    s.off_eps = s.unet_floats;
    s.off_t = s.off_eps + al4((size_t)B * H * W * u.cfg.in_ch);
    s.off_table = s.off_t + al4(2 * (size_t)(B + 3));  // int64 t_cur[B] + chain state {counter, seed, stream id}, in float units
    s.off_tact_all = s.off_table + al4(rows * u.temb_total);
    s.off_tall = s.off_tact_all + al4(rows * u.time_dim);
    s.total = s.off_tall + al4(2 * rows);
    return s;
}

__global__ void iota64_kernel(int64_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}
}  // namespace ddk

extern "C" size_t ddk_sampler_workspace_bytes(const ddk_unet* u, int B, int H, int W, int t_start) {
    if (check_shape(u, B, H, W) != DDK_OK || t_start < 0) return 0;
    return sampler_layout(*u, B, H, W, t_start).total * sizeof(float);
}

extern "C" int ddk_sampler_run(const ddk_sampler_args* a, ddk_stream_t s) {
    DDK_REQUIRE(a && a->unet && a->packed && a->x && a->workspace, "sampler: null pointer");
    DDK_REQUIRE(a->c_recip && a->c_recipm1 && a->c1 && a->c2 && a->sigma, "sampler: null schedule table");
    ddk_unet& u = *const_cast<ddk_unet*>(a->unet);     // the graph / table cache is logically mutable state of the plan
    DDK_TRY(check_shape(&u, a->B, a->H, a->W));
    DDK_REQUIRE(a->t_start >= a->t_end && a->t_end >= 0, "sampler: need t_start >= t_end >= 0");
    DDK_REQUIRE(aligned16(a->packed) && aligned16(a->workspace) && aligned16(a->x) && aligned16(a->noise), "sampler: alignment");
    const int B = a->B, H = a->H, W = a->W, C = u.cfg.in_ch;
    const long long per = (long long)H * W * C;
    DDK_REQUIRE(per % 4 == 0, "sampler: H*W*in_ch must be a multiple of 4");
    const SamplerLayout sl = sampler_layout(u, B, H, W, a->t_start);
    if (a->workspace_bytes < sl.total * sizeof(float)) {
        set_error("sampler: workspace too small (%zu < %zu)", a->workspace_bytes, sl.total * sizeof(float));
        return DDK_ERR_WORKSPACE;
    }
    DDK_TRY(ensure_device_init());
    const Layout ly = make_layout(u, B, H, W);
    hipStream_t st = as_stream(s);
    float* ws = static_cast<float*>(a->workspace);
    float* eps_hat = ws + sl.off_eps;
    int64_t* t_cur = reinterpret_cast<int64_t*>(ws + sl.off_t);
    int64_t* state = t_cur + B;                        // [0] step counter, [1] seed, [2] stream id
    const float* P = static_cast<const float*>(a->packed);
    const float* temb_table = ws + sl.off_table;
    const StepArgs step{state, a->x, eps_hat, a->noise, a->noise ? B * per : 0, a->t_start, a->c_recip, a->c_recipm1, a->c1, a->c2,
                        a->sigma, per};

    // one reverse step: bookkeeping (in the forward's first kernel), UNet, update of x (in its last kernel)
    auto one_step = [&]() -> int {
        return forward_core(u, P, a->x, t_cur, nullptr, B, H, W, ws, ly, st, u.cluster_gn >= 1, temb_table, &step);
    };

    auto capture = [&](int steps, hipGraph_t& graph, hipGraphExec_t& exec) -> int {
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            set_error("sampler: hipStreamBeginCapture failed (%s); the legacy NULL stream cannot be captured -- pass a created stream",
                      hipGetErrorString(e));
            return DDK_ERR_HIP;
        }
        int rc = DDK_OK;
        for (int i = 0; i < steps && rc == DDK_OK; ++i) rc = one_step();
        e = hipStreamEndCapture(st, &graph);
        if (rc != DDK_OK) { if (graph) (void)hipGraphDestroy(graph); graph = nullptr; return rc; }
        if (e != hipSuccess) { graph = nullptr; set_error("sampler: hipStreamEndCapture: %s", hipGetErrorString(e)); return DDK_ERR_HIP; }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void)hipGraphDestroy(graph);
            graph = nullptr; exec = nullptr;
            set_error("sampler: hipGraphInstantiate: %s", hipGetErrorString(e));
            return DDK_ERR_HIP;
        }
        return DDK_OK;
    };

    std::lock_guard<std::mutex> lock(u.mu);
    DDK_HIP(hipMemsetAsync(ws + ly.off_cl, 0, cl_counter_floats(B) * sizeof(float), st));    // cluster GroupNorm counters (outside the graph)
    hipLaunchKernelGGL(set_chain_state_kernel, dim3(1), dim3(1), 0, st, state, (int64_t)a->t_start, a->seed, a->stream_id);
    DDK_TRY(check_launch("set_chain_state_kernel"));
    // Time-shift table for t = 0..t_start: the same two kernels a forward runs, once, with "batch" = all timesteps.  It
    // lives in the caller's workspace and stays valid while neither the weights (pack_epoch) nor the workspace change;
    // a caller that rewrites or frees the workspace between calls says so with ddk_sampler_invalidate().
    if (!(u.table.ws == a->workspace && u.table.t_start == a->t_start && u.table.B == B && u.table.H == H && u.table.W == W &&
          u.table.pack_epoch == u.pack_epoch)) {
        const int rows = a->t_start + 1;
        int64_t* t_all = reinterpret_cast<int64_t*>(ws + sl.off_tall);
        float* tact_all = ws + sl.off_tact_all;
        hipLaunchKernelGGL(iota64_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, st, t_all, rows);
        DDK_TRY(check_launch("iota64_kernel"));
        DDK_TRY(time_mlp(t_all, P + u.freqs, P + u.w1t, P + u.b1, P + u.w2t, P + u.b2, tact_all, nullptr, rows, u.time_dim, st));
        DDK_TRY(time_proj(tact_all, P + u.temb_wt, P + u.temb_bias, ws + sl.off_table, rows, u.time_dim, u.temb_total, st));
        u.table.ws = a->workspace; u.table.t_start = a->t_start; u.table.B = B; u.table.H = H; u.table.W = W;
        u.table.pack_epoch = u.pack_epoch;
    }
    const int n_steps = a->t_start - a->t_end + 1;
    if (!a->use_graph || n_steps == 1) {
        for (int k = 0; k < n_steps; ++k) DDK_TRY(one_step());
        return DDK_OK;
    }

    int dev = 0;
    DDK_HIP(hipGetDevice(&dev));
    SamplerGraph* hit = nullptr;
    for (SamplerGraph& g : u.graphs)
        if (g.packed == a->packed && g.x == a->x && g.noise == a->noise && g.ws == a->workspace && g.c_recip == a->c_recip &&
            g.c_recipm1 == a->c_recipm1 && g.c1 == a->c1 && g.c2 == a->c2 && g.sigma == a->sigma && g.B == B && g.H == H &&
            g.W == W && g.t_start == a->t_start && g.device == dev && g.pack_epoch == u.pack_epoch) {
            hit = &g;
            break;
        }
    int first = 0;
    if (!hit) {
        // New buffer set: run the first step eagerly, capture the second, keep the executable graph in the plan.
        if (u.graphs.size() >= 4) {      // bounded cache: retire the least recently used entry (its launches must have drained)
            size_t lru = 0;
            for (size_t i = 1; i < u.graphs.size(); ++i)
                if (u.graphs[i].last_use < u.graphs[lru].last_use) lru = i;
            // wait for the evicted entry's own launches only (not the device: another stream may be mid-capture)
            destroy_entry(u.graphs[lru]);
            u.graphs.erase(u.graphs.begin() + (long)lru);
        }
        DDK_TRY(one_step());
        first = 1;
        SamplerGraph g{};
        g.packed = a->packed; g.x = a->x; g.noise = a->noise; g.ws = a->workspace; g.c_recip = a->c_recip; g.c_recipm1 = a->c_recipm1;
        g.c1 = a->c1; g.c2 = a->c2; g.sigma = a->sigma; g.B = B; g.H = H; g.W = W; g.t_start = a->t_start; g.device = dev;
        g.pack_epoch = u.pack_epoch;
        DDK_TRY(capture(1, g.graph, g.exec));
        u.graphs.push_back(g);
        hit = &u.graphs.back();
    }
    hit->last_use = ++u.use_clock;
    int k = first;
    // long chains: SAMPLER_MULTI steps per graph launch (t and the Philox counter live in device memory, so a graph of any number
    // of steps continues the chain); the one-step graph finishes the remainder
    // (captured with the one-step graph, on the first call for a buffer set whose chain can be that long -- not in a later, timed call)
    // (not for injected noise: a fresh noise tensor per call means a fresh cache entry per call -- parity tests -- and capturing
    //  SAMPLER_MULTI x ~80 launches for a chain that is never replayed is pure overhead)
    if (!hit->exec_multi && !a->noise && (first || n_steps - k >= 2 * SAMPLER_MULTI) && a->t_start + 1 >= 2 * SAMPLER_MULTI)
        DDK_TRY(capture(SAMPLER_MULTI, hit->graph_multi, hit->exec_multi));
    if (hit->exec_multi)
        for (; k + SAMPLER_MULTI <= n_steps; k += SAMPLER_MULTI) {
            const hipError_t e = hipGraphLaunch(hit->exec_multi, st);
            if (e != hipSuccess) { set_error("sampler: hipGraphLaunch: %s", hipGetErrorString(e)); return DDK_ERR_HIP; }
        }
    for (; k < n_steps; ++k) {
        const hipError_t e = hipGraphLaunch(hit->exec, st);
        if (e != hipSuccess) { set_error("sampler: hipGraphLaunch: %s", hipGetErrorString(e)); return DDK_ERR_HIP; }
    }
    if (!hit->done) DDK_HIP(hipEventCreateWithFlags(&hit->done, hipEventDisableTiming));
    DDK_HIP(hipEventRecord(hit->done, st));
    return DDK_OK;
}
