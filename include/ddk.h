/*
 * ddk.h -- C ABI of libddk.so, the MI355X (gfx950) HIP kernels for the DDPM / dDDPM denoising path.
 *
 * The reference (simonamtoft/downsampled-diffusion) has no FFI: the path sits behind Python
 * nn.Modules (models/unet/unet.py:9-104, models/diffusion/ddpm.py:22-457).  This header is the
 * boundary the new host code (downsampled-diffusion_amd/ddk/lib.py, ctypes) binds instead of
 * dispatching ATen ops; every entry cites the reference site whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers (16-byte aligned, dense), sizes are ints; no torch types.
 *   - activations are NHWC fp32 ("pixel-major": [B][H][W][C]); channel counts on the conv/GEMM
 *     kernels are multiples of 32 (pad with ddk_nchw_to_nhwc / ddk_pad_channels).
 *   - every call is asynchronous on `stream` (a hipStream_t), allocates no device memory, never syncs
 *     (the two calls that drop cached hipGraphs, ddk_unet_destroy and ddk_sampler_invalidate, wait for
 *     the device first); the caller owns all buffers including workspaces (sizes from the
 *     *_workspace_bytes calls).
 *   - return 0 on success, negative on error; ddk_last_error() gives the message (thread local).
 */
#ifndef DDK_H
#define DDK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ddk_stream_t; /* hipStream_t */

#define DDK_OK 0
#define DDK_ERR_ARG (-1)     /* shape / alignment / null pointer */
#define DDK_ERR_HIP (-2)     /* a HIP runtime call failed */
#define DDK_ERR_WORKSPACE (-3)
#define DDK_ERR_CLUSTER (-4)  /* ddk_unet_cluster_check: an in-launch GroupNorm exchange timed out; results are invalid */

int ddk_version(void);   /* 400 = 0.4.0; bumped whenever an argument struct or a workspace contract changes */
const char* ddk_last_error(void);
/* 1 when a gfx950 device is visible to this process. */
int ddk_device_ok(void);

/* ------------------------------------------------------------------ layout / packing */
/* NCHW -> NHWC with the channel dim zero-padded to c_pad (>= C).  Unet.forward entry (unet.py:74). */
int ddk_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int c_pad, ddk_stream_t s);
/* NHWC (row stride c_stride >= C) -> NCHW, first C channels. */
int ddk_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int c_stride, ddk_stream_t s);
/* [M][C] -> [M][c_pad], zero fill. */
int ddk_pad_channels(const float* src, float* dst, long long M, int C, int c_pad, ddk_stream_t s);
/* Conv2d weight OIHW -> [O][KH*KW][I_pad] (zero padded input channels). */
int ddk_pack_conv_weight(const float* w_oihw, float* dst, int O, int I, int KH, int KW, int i_pad, ddk_stream_t s);
/* The same with the input channels of TWO concatenated sources padded separately (unet.py:97 at widths that are not multiples of
 * 32): channels [0, split) land at [0, split), channels [split, I) at [split_pad, split_pad + I - split); the rest of a row is zero. */
int ddk_pack_conv_weight_split(const float* w_oihw, float* dst, int O, int I, int KH, int KW, int i_pad, int split, int split_pad,
                               ddk_stream_t s);
/* ConvTranspose2d(k4,s2,p1) weight (I,O,4,4) -> [phase 4][O][tap 4][I]; phase = py*2+px, tap = a*2+b
 * with input offset (dy,dx) = (py - a, px - b) and kernel index ky = 1 - py + 2a, kx = 1 - px + 2b. */
int ddk_pack_convT_weight(const float* w_iohw, float* dst, int I, int O, ddk_stream_t s);
/* ... with both channel counts padded to c_pad (a multiple of 32; zero rows / columns): dst [4][c_pad][4][c_pad] */
int ddk_pack_convT_weight_padded(const float* w_iohw, float* dst, int I, int O, int c_pad, ddk_stream_t s);
/* Linear weight [O][I] -> transposed [I][O] at column offset col0 of a [I][ld] matrix. */
int ddk_pack_linear_T(const float* w_oi, float* dst, int O, int I, int ld, int col0, ddk_stream_t s);

/* ------------------------------------------------------------------ conv family (fp32 MFMA implicit GEMM) */
enum ddk_conv_kind {
    DDK_CONV3X3_S1 = 0, /* Block conv, blocks.py:78 */
    DDK_CONV3X3_S2 = 1, /* Downsample, blocks.py:44 */
    DDK_CONV1X1 = 2,    /* res_conv / to_qkv / to_out, blocks.py:103,123-124 */
    DDK_CONVT4X4_S2 = 3, /* Upsample, blocks.py:35 */
    DDK_CONV4X4_S2 = 4   /* Conv2d k4 s2 p1: the input gradient of Upsample (weights: its (I,O,4,4) tensor read as OIHW) */
};

typedef struct ddk_conv_args {
    int kind;            /* enum ddk_conv_kind */
    const float* src0;   /* NHWC [B][H][W][c0] */
    const float* src1;   /* optional second source (channel concat without materialising it, unet.py:97) */
    int c0, c1;          /* channels of each source, multiples of 32 (c1 = 0 when src1 is NULL) */
    const float* weight; /* packed by ddk_pack_conv_weight / ddk_pack_convT_weight, I = c0 + c1; may be NULL when weight_wino is given
                          * and the launch takes the Winograd path (ddk_conv_wino_splits() > 0, no pre_mish / mish_out / dmish_src) */
    const float* bias;   /* [N] or NULL */
    const float* resid;  /* optional [B][Ho][Wo][N] added in the epilogue (Residual, blocks.py:13-14) */
    float* out;          /* NHWC [B][Ho][Wo][N] */
    int B, H, W;         /* input spatial size */
    int N;               /* output channels, multiple of 32 */
    int pre_mish;        /* 1: apply Mish to the input while staging it (convblocks.py:114) */
    int post_mish;       /* 1: write Mish(out): the next conv's input activation, applied once (convblocks.py:115-117) */
    int defer_reduce;    /* 1: when the launch splits k (ddk_conv_splits() > 1) leave the partial slabs in `workspace`
                          *    (no bias / resid / Mish applied) for ddk_groupnorm_mish_slabs to sum while it loads */
    void* workspace;     /* split-K slabs; may be NULL when ddk_conv_workspace_bytes() == 0 */
    size_t workspace_bytes;
    const float* weight_wino; /* optional (DDK_CONV3X3_S1 only): the same filter packed by ddk_pack_conv_weight_wino.  When given
                               * and the shape is eligible (ddk_conv_wino_splits() > 0) the conv runs as Winograd F(2x2,3x3):
                               * 2.25x fewer MFMA FLOPs, same result up to fp32 summation order.  Its split count (slabs in
                               * `workspace`) is ddk_conv_wino_splits(), workspace = splits * B*H*W*N floats when > 1. */
    float* gn_partials;  /* optional (Winograd path, ddk_conv_gn_partials() > 0, no resid / post_mish): the kernel also writes, per
                          * (128-pixel tile, GroupNorm group), {mean, M2} of its output: B*H*W/128 * gn_groups float pairs, which
                          * ddk_groupnorm_mish_partials turns into GroupNorm+Mish with one read and one write of the tensor */
    int gn_groups;
    float* mish_out;     /* optional: the launch also writes Mish(out) here (same shape as out): the next conv's input activation,
                          * while `out` keeps the pre-activation its backward needs (training path of convblocks.py:112-130) */
    const float* dmish_src; /* optional [B][Ho][Wo][N]: out = (conv + bias) * Mish'(dmish_src) (+ resid): an input-gradient conv that
                             * hands back the gradient of the PRE-activation (replaces a separate Mish-backward launch) */
} ddk_conv_args;

/* Conv2d 3x3 weight OIHW -> Winograd-domain filter U = G g G^T, [I_pad/32][16 positions][O][32] (blocks.py:78). */
int ddk_pack_conv_weight_wino(const float* w_oihw, float* dst, int O, int I, int i_pad, ddk_stream_t s);
/* > 0: the Winograd kernel can emit GroupNorm partials for this shape (the value = 128-pixel tiles per image); 0: it cannot
 * (channel-chunk splits, ragged tiles, groups that straddle a 64-channel tile). */
int ddk_conv_gn_partials(int B, int H, int W, int cin, int N, int groups);
/* 0: shape not eligible for the Winograd kernel (needs even H, W; cin % 32 == 0; N % 64 == 0); else its channel-chunk splits */
int ddk_conv_wino_splits(int B, int H, int W, int cin, int N);
size_t ddk_conv_workspace_bytes(int kind, int B, int H, int W, int cin, int N);
/* number of k-splits (partial slabs) the launch for this shape uses; 1 = written directly */
int ddk_conv_splits(int kind, int B, int H, int W, int cin, int N);
int ddk_conv_forward(const ddk_conv_args* a, ddk_stream_t s);
/* tuning diagnostic (env DDK_DEBUG & 32): per-workgroup {shader cycles, 100 MHz ticks, k-chunks, valid} of the conv
 * k-loop; synchronises the device, copies 6*4096 u64 to host_out and clears the buffer.  Not used by the product. */
int ddk_debug_read_stamps(unsigned long long* host_out);

/* ------------------------------------------------------------------ normalisation / activation */
/* out = Mish(GroupNorm_g(x)) [+ temb[b*temb_stride + c]] [+ addend]   (blocks.py:79-80,106-115)
 * x, addend, out: [B][HW][C]; gamma/beta [C]; stats per (b, group), biased variance, eps inside sqrt. */
int ddk_groupnorm_mish(const float* x, const float* gamma, const float* beta, const float* temb,
                       int temb_stride, const float* addend, float* out, int B, int HW, int C, int groups,
                       float eps, void* workspace, size_t workspace_bytes, ddk_stream_t s);
size_t ddk_groupnorm_workspace_bytes(int B, int HW, int C, int groups);
/* Same, reading the conv output still as `nslab` split-K slabs (stride slab_stride floats) + conv_bias: the
 * conv's reduction is folded into this kernel's load.  Only for slabs that fit the register-resident path
 * (ddk_groupnorm_workspace_bytes() == 0). */
int ddk_groupnorm_mish_slabs(const float* slabs, int nslab, long long slab_stride, const float* conv_bias,
                             const float* gamma, const float* beta, const float* temb, int temb_stride,
                             const float* addend, float* out, int B, int HW, int C, int groups, float eps,
                             ddk_stream_t s);
/* Conv2d(3, padding=1) -> GroupNorm(groups) -> Mish (+ temb[b][c] shift) (+ addend) in ONE launch, for small maps
 * (H*W == 16 or 64): Block of models/unet/blocks.py:75-84 with the ResnetBlock additions of blocks.py:110-115.  The
 * input is src0 (c0 channels) followed by src1 (c1 channels, may be 0/NULL: the concat of unet.py:97), NHWC; `weight` is
 * the filter packed by ddk_pack_conv_weight_local (O*9*i_pad floats: the kernel's MFMA operand order, O % 32 == 0).
 * ddk_conv3x3_gn_mish_ok() != 0 when the shape is eligible. */
int ddk_pack_conv_weight_local(const float* w_oihw, float* dst, int O, int I, int i_pad, ddk_stream_t s);
/* the same operand order for a 1x1 filter [O][I][1][1] (one tap: O*i_pad floats) -- the 1x1 ops of the level chain (DDK_OPT_LEVEL_CHAIN) */
int ddk_pack_conv1x1_weight_local(const float* w_oi, float* dst, int O, int I, int i_pad, ddk_stream_t s);
/* ... and for a ConvTranspose2d(4, stride 2, padding 1) filter [I][O][4][4] (O*16*I floats: four output phases of 2 x 2 taps each) -- the
 * transpose-conv op that ends the level chain (reference models/unet/blocks.py:32-38) */
int ddk_pack_convT_weight_local(const float* w_iohw, float* dst, int I, int O, ddk_stream_t s);
int ddk_conv3x3_gn_mish_ok(int H, int W, int cin, int c0, int N, int groups);
int ddk_conv3x3_gn_mish(const float* src0, int c0, const float* src1, int c1, const float* weight, const float* bias,
                        const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                        float* out, int B, int H, int W, int N, int groups, float eps, ddk_stream_t s);
/* GroupNorm+Mish(+temb)(+addend) of x [B][HW][C] from the {mean, M2} partials a conv left in ddk_conv_args.gn_partials
 * (tiles_per_image = ddk_conv_gn_partials()): statistics merged in fixed order, x read once (blocks.py:78-80). */
int ddk_groupnorm_mish_partials(const float* x, const float* partials, int tiles_per_image, const float* gamma, const float* beta,
                                const float* temb, int temb_stride, const float* addend, float* out, int B, int HW, int C,
                                int groups, float eps, ddk_stream_t s);
/* The network's first conv (unet.py:43-50, blocks.py:78): Conv2d(C_in, N, 3, padding=1) with 1 <= C_in <= 8 on the UNPADDED NHWC
 * input x [B][H][W][C_in]; K = 9 * C_in exactly.  w_first = ddk_pack_conv_weight_first(OIHW weight) ((N/32) * ceil(9 C_in / 2) * 64
 * floats); gn_partials optional: per (128-pixel tile, group) {mean, M2}, B*H*W/128 * groups float pairs, for
 * ddk_groupnorm_mish_partials.  Eligible shapes: ddk_conv_first_ok() != 0 (N % 32 == 0, N <= 256, H*W % 128 == 0). */
int ddk_pack_conv_weight_first(const float* w_oihw, float* dst, int O, int I, ddk_stream_t s);
int ddk_conv_first_ok(int cin, int N, int H, int W, int groups);
int ddk_conv_first(const float* x, const float* w_first, const float* bias, float* out, float* gn_partials, int B, int H, int W,
                   int cin, int N, int groups, ddk_stream_t s);
/* 1x1 conv with exactly 128 input channels on M = B*H*W >= 2048 pixels (M % 64 == 0, N % 128 == 0) as a weights-stationary
 * GEMM: a workgroup keeps a 128-output-channel slice of the weight in LDS and streams 64-pixel tiles of x through it
 * (replaces the nn.Conv2d(k=1) dispatches of reference models/unet/blocks.py:103,123,124 on the 32x32 / 16x16 maps; ddk_conv_forward
 * takes this path by itself when the shape is eligible).  x [M][128]; w [N][128] = ddk_pack_conv_weight(kind 1x1); bias [N] or null;
 * resid [M][N] or null (added to the output); ln_c1 != null: the channel LayerNorm of blocks.py:57-60 is folded in -- w must then
 * hold W o g and (ln_c1, ln_c2) = (W g, W b) per output channel (what ddk_unet_pack derives). */
int ddk_conv1x1_ws_ok(long long M, int K, int N);
int ddk_conv1x1_ws(const float* x, const float* w, const float* bias, const float* resid, float* out, long long M, int N,
                   const float* ln_c1, const float* ln_c2, float ln_eps, ddk_stream_t s);
/* The same with PER-IMAGE weights: w [images][128][128], bias / ln_c1 / ln_c2 [images][128], N == 128, M / images pixels per image
 * (a multiple of 64).  With ddk_attention_fold this evaluates a whole attention block's output on maps with HW >> C. */
int ddk_conv1x1_ws_images(const float* x, const float* w, const float* bias, const float* resid, float* out, long long M, int N,
                          const float* ln_c1, const float* ln_c2, float ln_eps, int images, ddk_stream_t s);
/* ddk_groupnorm_mish_partials whose addend is a 1x1 conv of a narrow tensor, evaluated on the fly (the first ResnetBlock's
 * res_conv, blocks.py:103,115): out = Mish(GN(x)) [+ temb] + (res_b[c] + sum_k res_x[pix][k] res_w[c][k]), res_x [B*HW][res_cin],
 * res_w [C][res_cin] (the OIHW 1x1 weight as is), 1 <= res_cin <= 8; C/4 must divide 256. */
int ddk_groupnorm_mish_partials_res1x1(const float* x, const float* partials, int tiles_per_image, const float* gamma, const float* beta,
                                       const float* temb, int temb_stride, const float* res_x, const float* res_w, const float* res_b,
                                       int res_cin, float* out, int B, int HW, int C, int groups, float eps, ddk_stream_t s);
/* The same Block in one launch for 64-pixel maps (8x8; H, W even), as Winograd F(2x2,3x3) inside an image-local tiling.
 * `weight`: ddk_pack_conv_weight_wino_local (O*16*i_pad floats, O % 32 == 0); cin <= 320. */
int ddk_pack_conv_weight_wino_local(const float* w_oihw, float* dst, int O, int I, int i_pad, ddk_stream_t s);
int ddk_conv3x3_gn_mish_wino_ok(int H, int W, int cin, int c0, int N, int groups);
int ddk_conv3x3_gn_mish_wino(const float* src0, int c0, const float* src1, int c1, const float* weight, const float* bias,
                             const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                             float* out, int B, int H, int W, int N, int groups, float eps, ddk_stream_t s);
/* The two one-launch Blocks above (H*W == 16: `weight` = ddk_pack_conv_weight_local; H*W == 64: ddk_pack_conv_weight_wino_local) reading
 * operands that are still in split-K form: src_slabs > 1 -- `src` is that many partial slabs of the input, src_stride floats apart
 * (what a conv left with ddk_conv_args.defer_reduce), summed in slab order + src_bias[c] while the image is staged; addend_slabs > 1 -- the
 * same for the residual.  Sum order and arithmetic are the split-K reduce's, so the result equals the Block of the reduced tensors bit
 * for bit.  How the UNet plan drops the reduce launch behind a Downsample conv (blocks.py:41-47, DDK_OPT_FOLD_DOWNSAMPLE_REDUCE). */
int ddk_conv3x3_gn_mish_slabs(const float* src, int src_slabs, long long src_stride, const float* src_bias, int c0, const float* weight,
                              const float* bias, const float* gamma, const float* beta, const float* temb, int temb_stride,
                              const float* addend, int addend_slabs, long long addend_stride, const float* addend_bias, float* out, int B,
                              int H, int W, int N, int groups, float eps, ddk_stream_t s);
/* The same Block in ONE launch on maps whose images span several 128-pixel tiles (32x32, 16x16): the Winograd conv's workgroups
 * of one image exchange their tile statistics through `workspace` and finish GroupNorm + Mish (+ temb[b][c]) (+ addend) on their own
 * tile in registers (blocks.py:75-84,110-115).  Eligible when ddk_conv3x3_gn_mish_cluster_ok() > 0: one-pass Winograd shape
 * (ddk_conv_gn_partials() > 0), <= 8 tiles per image, whole clusters per dispatch round.  weight_wino: ddk_pack_conv_weight_wino. */
/* The exchange assumes the cluster's workgroups are resident together (see DDK_OPT_CLUSTER_GROUPNORM); _ok() is 0 on a device where
 * that is implausible, and after the call ddk_conv3x3_gn_mish_cluster_check(workspace, B, s) -- which waits for `s` -- says whether a
 * wait timed out (DDK_ERR_CLUSTER: the output has NaN tiles). */
int ddk_conv3x3_gn_mish_cluster_ok(int B, int H, int W, int cin, int N, int groups);
int ddk_conv3x3_gn_mish_cluster_check(void* workspace, int B, ddk_stream_t s);
size_t ddk_conv3x3_gn_mish_cluster_workspace_bytes(int B, int H, int W, int N);
/* _ok() also admits the shapes whose channel chunks the Winograd conv splits over 2-4 workgroups (ddk_conv_wino_splits() > 1; 16x16 maps
 * of 64-channel tiles at batch 32) when all of them make one dispatch round (<= 256): a tile's first workgroup sums its partners'
 * partial tiles inside the launch, then takes part in the exchange -- no slabs for a GroupNorm launch to sum.  Such a shape needs a larger
 * workspace (pair counters + splits - 1 slabs): _split_workspace_bytes() gives the whole size, or 0 for a shape of the plain kind. */
size_t ddk_conv3x3_gn_mish_cluster_split_workspace_bytes(int B, int H, int W, int cin, int N, int groups);
int ddk_conv3x3_gn_mish_cluster(const float* src0, int c0, const float* src1, int c1, const float* weight_wino, const float* bias,
                                const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                                float* out, int B, int H, int W, int N, int groups, float eps, void* workspace, size_t workspace_bytes,
                                ddk_stream_t s);
/* per-pixel channel LayerNorm, (x-mean)/(sqrt(var)+eps)*g+b, biased var (blocks.py:57-60). */
int ddk_chan_layernorm(const float* x, const float* g, const float* b, float* out, long long M, int C,
                       float eps, ddk_stream_t s);
/* elementwise Mish / tanh (convblocks.py:110, dddpm.py:99,110). */
int ddk_mish(const float* x, float* out, long long n, ddk_stream_t s);
int ddk_tanh(const float* x, float* out, long long n, ddk_stream_t s);
/* out[b][y][x][c] = mean of the 2x2 window (F.avg_pool2d(2), convblocks.py:129). */
int ddk_avgpool2(const float* x, float* out, int B, int H, int W, int C, ddk_stream_t s);
/* nearest-neighbour x2 (F.interpolate(scale_factor=2), convblocks.py:127). */
int ddk_upsample_nearest2(const float* x, float* out, int B, int H, int W, int C, ddk_stream_t s);
/* out = a + b */
int ddk_add(const float* a, const float* b, float* out, long long n, ddk_stream_t s);

/* ------------------------------------------------------------------ linear attention (blocks.py:126-134) */
/* qkv: [B][HW][3*heads*32], channel = (qkv, head, c).  ctx[b][h][d][e] = sum_n softmax_n(k[d,:])[n] v[e,n].
   The pixel range is split over workgroups when a workspace of ddk_linattn_context_workspace_bytes() is given
   (partials merged in split order: deterministic); workspace == NULL runs one workgroup per (b, head). */
size_t ddk_linattn_context_workspace_bytes(int B, int HW, int heads);
int ddk_linattn_context(const float* qkv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* Both steps in one launch for maps with HW <= 64 (also writes ctx, which the backward needs). */
/* ddk_linattn_context on rows [k | v] (2 * heads * 32 floats per pixel) instead of [q | k | v] */
int ddk_linattn_context_kv(const float* kv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* Folded attention output (blocks.py:126-134 + to_out :124 + the PreNorm LayerNorm :57-60; C = heads * 32 = 128): q is linear in this
 * attention, so y = to_out(ctx^T q) + b collapses to a per-image matrix applied to LayerNorm(x):
 *   A[b] = W_out . blockdiag(ctx[b]^T) . (W_q o g)   [128][128],   a1[b] = W_out ctx^T (W_q g),   a2[b] = W_out ctx^T (W_q beta) + b_out
 * so that y = r (A x) - r mean a1 + a2 with r = 1 / (std + eps) per pixel -- ddk_conv1x1_ws_images(x, A, NULL, resid = x, ..., a1, a2).
 * ctx [B][4][32][32] from ddk_linattn_context(_kv); wqg = rows 0..127 of the LayerNorm-folded to_qkv weight [384][128], c1q / c2q the
 * first 128 entries of its fold vectors; wout [128][128] = ddk_pack_conv_weight(to_out), bout [128] or NULL. */
int ddk_attention_fold(const float* ctx, const float* wqg, const float* c1q, const float* c2q, const float* wout, const float* bout, float* A,
                       float* a1, float* a2, int B, int C, int heads, ddk_stream_t s);
/* The k and v thirds of to_qkv (PreNorm LayerNorm folded in) + the context of the block in ONE launch, for C = 128, 4 heads, H*W a multiple
 * of 64 (>= 256): no [M][256] kv tensor is written (blocks.py:57-60, 123, 129-131).  x [B*HW][128]; w_kv [256][128] = rows 128..383 of the
 * folded weight W o g (k rows, then v rows); c1 / c2 [256] = W g / W b of those rows; ctx [B][4][32][32] as ddk_linattn_context_kv writes it. */
size_t ddk_attention_kv_context_workspace_bytes(int B, int HW);
int ddk_attention_kv_context_ok(int B, int HW, int C, int heads);
int ddk_attention_kv_context(const float* x, const float* w_kv, const float* c1, const float* c2, float ln_eps, float* ctx, int B, int HW,
                             void* workspace, size_t workspace_bytes, ddk_stream_t s);
int ddk_linattn_fused_small(const float* qkv, float* ctx, float* out, int B, int HW, int heads, ddk_stream_t s);
/* out[b][n][h*32+e] = sum_d ctx[b][h][d][e] * q[b][n][h*32+d]. */
int ddk_linattn_apply(const float* qkv, const float* ctx, float* out, int B, int HW, int heads, ddk_stream_t s);
/* Small maps (H*W <= 64): to_qkv with the channel LayerNorm folded in (blocks.py:57-60, 123) + context + apply for one (image,
 * head) per workgroup -- no qkv tensor.  x [B][HW][C]; w_operand = ddk_pack_qkv_operand(folded_w [3*heads*32][c_pad] = W o g);
 * c1 = W g, c2 = W b per output column; ln_eps is added to the std.  out [B][HW][heads*32], ctx [B][heads][32][32]. */
int ddk_pack_qkv_operand(const float* folded_w, float* dst, int heads, int c_pad, ddk_stream_t s);
int ddk_linattn_small_from_x(const float* x, const float* w_operand, const float* c1, const float* c2, float ln_eps, float* ctx,
                             float* out, int B, int HW, int C, int heads, ddk_stream_t s);

/* ------------------------------------------------------------------ time embedding (blocks.py:22-29, unet.py:30-35, blocks.py:92-95) */
/* act[b][:] = Mish(Linear2(Mish(Linear1(sincos(t[b] * freqs)))))  -- the vector every ResnetBlock's
 * Linear consumes.  w1t [dim][4dim], w2t [4dim][dim] are transposed Linear weights; scratch none. */
int ddk_time_mlp(const int64_t* t, const float* freqs, const float* w1t, const float* b1, const float* w2t,
                 const float* b2, float* act, float* raw, int B, int dim, ddk_stream_t s);
/* out[b][j] = sum_k act[b][k] * wt[k][j] + bias[j], j < n_out (all 17 ResnetBlock mlps at once). */
int ddk_time_proj(const float* act, const float* wt, const float* bias, float* out, int B, int dim, int n_out,
                  ddk_stream_t s);

/* ------------------------------------------------------------------ small-N 1x1 (final_conv.1, unet.py:71) */
/* out[m][co] = sum_c x[m][c] w[co][c] + b[co]; x [M][C], w [n_out][C] (the OIHW tensor as is), out [M][n_out]. */
int ddk_conv1x1_small_n(const float* x, const float* w, const float* bias, float* out, long long M, int C,
                        int n_out, ddk_stream_t s);

/* ------------------------------------------------------------------ noise-schedule arithmetic */
/* x_t = sqrt_acp[t_b] * x + sqrt_1m_acp[t_b] * eps     (ddpm.py:256-273); per = elements per sample. */
int ddk_q_sample(const float* x, const float* eps, const int64_t* t, const float* sqrt_acp,
                 const float* sqrt_1m_acp, float* out, int B, long long per, ddk_stream_t s);
/* One reverse step after the UNet call (ddpm.py:149-158,177-185,216-227), in place on x:
 *   x0 = clamp(c_recip[t] x - c_recipm1[t] eps_hat, -1, 1); mean = c1[t] x0 + c2[t] x;
 *   x <- mean + [t > 0] sigma[t] z,  sigma = exp(0.5 posterior_log_variance_clipped).
 * z comes from `noise` when non-NULL, else from Philox4x32-10 keyed (seed, t_b, stream_id). */
int ddk_p_sample_update(float* x, const float* eps_hat, const float* noise, const int64_t* t,
                        const float* c_recip, const float* c_recipm1, const float* c1, const float* c2,
                        const float* sigma, int B, long long per, uint64_t seed, uint32_t stream_id,
                        ddk_stream_t s);
/* The end of a forward in one launch (unet.py:69-72 behind the final Block's conv; ddpm.py:203-227): GroupNorm from the conv's
 * partials -> Mish -> 1x1 projection to n_out <= 8 channels (w [n_out][C], bias [n_out]) -> eps_hat; eps_out and / or x may be
 * given: eps_out [B][HW][n_out] receives eps_hat, x [B][HW][n_out] gets the reverse-step update of ddk_p_sample_update in place
 * (bit-identical to it given the same eps_hat).  C in {32, 64, 128, 256}, HW == tiles_per_image * 128. */
int ddk_final_tail(const float* raw, const float* partials, int tiles_per_image, const float* gamma, const float* beta, float eps,
                   const float* w, const float* bias, int n_out, float* eps_out, float* x, const float* noise, const int64_t* t,
                   const float* c_recip, const float* c_recipm1, const float* c1, const float* c2, const float* sigma, uint64_t seed,
                   uint32_t stream_id, int B, int HW, int C, int groups, ddk_stream_t s);
/* out[i] ~ N(0,1): Philox4x32-10 + Box-Muller, counter (i/4, step, stream_id)  (ddpm.py:241). */
int ddk_randn(float* out, long long n, uint64_t seed, uint32_t step, uint32_t stream_id, ddk_stream_t s);
/* Sampler output stage (utils/eval_helpers.py:37-41): per-image min-max over C*H*W, x255, NCHW -> NHWC:
   out[b][h][w][c] = ((x[b][c][h][w] - min_b) / (max_b - min_b)) * 255, bit-identical to the reference expression. */
int ddk_fix_samples(const float* x_nchw, float* out_nhwc, int B, int C, int H, int W, ddk_stream_t s);
/* Evaluation-time variational bound, per sample (ddpm.py:317-366 after the UNet call; models/utils/losses.py:17-109;
 * utils/utils.py:43-48 flat_bits): vlb[b] = mean over the sample of { t_b > 0: KL(q(x_{t-1}|x_t,x) || p(x_{t-1}|x_t));
 * t_b == 0: discretised-Gaussian NLL of x } / ln 2, with p's mean from eps_hat (x0 clamped to [-1,1]) and both
 * variances = exp(post_logvar[t_b]).  eps / sqerr optional (both or neither): sqerr[b] = sum (eps - eps_hat)^2, the
 * L_simple term of test_losses_ (ddpm.py:424-426).  Any layout, as long as the five tensors agree.
 * Every sample is split over several workgroups (ddk_vlb_terms_workspace_bytes of scratch); the partial sums are added in slice
 * order by whichever workgroup finishes last, so the result does not depend on the arrival order. */
size_t ddk_vlb_terms_workspace_bytes(int B, long long per);
int ddk_vlb_terms(const float* x, const float* x_t, const float* eps_hat, const float* eps, const int64_t* t,
                  const float* c_recip, const float* c_recipm1, const float* c1, const float* c2,
                  const float* post_logvar, float* vlb, float* sqerr, int B, long long per, void* workspace,
                  size_t workspace_bytes, ddk_stream_t s);
/* per_sample[b] = sum_i (a - b)^2 over the sample's `per` elements (ddpm.py:279, utils/utils.py:34-40). */
int ddk_sq_err_sum(const float* a, const float* b, float* per_sample, int B, long long per, ddk_stream_t s);

/* ------------------------------------------------------------------ whole-UNet plan (unet.py:74-104, eval mode) */
typedef struct ddk_unet_config {
    int in_ch;      /* config['unet_in'] */
    int chan;       /* config['unet_chan'], multiple of 32 */
    int n_levels;   /* len(config['unet_dims']) */
    int mults[8];   /* config['unet_dims'] */
} ddk_unet_config;

typedef struct ddk_unet ddk_unet; /* opaque */

ddk_unet* ddk_unet_create(const ddk_unet_config* cfg);
void ddk_unet_destroy(ddk_unet* u);
/* Weight slots: the plan lists every state_dict tensor it needs (name = reference key, Appendix B of
 * SURVEY.md; the synthetic slot "@sinusoidal_freqs" is the fp32 table of blocks.py:24-26). */
int ddk_unet_num_slots(const ddk_unet* u);
const char* ddk_unet_slot_name(const ddk_unet* u, int slot);
long long ddk_unet_slot_numel(const ddk_unet* u, int slot);
/* bytes of the packed weight arena the caller allocates once per model */
size_t ddk_unet_packed_bytes(const ddk_unet* u);
/* repack one canonical (state_dict layout) tensor into the arena */
int ddk_unet_pack_slot(const ddk_unet* u, int slot, const float* canonical, void* packed, ddk_stream_t s);
/* Weights derived from the packed slots (LayerNorm folded into to_qkv: W o g, W g, W b).  ddk_unet_pack_slot re-derives a
   site's weights whenever one of its source slots (to_qkv.weight, norm.g, norm.b) is packed, so packing every slot -- in
   any order -- is enough; this call re-derives all sites explicitly (e.g. after writing into `packed` by other means). */
int ddk_unet_finalize_pack(const ddk_unet* u, void* packed, ddk_stream_t s);
size_t ddk_unet_workspace_bytes(const ddk_unet* u, int B, int H, int W);
/* eps_hat = Unet(x, t).  x: NHWC [B][H][W][in_ch] (unpadded), out same shape. */
int ddk_unet_forward(const ddk_unet* u, const void* packed, const float* x, const int64_t* t, float* out,
                     int B, int H, int W, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* Plan options.  DDK_OPT_CLUSTER_GROUPNORM (default 1): where a Block's 3x3 conv runs as a one-pass Winograd launch whose
 * workgroups of one image are co-resident, GroupNorm + Mish + shift + residual finish INSIDE that launch (the workgroups exchange
 * their tile statistics through the workspace); 0 keeps the conv + GroupNorm-apply pair.  1 = in ddk_sampler_run only, 2 = in
 * ddk_unet_forward as well.  Co-residency is an assumption (whole MI355X: 256 CUs, 8 XCDs, no CU mask -- checked; nothing else
 * running on the GPU -- not checkable), so the wait is bounded (20 ms) and a give-up is loud: the tile becomes NaN and
 * ddk_unet_cluster_check() returns DDK_ERR_CLUSTER.  EVERY caller that leaves the option on must call ddk_unet_cluster_check at
 * its sync point (the end of a chain) and rerun with the option off on error.  Changing it drops cached sampler graphs. */
#define DDK_OPT_CLUSTER_GROUPNORM 1
/* DDK_OPT_ATTENTION_FOLD (default 1): on maps with more than 256 pixels and 128 channels the attention block's q projection, apply and
 * to_out run as ONE 1x1 conv of x with a per-image 128x128 matrix W_out . ctx^T . W_q (q is linear in this attention; the
 * PreNorm LayerNorm is folded in as well); 0 keeps to_qkv / context / apply / to_out.  Same result up to fp32 summation order. */
#define DDK_OPT_ATTENTION_FOLD 3
/* DDK_OPT_FOLD_DOWNSAMPLE_REDUCE (default 0: measured slower on MI355X at batch 32 -- every one of an image's eight workgroups re-sums the
 * slabs; kept as an option and as ddk_conv3x3_gn_mish_slabs): where a Downsample conv (blocks.py:41-47) splits its contraction and the ResnetBlock
 * behind it runs on the image-local kernels (8x8 / 4x4 maps, no skip conv), the conv leaves its split-K slabs and that block's two
 * readers -- the first Block's staging loop, the second Block's residual -- sum them in slab order (+ bias): no reduce launch.
 * 0 keeps the reduce launch; bit-identical results either way (tests/test_step_edges_gpu.py). */
#define DDK_OPT_FOLD_DOWNSAMPLE_REDUCE 5
/* DDK_OPT_ATTENTION_KV_CONTEXT (default 1): inside the folded attention block the k, v projection and the context run as one launch
 * (ddk_attention_kv_context) instead of a 1x1 conv that writes the kv tensor + the context kernel that reads it back. */
#define DDK_OPT_ATTENTION_KV_CONTEXT 6
/* DDK_OPT_LEVEL_CHAIN (default 1): where the last level of the UNet is a 4x4 map of 256 channels (unet_chan 128, dims (1,2,2,2) on 32x32
 * inputs), its 19 launches -- the ResnetBlocks and attention blocks of downs[-1], mid and ups[0], reference models/unet/unet.py:83-101 --
 * run as ONE persistent launch whose workgroups hand the 16-pixel images to each other in memory (csrc/level_chain.hip).  Like the
 * in-launch GroupNorm it needs all its workgroups resident together, so it runs exactly where DDK_OPT_CLUSTER_GROUPNORM lets that one
 * run (ddk_sampler_run; ddk_unet_forward with that option at 2), and a wait that times out moves the same sticky word
 * (ddk_unet_cluster_check -> DDK_ERR_CLUSTER).  Values: 0 off, 1 the 4x4 level (default), 3 also the two 8x8 levels (downs[-2], ups[1]: one
 * launch each; built and tested, measured slower on MI355X at batch 32 -- a hop moves 64 KB per image and workgroup there), 4 only those. */
#define DDK_OPT_LEVEL_CHAIN 7
/* DDK_OPT_FIRST_GROUPNORM (default 1): the network's first Block (conv on the <= 8-channel input, reference models/unet/blocks.py:74-84)
 * finishes its GroupNorm + Mish + time shift inside the conv's own launch -- the eight 128-pixel tiles of an image exchange their statistics
 * as the Winograd convs of DDK_OPT_CLUSTER_GROUPNORM do -- wherever that option lets the exchange run; 0 keeps conv + GroupNorm-apply. */
#define DDK_OPT_FIRST_GROUPNORM 8
int ddk_unet_set_option(ddk_unet* u, int option, int value);
/* Waits for `s`, then reads and clears the sticky give-up count of the launches issued on `workspace` (a ddk_unet_forward or
 * ddk_sampler_run workspace of this shape): DDK_OK, or DDK_ERR_CLUSTER when any in-launch GroupNorm exchange timed out. */
int ddk_unet_cluster_check(const ddk_unet* u, void* workspace, int B, int H, int W, ddk_stream_t s);
/* workgroups that ever gave up waiting for their cluster in this process (0 unless the GPU could not host a whole cluster) */
unsigned ddk_debug_cluster_timeouts(void);
/* Measurement support: `workgroups` records of 4 x uint64 {XCC id, shader-cycle counter, 100 MHz counter, 1} into `out`; two
 * probes around a region give the shader clock held in it (per XCC: d cycles / d ticks x 100 MHz). */
int ddk_debug_clock_probe(unsigned long long* out, int workgroups, ddk_stream_t s);
/* Test support: occupies the device with `workgroups` workgroups of 256 threads holding `lds_bytes` of LDS each for about
 * `microseconds` (a spin on the 100 MHz clock; bounded), on stream `s` -- the "foreign kernel" of the cluster tests. */
int ddk_debug_occupy(int workgroups, int lds_bytes, int microseconds, ddk_stream_t s);
/* FLOPs (2*MAC) of one forward for B samples at HxW: the algorithmic work bench.py prices. */
double ddk_unet_flops(const ddk_unet* u, int B, int H, int W);
/* FLOPs the plan's kernels really issue for that forward: 3x3 convs dispatched to a Winograd F(2x2,3x3) kernel count 16/36 of
 * their direct multiplies, tiles / channels padded as launched.  executed / time / peak is a fraction of the MFMA peak. */
double ddk_unet_flops_executed(const ddk_unet* u, int B, int H, int W);

/* ------------------------------------------------------------------ T-step sampler (ddpm.py:229-249) */
typedef struct ddk_sampler_args {
    const ddk_unet* unet;
    const void* packed;
    float* x;                 /* NHWC [B][H][W][in_ch]: x_T in, x_{t_end} out */
    const float* noise;       /* NULL -> in-kernel Philox; else [n_steps][B][H][W][in_ch], k-th draw for step k */
    const float* c_recip;     /* sqrt_recip_alphas_cumprod [T] */
    const float* c_recipm1;   /* sqrt_recipm1_alphas_cumprod [T] */
    const float* c1;          /* posterior_mean_coef1 [T] */
    const float* c2;          /* posterior_mean_coef2 [T] */
    const float* sigma;       /* exp(0.5 * posterior_log_variance_clipped) [T] */
    int B, H, W;
    int t_start;              /* first timestep (T-1) */
    int t_end;                /* last timestep inclusive (0, or early_stop) */
    uint64_t seed;
    uint32_t stream_id;       /* rank / shard id: independent Philox stream per GPU */
    int use_graph;            /* capture one step into a hipGraph and replay it */
    void* workspace;
    size_t workspace_bytes;
} ddk_sampler_args;

/* workspace of ddk_sampler_run for chains starting at t_start (it holds, besides the UNet's scratch, the per-block time
   shifts of every timestep 0..t_start, computed once: no time-embedding kernel runs inside the loop) */
size_t ddk_sampler_workspace_bytes(const ddk_unet* u, int B, int H, int W, int t_start);
/* Runs steps t_start .. t_end.  With use_graph the plan keeps the captured step (a hipGraphExec_t) and the per-timestep
 * shift table it computed in `workspace`, keyed by the buffer pointers, shape, t_start and the plan's weight epoch: the
 * first call on a buffer set runs one step eagerly, captures and instantiates (a one-step graph and, for chains that can be
 * long, a 16-step graph); every later call on the same buffers only writes {t_start, seed, stream_id} into the workspace and
 * issues one hipGraphLaunch per 16 steps plus one per remaining step (seed, stream id and t are read from device memory by
 * the kernels, so they are not part of the key).  At most 4 buffer sets are cached (LRU).
 * Contract: a caller that frees or overwrites `workspace` (or frees any buffer passed here) between calls must call
 * ddk_sampler_invalidate() first. */
int ddk_sampler_run(const ddk_sampler_args* a, ddk_stream_t s);
/* Drops the plan's cached sampler graphs and shift table (waits for the device when graphs exist). */
int ddk_sampler_invalidate(ddk_unet* u);
/* Drops only the cached graphs (and shift table) that live in / point into `workspace`, after waiting for the launches of
 * those graphs alone; what was captured on other workspaces stays cached.  Call it before freeing or reusing one workspace. */
int ddk_sampler_release_workspace(ddk_unet* u, const void* workspace);


/* ================================================================== training path (backward kernels) ==
 * Autograd counterparts of the ops above, driven by objective.backward() in trainers/trainer_ddpm.py:124-128.
 * Parameter gradients ACCUMULATE into the caller's (canonical-layout) gradient tensors. */

/* Winograd-domain filter of the INPUT-gradient conv of a 3x3 stride-1 conv, for its input channels [c_lo, c_hi): the 3x3 conv of
 * dY with g'[n][c][a][b] = w[c][c_lo + n][2-a][2-b] -> [o_pad/32][16][c_hi - c_lo][32] (o_pad = O rounded up to 32), the layout
 * ddk_conv_args.weight_wino takes; w is the forward OIHW tensor (O, I, 3, 3).  Flip, transpose and pack in one kernel. */
int ddk_pack_conv_weight_wino_dgrad(const float* w_oihw, float* dst, int O, int I, int c_lo, int c_hi, int o_pad, ddk_stream_t s);
/* ConvTranspose2d(C, N, 4, stride 2, padding 1) weight (I, O, 4, 4) (blocks.py:35) -> its Winograd F(2x2, 2x2) form, one 3x3 set of
 * position filters per output phase: dst[4][i_pad / 32][9][O][32] (36 * O * i_pad floats).  Passed as ddk_conv_args.weight_wino with
 * kind DDK_CONVT4X4_S2 (c1 == 0, no resid / Mish, N % 128 == 0, even H and W) the conv runs 9/16 of the direct multiplies;
 * ddk_convT_wino_splits() > 0 says the shape is eligible and how many channel-chunk slabs it leaves in the workspace. */
int ddk_pack_convT_weight_wino(const float* w_iohw, float* dst, int I, int O, int i_pad, ddk_stream_t s);
int ddk_convT_wino_splits(int B, int H, int W, int cin, int N);
/* input gradient of conv3x3 s1 / 1x1: run ddk_conv_forward on dY with this operand; [I_pad][taps][O_pad], taps flipped */
int ddk_pack_conv_weight_dgrad(const float* w_oihw, float* dst, int O, int I, int KH, int KW, int i_pad, int o_pad,
                               ddk_stream_t s);
/* [B][H][W][C] -> [B][Ho][Wo][C], values on the even grid, zeros elsewhere (input gradient of the stride-2 conv) */
int ddk_zero_stuff2(const float* in, float* out, int B, int H, int W, int Ho, int Wo, int C, ddk_stream_t s);
size_t ddk_conv_wgrad_workspace_bytes(int kind, int B, int H, int W, int cx, int N);
/* grad_w[(n*cw + c_off + c)*taps + tap] += sum_m dy[m][n] x[pix(m)+tap][c], c < c_real (see csrc/conv_wgrad.hip) */
int ddk_conv_wgrad(int kind, const float* x, const float* dy, float* grad_w, int B, int H, int W, int cx, int c_real,
                   int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* ddk_conv_wgrad that also accumulates the bias gradient: grad_b[n] += sum_m dy[m][n] when grad_b != null (the column sums ride
 * on the weight-gradient GEMM's launches; same workspace size).  Not for the ConvTranspose2d form (x and dy swapped there). */
int ddk_conv_wgrad_bias(int kind, const float* x, const float* dy, float* grad_w, float* grad_b, int B, int H, int W, int cx, int c_real,
                        int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* grad_b[n] (+)= sum_m dy[m][n]; N % 4 == 0; workspace >= 256*N floats */
int ddk_bias_grad(const float* dy, float* grad_b, long long M, int N, int accumulate, void* workspace,
                  size_t workspace_bytes, ddk_stream_t s);
/* y = dropout_p(mish(gn(x)) + temb) + addend; the keep mask is a pure function of (seed, layer, element).
   Group slabs up to 16384 elements stay in registers (workspace unused, may be NULL); larger ones (full-resolution
   DDPM) stream through ddk_groupnorm_train_workspace_bytes() of scratch. */
size_t ddk_groupnorm_train_workspace_bytes(int B, int HW, int C, int groups);
/* Device-side dropout epoch (mixed into every mask key): bump != 0 increments it, else it is set to `set_to`.
   Lets a captured (hipGraph) training step draw fresh masks on every replay. */
int ddk_dropout_epoch(unsigned long long set_to, int bump, ddk_stream_t s);
int ddk_groupnorm_mish_train_fwd(const float* x, const float* gamma, const float* beta, const float* temb,
                                 int temb_stride, const float* addend, float drop_p, uint64_t seed, uint32_t layer,
                                 float* out, int B, int HW, int C, int groups, float eps, void* workspace,
                                 size_t workspace_bytes, ddk_stream_t s);
/* dx and per-sample partial rows part[4][B][C] = (dtemb, dgamma, dbeta, sum_hw dx); d(addend) is dy itself.  The
   last row, summed over b, is the bias gradient of the conv that produced x (blocks.py:78). */
int ddk_groupnorm_mish_bwd(const float* x, const float* gamma, const float* beta, float drop_p, uint64_t seed,
                           uint32_t layer, const float* dy, float* dx, float* part, int B, int HW, int C, int groups,
                           float eps, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* round 4: the two above with the tensor they read first still in `nslab` >= 2 split-K slabs of the conv that produced it
 * (ddk_conv_args.defer_reduce; stride slab_stride floats): the forward takes x = sum of the slabs (in slab order) + conv_bias while
 * loading and also writes it to raw_out (what the backward reads as x); the backward takes dy = sum of the slabs.  One launch less
 * per conv.  Only group slabs of the register-resident path (ddk_groupnorm_train_workspace_bytes() == 0). */
int ddk_groupnorm_mish_train_fwd_slabs(const float* slabs, int nslab, long long slab_stride, const float* conv_bias, float* raw_out,
                                       const float* gamma, const float* beta, const float* temb, int temb_stride,
                                       const float* addend, float drop_p, uint64_t seed, uint32_t layer, float* out, int B, int HW,
                                       int C, int groups, float eps, ddk_stream_t s);
int ddk_groupnorm_mish_bwd_slabs(const float* x, const float* gamma, const float* beta, float drop_p, uint64_t seed, uint32_t layer,
                                 const float* dy_slabs, int nslab, long long slab_stride, float* dx, float* part, int B, int HW,
                                 int C, int groups, float eps, ddk_stream_t s);
int ddk_rows_sum(const float* rows, int nrows, long long row_stride, float* out, int n, int accumulate, ddk_stream_t s);
/* out[k][n] (+)= sum_r rows[k*batch_stride + r*row_stride + n], k < nbatch */
int ddk_rows_sum_batched(const float* rows, int nbatch, long long batch_stride, int nrows, long long row_stride, float* out,
                         int n, int accumulate, ddk_stream_t s);
/* the same with one target per batch entry, nbatch <= 4, a null target is skipped (dgamma / dbeta / conv-bias gradient of a
 * GroupNorm backward into their three gradient buffers in one launch) */
int ddk_rows_sum_targets(const float* rows, int nbatch, long long batch_stride, int nrows, long long row_stride, float* out0, float* out1,
                         float* out2, float* out3, int n, int accumulate, ddk_stream_t s);
/* round 4: ddk_rows_sum_targets (accumulate = 1) of many calls in a few launches -- the per-channel parameter gradients of every
 * GroupNorm / LayerNorm of one backward pass; records from host memory, 48 to a launch as kernel arguments; same bits */
typedef struct ddk_rows_sum_job {
    const float* rows;
    float* out[4];           /* one target per batch entry, null = skipped */
    long long batch_stride, row_stride;
    long long block0;        /* set by ddk_rows_sum_jobs */
    int nbatch, nrows, n, reserved;
} ddk_rows_sum_job;          /* 80 bytes */
int ddk_rows_sum_jobs(const ddk_rows_sum_job* jobs_host, int n, ddk_stream_t s);
/* dst_k[i] += src[off_k + i], k < nseg; table [nseg][3] int64 on the device = {source offset (floats), destination address, count};
 * max_count = the largest count (sizes the grid).  One launch for the many parameter gradients of one backward. */
int ddk_multi_add(const float* src, const long long* table, int nseg, long long max_count, ddk_stream_t s);
/* round 4: EVERY kernel-layout copy of a model's conv weights refreshed by one launch -- what the optimiser step of
 * trainers/trainer_ddpm.py:142-144 invalidates and the next forward / backward needs again (285 single-tensor pack launches per
 * cfg3 step before).  A job = one copy; the element arithmetic is that of the single-tensor entry point of the same kind, so the
 * bits are the same.  ddk_pack_jobs_layout (host) validates the jobs and fills `total` / `block0`, returning the block count of the
 * launch; ddk_pack_jobs takes the SAME table in device memory. */
enum {
    DDK_PACK_CONV = 0,       /* ddk_pack_conv_weight(_split):   p = O, I, taps, i_pad, split, split_pad (plain: split = I, split_pad = i_pad) */
    DDK_PACK_CONVT = 1,      /* ddk_pack_convT_weight(_padded): p = I, O, Ip, Op */
    DDK_PACK_DGRAD = 2,      /* ddk_pack_conv_weight_dgrad:     p = O, I, taps, i_pad, o_pad */
    DDK_PACK_WINO = 3,       /* ddk_pack_conv_weight_wino:      p = O, I, i_pad */
    DDK_PACK_WINO_DGRAD = 4  /* ddk_pack_conv_weight_wino_dgrad: p = O, I, c_lo, c_hi, o_pad */
};
typedef struct ddk_pack_job {
    const float* src;        /* canonical weight (device) */
    float* dst;              /* its kernel-layout copy (device) */
    long long total;         /* filled by ddk_pack_jobs_layout */
    long long block0;        /* filled by ddk_pack_jobs_layout */
    int kind;                /* DDK_PACK_* */
    int p[7];
} ddk_pack_job;              /* 64 bytes */
/* round 4: the slab reduces of a whole backward pass in a few launches.  ddk_conv_wgrad_defer is ddk_conv_wgrad_bias without its
 * reduce launch: the slabs stay in `workspace` (this call's own until the reduce has run) and *job_out describes the reduce;
 * ddk_wgrad_reduce_jobs takes the records of many such calls from HOST memory and runs them 48 to a launch (they travel as kernel
 * arguments: no device table, capturable).  Same summation order as the per-call reduce, same bits.  Two jobs of one call must not
 * target the same gradient elements. */
typedef struct ddk_wgrad_reduce_job {
    const float* slab;
    float* grad;
    const float* bias_slab;  /* with grad_b */
    float* grad_b;           /* or null */
    long long slab_stride;
    long long block0;        /* set by ddk_wgrad_reduce_jobs */
    int splits, N, ntaps, cx, c_real, cw, c_off, reserved;
} ddk_wgrad_reduce_job;      /* 80 bytes */

/* Training at widths that are not multiples of 32 (blocks.py:75: GroupNorm(8, C) takes any C % 8 == 0): activations keep a pitch
 * CP = pad32(C) whose channels [C, CP) are ZERO, so the conv kernels (forward, input gradient, weight gradient) run unchanged on
 * zero-padded weights and only the two normalisations see the real channel count.  GroupNorm + Mish (+ temb[b][c], temb_stride floats
 * per image) (+ Dropout(drop_p), mask = f(seed, layer, element)) (+ addend): training forward; backward: dx and part [4][B][C] = per
 * (image, channel) sums (dtemb, dgamma, dbeta, sum of dx over the pixels = the bias gradient of the conv in front). */
int ddk_groupnorm_mish_generic_train_fwd(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                                         const float* addend, float drop_p, unsigned long long seed, unsigned layer, float* out, int B, int HW,
                                         int CP, int C, int groups, float eps, ddk_stream_t s);
int ddk_groupnorm_mish_generic_bwd(const float* x, const float* gamma, const float* beta, float drop_p, unsigned long long seed, unsigned layer,
                                   const float* dy, float* dx, float* part, int B, int HW, int CP, int C, int groups, float eps,
                                   ddk_stream_t s);
/* Channel LayerNorm (blocks.py:57-60, eps on the std) over the C real channels of CP-pitched rows (padding written as zero), and its
 * backward: dx (+ addend) and the partial rows part [2][nparts][C] of (dg, db); *nparts_out workgroups were used (<= max_parts). */
int ddk_chan_layernorm_generic(const float* x, const float* g, const float* b, float* out, long long M, int CP, int C, float eps,
                               ddk_stream_t s);
int ddk_chan_layernorm_generic_bwd(const float* x, const float* g, const float* dy, const float* addend, float* dx, float* part,
                                   int max_parts, int* nparts_out, long long M, int CP, int C, float eps, ddk_stream_t s);
/* the deferred weight-gradient reduce (ddk_wgrad_reduce_job above) */
int ddk_conv_wgrad_defer(int kind, const float* x, const float* dy, float* grad_w, float* grad_b, int B, int H, int W, int cx, int c_real,
                         int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_wgrad_reduce_job* job_out, ddk_stream_t s);
int ddk_wgrad_reduce_jobs(const ddk_wgrad_reduce_job* jobs_host, int n, ddk_stream_t s);
long long ddk_pack_jobs_layout(ddk_pack_job* jobs_host, int n);
int ddk_pack_jobs(const ddk_pack_job* jobs_dev, int n, long long blocks, ddk_stream_t s);
int ddk_chan_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* part, int max_parts,
                           int* nparts_out, long long M, int C, float eps, ddk_stream_t s);
/* round 4: the same with dx += addend -- the gradient that reaches x over the Residual around the PreNorm (blocks.py:13-14) */
int ddk_chan_layernorm_bwd_add(const float* x, const float* g, const float* dy, const float* addend, float* dx, float* part,
                               int max_parts, int* nparts_out, long long M, int C, float eps, ddk_stream_t s);
/* training-path linear attention: softmax statistics of k (column max, sum of exp) and the backward.  Their reductions over the
 * pixels are split over workgroups on large maps; `workspace` holds the partials (ddk_linattn_train_workspace_bytes, may be 0). */
size_t ddk_linattn_train_workspace_bytes(int B, int HW, int heads);
int ddk_linattn_stats(const float* qkv, float* stats, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s);
int ddk_linattn_bwd(const float* qkv, const float* dout, const float* ctx, const float* stats, float* dctx,
                    float* dqkv, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* round 4: ddk_linattn_bwd without statistics saved by the forward: recomputed from qkv into stats_scratch [B][heads][2][32] -- inside
 * the dctx launch where one workgroup covers the pixels of an (image, head), by ddk_linattn_stats' launches on large maps */
int ddk_linattn_bwd_recompute(const float* qkv, const float* dout, const float* ctx, float* stats_scratch, float* dctx, float* dqkv,
                              int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s);
int ddk_mish_bwd(const float* x, const float* dy, float* dx, long long n, ddk_stream_t s);
int ddk_tanh_bwd(const float* y, const float* dy, float* dx, long long n, ddk_stream_t s);
int ddk_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, ddk_stream_t s);
int ddk_upsample_nearest2_bwd(const float* dy, float* dx, int B, int H, int W, int C, ddk_stream_t s);
/* round 4: the dDDPM autoencoder objective of the 'simple' loss (dddpm.py:155-177) from the two per-sample losses in ONE launch:
 * out[0] = latent + recon, out[1] = latent = mean_b l_ddpm[b], out[2] = recon = mean_b (t[b] < t_rec_max ? l_rec[b] : 0); and its
 * backward d l_ddpm[b] = g[0] / B, d l_rec[b] = (t[b] < t_rec_max) g[0] / B */
int ddk_ae_objective(const float* l_ddpm, const float* l_rec, const int64_t* t, int t_rec_max, int B, float* out, ddk_stream_t s);
int ddk_ae_objective_bwd(const float* g, const int64_t* t, int t_rec_max, int B, float* d_ddpm, float* d_rec, ddk_stream_t s);
int ddk_sq_err_grad(const float* a, const float* b, const float* scale, float* out, int B, long long per,
                    ddk_stream_t s);
int ddk_scale_per_sample(const float* x, const float* scale, float* out, int B, long long per, ddk_stream_t s);
int ddk_conv1x1_small_n_bwd(const float* a, const float* w, const float* dy, float* da, float* part, int max_rows,
                            int* nrows_out, long long M, int C, int n_out, ddk_stream_t s);
/* workspace of ddk_small_gemm when it splits a long contraction over workgroups (0: it does not) */
size_t ddk_small_gemm_workspace_bytes(int M, int N, int K, int ldc);
int ddk_small_gemm(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                   int accumulate, void* workspace, size_t workspace_bytes, ddk_stream_t s);
int ddk_sincos_embed(const int64_t* t, const float* freqs, float* e, int B, int dim, ddk_stream_t s);
int ddk_bias_act(float* y, const float* bias, float* act, long long M, int N, ddk_stream_t s);
/* optimiser on flat fp32 buffers (trainer_ddpm.py:142-148, trainers/ema.py:36-44) */
int ddk_grad_norm_clip(const float* g, long long n, float max_norm, float* out2, void* workspace,
                       size_t workspace_bytes, ddk_stream_t s);
/* scalars are doubles: torch derives 1-beta, the bias corrections and lr/bc1 in double before rounding to fp32 */
int ddk_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                  double eps, int step, const float* clip2, ddk_stream_t s);
int ddk_ema_update(float* p_ema, const float* p, long long n, float decay, ddk_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* DDK_H */
