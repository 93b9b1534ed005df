// Internal helpers shared by the libddk.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "ddk.h"

#ifdef DDK_HOST_SANITIZE      // `make asan`: the host half checked on a GPU-less machine (host_sanitize.h); never part of libddk.so
#include "host_sanitize.h"
#endif

namespace ddk {

void set_error(const char* fmt, ...);

inline int fail_arg(const char* what) {
    set_error("bad argument: %s", what);
    return DDK_ERR_ARG;
}

// Checks the launch that was just enqueued (no sync: only catches configuration errors).
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DDK_ERR_HIP;
    }
    return DDK_OK;
}

#define DDK_REQUIRE(cond, msg)              \
    do {                                    \
        if (!(cond)) return ddk::fail_arg(msg); \
    } while (0)

#define DDK_HIP(call)                                                    \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) {                                          \
            ddk::set_error("%s: %s", #call, hipGetErrorString(e_));      \
            return DDK_ERR_HIP;                                          \
        }                                                                \
    } while (0)

#define DDK_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != DDK_OK) return rc_; \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline hipStream_t as_stream(ddk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------------
// Mish(x) = x * tanh(softplus(x)).  With n = e^x (e^x + 2): tanh(ln(1+e^x)) = n / (n + 2), which has
// no cancellation for x << 0 and one exp + one divide.  For x > 20 softplus(x) == x in torch
// (threshold 20) and tanh(x) rounds to 1 in fp32, so Mish(x) == x.
// exp and the divide use the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): ~8 instructions per
// element instead of ~30 with the IEEE-exact library versions -- in the GroupNorm kernels the exact form was a third
// of the kernel's time.  Relative error of the result <= ~4e-7 (tests hold elementwise ops to 5e-6).
__device__ __forceinline__ float mish_f(float x) {
    if (x > 20.0f) return x;
    const float e = __expf(x);
    const float n = e * (e + 2.0f);
    return x * (n * __frcp_rn(n + 2.0f));
}

// d/du [u * tanh(softplus(u))]: with e = exp(u), n = e(e+2), t = n/(n+2):  t + u * 4 e (e+1) / (n+2)^2
// (hardware exp / reciprocal as in mish_f: one v_exp_f32 + one v_rcp_f32 instead of a library exp and two IEEE divides -- this
// runs per element in conv epilogues (ddk_conv_args.dmish_src) and in the GroupNorm backward)
__device__ __forceinline__ float mish_grad_f(float u) {
    if (u > 20.0f) return 1.0f;
    const float e = __expf(u);
    const float n = e * (e + 2.0f);
    const float r = __frcp_rn(n + 2.0f);
    return n * r + u * (4.0f * e * (e + 1.0f)) * (r * r);
}

// u / upr for the float4-units-per-pixel count of a GroupNorm group (1, 2, 4 or 8 in every reference configuration): a
// shift when upr is a power of two -- the integer divide is ~30 VALU ops and these kernels do it per element, per pass.
__device__ __forceinline__ int div_upr(int u, int upr) {
    return (upr & (upr - 1)) == 0 ? u >> (31 - __builtin_clz(upr)) : u / upr;
}
__device__ __forceinline__ long long div_upr(long long u, int upr) {
    return (upr & (upr - 1)) == 0 ? u >> (31 - __builtin_clz(upr)) : u / upr;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over the whole workgroup (blockDim.x a multiple of 64, <= 1024); result valid in every thread.
// `red` is at least 17 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : 0.0f;
    t = wave_sum(t);
    return t;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : -INFINITY;
    t = wave_max(t);
    return t;
}

// core.hip: per-device one-time setup; conv_igemm.hip / conv_wgrad.hip provide the per-file parts
int ensure_device_init();
int conv_init_device();
int wgrad_init_device();

// ---- internal launchers used by the UNet plan (same arithmetic as the public entry points) ----
// conv_igemm.hip
size_t conv_workspace_bytes(int kind, int B, int H, int W, int cin, int N);
// Channel LayerNorm folded into a 1x1 conv: a.weight must hold W o g (per input channel), c1 = W g, c2 = W b per output
// channel.  Only valid when conv_ln_fold_ok() says so for the shape.
struct ConvLnFold {
    const float* c1;
    const float* c2;
    float eps;
};
// GroupNorm + Mish (+ time shift, + residual = ddk_conv_args.resid) finished inside the Winograd conv's launch: the workgroups
// of one image exchange their per-tile statistics through `records` / `counters` (conv_wino_cluster_ws_floats() floats, the
// counters zero before the first launch; they re-arm themselves)
struct WinoGnFuse {
    const float* gamma;
    const float* beta;
    const float* temb;
    const long long* temb_rows;
    int temb_stride;
    float eps;
    int groups;
    float* records;
    unsigned* counters;
    unsigned* fail;         // sticky count of workgroups that gave up waiting (may be null)
    unsigned* pairs = nullptr;   // shapes whose channel chunks are split (conv_wino_cluster_split_np): conv_wino_cluster_pair_words() zeroed
                                 // words; the partial tiles go through ddk_conv_args::workspace ((splits - 1) slabs)
    // optional: the addend is a 1x1 conv of a narrow tensor, evaluated in the epilogue (the first ResnetBlock's res_conv of the
    // <= 8-channel input, blocks.py:103,115): res_x [B*H*W][res_cin], res_w [N][res_ld] (first res_cin entries of a row), res_b [N] or null
    const float* res_x = nullptr;
    const float* res_w = nullptr;
    const float* res_b = nullptr;
    int res_cin = 0, res_ld = 0;
};
bool conv_ln_fold_ok(int B, int H, int W, int cin, int N);
int conv_forward(const ddk_conv_args& a, hipStream_t st, const ConvLnFold* ln = nullptr, const WinoGnFuse* fuse = nullptr);
int conv_splits(int kind, int B, int H, int W, int cin, int N);
// conv_wino.hip
bool conv_wino_ok(int kind, int H, int W, int cin, int N);
int conv_wino_splits(int B, int H, int W, int cin, int N);
int conv_wino_forward(const ddk_conv_args& a, int splits, hipStream_t st, const WinoGnFuse* fuse = nullptr);
int conv_wino_cluster_np(int B, int H, int W, int cin, int N, int groups);    // m tiles per image when eligible, else 0
int conv_wino_cluster_split_np(int B, int H, int W, int cin, int N, int groups, int* splits_out);   // the same for k-split shapes
size_t conv_wino_cluster_pair_words(int B, int H, int W, int N);
bool conv_wino_variant_new(int B, int H, int W, int N);   // the shape runs on the 8-matrix-wave kernels (they carry the 1x1 addend)
bool conv_wino_cluster_device_ok();     // a whole MI355X (256 CUs, 8 XCDs, no CU mask): the only place a cluster is co-resident
size_t conv_wino_cluster_ws_floats(int B, int H, int W, int N);
unsigned conv_wino_cluster_timeouts();
// transpose conv 4x4 stride 2 as Winograd F(2x2, 2x2) per output phase (weight_wino = ddk_pack_convT_weight_wino)
bool convT_wino_ok(int H, int W, int cin, int N);
int convT_wino_splits(int B, int H, int W, int cin, int N);
int convT_wino_forward(const ddk_conv_args& a, int splits, hipStream_t st);
int conv_wino_init_device();
int conv_wino_stats_parts(int B, int H, int W, int cin, int N, int groups);   // tiles per image, or 0
// conv_first.hip: Conv2d(C_in <= 8, N, 3, padding=1) on the unpadded input, GroupNorm partials in the epilogue; with
// counter / t_cur it also does the sampler's per-step bookkeeping (t_cur[b] <- counter; counter -= 1)
bool conv_first_ok(int cin, int N, int H, int W, int groups);
int conv_first(const float* x, const float* wp, const float* bias, float* out, float* gn_partials, int B, int H, int W, int cin, int N,
               int groups, int64_t* counter, int64_t* t_cur, hipStream_t st);
int conv_first_init_device();
// ... with the Block's GroupNorm + Mish + time shift finished in the same launch (the image's tiles exchange their statistics):
// records = conv_first_gn_ws_floats() floats, counters = [B][8 * 16] zeroed words that re-arm themselves, fail = sticky give-up count
bool conv_first_gn_ok(int cin, int N, int H, int W, int groups);
size_t conv_first_gn_ws_floats(int B, int H, int W);
int conv_first_gn(const float* x, const float* wp, const float* bias, const float* gamma, const float* beta, const float* temb,
                  int temb_stride, const long long* temb_rows, float eps, float* out, int B, int H, int W, int cin, int N, int groups,
                  float* records, unsigned* counters, unsigned* fail, int64_t* counter, int64_t* t_cur, hipStream_t st);
// (in the sampler this kernel reads the step counter in EVERY workgroup -- the time shift's row -- so it must not decrement it: the
//  step's last kernel does, final_tail / p_sample_update with dec_counter)
// conv1x1_ws.hip: 1x1 conv with 128 input channels on a large map as a weights-stationary, pixel-streaming GEMM
// (w = the packed 1x1 weight [N][128]; ln as in conv_forward)
// conv1x1_sm.hip: 1x1 conv + bias + residual on small maps (32x32 tiles, the four waves split K, no ring)
bool conv1x1_sm_ok(long long M, int c0, int c1, int N);
int conv1x1_sm(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* resid, float* out,
               long long M, int N, const ConvLnFold* ln, hipStream_t st);
int conv1x1_sm_init_device();
// conv1x1_stream.hip: 1x1 conv between 32 / 64-channel tensors on large maps as a memory stream (weights in registers, no LDS)
bool conv1x1_stream_ok(long long M, int cin, int N);
int conv1x1_stream(const float* x, int cin, const float* w, const float* bias, const float* dmish_src, const float* resid, float* out,
                   float* mish_out, long long M, int N, int pre_mish, int post_mish, hipStream_t st);
bool conv1x1_ws_ok(long long M, int K, int N);
// images > 0: PER-IMAGE weights -- w is [images][128][128], the LayerNorm vectors [images][128], N == 128, M / images pixels per image
int conv1x1_ws(const float* x, const float* w, const float* bias, const float* resid, float* out, long long M, int N, const ConvLnFold* ln,
               hipStream_t st, int images = 0);
int conv1x1_ws_init_device();
// conv_local.hip: conv3x3 + GroupNorm + Mish (+shift, +residual) in one launch for 4x4 / 8x8 maps
// the addend of the image-local kernels may still be in split-K form (the 1x1 skip conv's slabs): n slabs `stride` floats apart,
// summed in order, plus bias[c] -- the skip conv's reduce launch folded into the consumer's load
struct AddendSlabs {
    int n = 1;
    long long stride = 0;
    const float* bias = nullptr;
};
bool conv_gn_local_ok(int H, int W, int cin, int c0, int N, int groups);
int conv_gn_local(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* gamma,
                  const float* beta, const float* temb, int temb_stride, const long long* temb_rows, const float* addend, float* out,
                  int B, int H, int W, int N, int groups, float eps, hipStream_t st, const AddendSlabs& as = AddendSlabs(),
                  const AddendSlabs& src_slabs = AddendSlabs());      // src_slabs.n > 1: src0 is in split-K form (summed while staged)
int conv_gn_local_init_device();
bool conv_gn_wlocal_ok(int H, int W, int cin, int c0, int N, int groups);     // 64-pixel maps: the same in Winograd form
int conv_gn_wlocal(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* gamma,
                   const float* beta, const float* temb, int temb_stride, const long long* temb_rows, const float* addend, float* out,
                   int B, int H, int W, int N, int groups, float eps, hipStream_t st, const AddendSlabs& as = AddendSlabs(),
                   const AddendSlabs& src_slabs = AddendSlabs());
double conv_flops(int kind, int B, int H, int W, int cin, int N);
// norm_act.hip
size_t groupnorm_workspace_bytes(int B, int HW, int C, int groups);
int gn_train_nsplit(int HW, int cpg);
int gn_stats_partials(const float* x, float* part, int B, int HW, int C, int groups, int ns, hipStream_t st);
int groupnorm_mish(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                   const float* addend, float* out, int B, int HW, int C, int groups, float eps, void* ws, size_t ws_bytes,
                   hipStream_t st, const long long* temb_rows = nullptr);
int groupnorm_mish_ex(const float* x, int nslab, long long slab_stride, const float* cbias, const float* gamma,
                      const float* beta, const float* temb, int temb_stride, const float* addend, float* out, int B, int HW, int C,
                      int groups, float eps, void* ws, size_t ws_bytes, hipStream_t st, const long long* temb_rows = nullptr);
// rc_*: optional on-the-fly 1x1 addend  rc_b[c] + sum_k rc_x[pix][k] rc_w[c*rc_ld + k]  (k < rc_cin <= 8), instead of `addend`
int groupnorm_mish_parts(const float* x, const float* part, int np, const float* gamma, const float* beta, const float* temb,
                         int temb_stride, const float* addend, float* out, int B, int HW, int C, int groups, float eps, hipStream_t st,
                         const long long* temb_rows = nullptr, const float* rc_x = nullptr, const float* rc_w = nullptr,
                         const float* rc_b = nullptr, int rc_cin = 0, int rc_ld = 0);
int chan_layernorm(const float* x, const float* g, const float* b, float* out, long long M, int C, float eps, hipStream_t st);
// widths that are not multiples of 32: C real channels in rows of pitch CP = pad32(C), padding kept zero
int groupnorm_mish_generic(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                           float* out, int B, int HW, int CP, int C, int groups, float eps, hipStream_t st, const long long* temb_rows = nullptr);
int chan_layernorm_generic(const float* x, const float* g, const float* b, float* out, long long M, int CP, int C, float eps, hipStream_t st);
int unary(int op, const float* x, float* out, long long n, hipStream_t st);
int add(const float* a, const float* b, float* out, long long n, hipStream_t st);
int avgpool2(const float* x, float* out, int B, int H, int W, int C, hipStream_t st);
int upsample_nearest2(const float* x, float* out, int B, int H, int W, int C, hipStream_t st);
// attention.hip
size_t linattn_context_workspace_bytes(int B, int HW, int heads);
int linattn_context(const float* qkv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, hipStream_t st,
                    bool kv_only = false);
// folded attention output (attention.hip): per-image C x C matrix A and fold vectors from the context
bool attn_fold_ok(int C, int heads);
bool attn_kvctx_ok(int B, int HW, int C, int heads);     // k, v projection + context in one launch (attention.hip, attn_kvctx_kernel)
size_t attn_kvctx_workspace_bytes(int B, int HW);
int attn_kvctx(const float* x, const float* w_kv, const float* c1, const float* c2, float ln_eps, float* ctx, int B, int HW, void* workspace,
               size_t workspace_bytes, hipStream_t st);
size_t attn_fold_out_floats(int B);
int attn_fold(const float* ctx, const float* wqg, const float* c1q, const float* c2q, const float* wout, const float* bout, float* A,
              float* a1, float* a2, int B, int C, int heads, hipStream_t st);
int linattn_apply(const float* qkv, const float* ctx, float* out, int B, int HW, int heads, hipStream_t st);
int linattn_fused_small(const float* qkv, float* ctx, float* out, int B, int HW, int heads, hipStream_t st);
bool linattn_small_qkv_ok(int HW, int C);
int linattn_small_qkv_init_device();
int qkv_operand_pack(const float* lnw, float* wop, int heads, int cp, hipStream_t st);
int linattn_small_qkv(const float* x, const float* wop, const float* c1, const float* c2, float ln_eps, float* ctx, float* out, int B, int HW,
                      int C, int heads, hipStream_t st);
// time_embed.hip
int time_mlp(const int64_t* t, const float* freqs, const float* w1t, const float* b1, const float* w2t, const float* b2,
             float* act, float* raw, int B, int dim, hipStream_t st);
int time_proj(const float* act, const float* wt, const float* bias, float* out, int B, int dim, int n_out, hipStream_t st);
// layout_pack.hip
int conv1x1_small_n(const float* x, const float* w, const float* bias, float* out, long long M, int C, int n_out, hipStream_t st);
// diffusion.hip
int p_sample_update(float* x, const float* eps_hat, const float* noise, long long noise_step_stride, int t_first,
                    const int64_t* t, const float* c_recip, const float* c_recipm1, const float* c1, const float* c2,
                    const float* sigma, int B, long long per, uint64_t seed, uint32_t stream_id, hipStream_t st,
                    const int64_t* chain_state = nullptr, int64_t* dec_counter = nullptr);
int randn(float* out, long long n, uint64_t seed, uint32_t step, uint32_t stream_id, hipStream_t st);
// GroupNorm (from conv partials) + Mish + 1x1 projection to n_out <= 8 channels (+ the reverse-step update of x) in one launch
bool final_tail_ok(int HW, int C, int groups, int n_out, int np);
int final_tail(const float* raw, const float* part, int np, const float* gamma, const float* beta, float eps, const float* w,
               const float* bias, int n_out, float* eps_out, float* x, const float* noise, long long noise_step_stride, int t_first,
               const int64_t* t, const float* c_recip, const float* c_recipm1, const float* c1, const float* c2, const float* sigma,
               const int64_t* chain_state, uint64_t seed, uint32_t stream_id, int B, int HW, int C, int groups, hipStream_t st,
               int64_t* dec_counter = nullptr);     // dec_counter: the sampler's step counter, decremented by this (last) kernel of the step

}  // namespace ddk
