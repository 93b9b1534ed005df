// attention.hip -- LinearAttention core (reference models/unet/blocks.py:126-134).
//
//   q, k, v = split(to_qkv(x));  k = softmax over the n = H*W pixels;
//   ctx[d][e] = sum_n k[d][n] v[e][n]   (32 x 32 per (sample, head));   out[e][n] = sum_d ctx[d][e] q[d][n]
//
// There is no n x n matrix: per (sample, head) the state is 32x32, so the work is two streaming passes
// over the [n][3*heads*32] projection (NHWC: a pixel's q|k|v are contiguous).  The 1x1 projections
// to_qkv / to_out run on the MFMA implicit-GEMM kernel (conv_igemm.hip); this file is the part between.
#include "ddk_internal.h"

namespace ddk {

constexpr int DH = 32;  // dim_head (blocks.py:119)

// One workgroup per (b, head, split of the pixel range).  Pass 1: column max of k over its pixels.  Pass 2: tiles of
// 64 pixels -> LDS (exp(k - max), v), each thread accumulates a 1x4 strip of ctx plus the softmax denominator of its
// row.  With one split the normalised context is written directly; otherwise the split's (max, denominator,
// unnormalised ctx) go to the workspace and linattn_merge_kernel combines them in split order (deterministic):
// softmax is invariant to the subtracted constant, so partials rescale by exp(m_s - max_s m_s).
constexpr int PART = DH + DH + DH * DH;  // floats per partial: max[32], den[32], acc[32][32]

__global__ __launch_bounds__(256) void linattn_context_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                              float* __restrict__ part, int HW, int heads, int splits, int rows_per_split) {
    __shared__ __attribute__((aligned(16))) float kexp[64 * DH];
    __shared__ __attribute__((aligned(16))) float vs[64 * DH];
    __shared__ float smax[8 * DH];
    const int bh = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int b = bh / heads, h = bh % heads;
    const int HC = heads * DH, RS = 3 * HC;
    const float* base = qkv + (long long)b * HW * RS;
    const float* kp = base + HC + h * DH;
    const float* vp = base + 2 * HC + h * DH;
    const int tid = threadIdx.x;
    const int n_begin = sp * rows_per_split, n_end = min(HW, n_begin + rows_per_split);

    {   // pass 1: max_n k[n][d] over this split
        const int d = tid & 31, ng = tid >> 5;
        float m = -INFINITY;
        for (int n = n_begin + ng; n < n_end; n += 8) m = fmaxf(m, kp[(long long)n * RS + d]);
        smax[ng * DH + d] = m;
        __syncthreads();
        if (tid < DH) {
            float mm = smax[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) mm = fmaxf(mm, smax[j * DH + tid]);
            smax[tid] = mm;
        }
        __syncthreads();
    }

    const int d = tid >> 3, e0 = (tid & 7) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float den = 0.f;
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx4 = tid + j * 256;
            const int row = idx4 >> 3, c = (idx4 & 7) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (n0 + row < n_end) {
                kv = *reinterpret_cast<const float4*>(kp + (long long)(n0 + row) * RS + c);
                vv = *reinterpret_cast<const float4*>(vp + (long long)(n0 + row) * RS + c);
                kv.x = expf(kv.x - smax[c]);
                kv.y = expf(kv.y - smax[c + 1]);
                kv.z = expf(kv.z - smax[c + 2]);
                kv.w = expf(kv.w - smax[c + 3]);
            }
            *reinterpret_cast<float4*>(kexp + row * DH + c) = kv;
            *reinterpret_cast<float4*>(vs + row * DH + c) = vv;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < 64; ++n) {
            const float kd = kexp[n * DH + d];
            const float4 v4 = *reinterpret_cast<const float4*>(vs + n * DH + e0);
            acc.x += kd * v4.x; acc.y += kd * v4.y; acc.z += kd * v4.z; acc.w += kd * v4.w;
            den += kd;
        }
        __syncthreads();
    }
    if (splits == 1) {
        const float inv = 1.0f / den;
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
        *reinterpret_cast<float4*>(ctx + ((long long)bh * DH + d) * DH + e0) = acc;
    } else {
        float* pp = part + ((long long)bh * splits + sp) * PART;
        if (e0 == 0) { pp[d] = smax[d]; pp[DH + d] = den; }
        *reinterpret_cast<float4*>(pp + 2 * DH + d * DH + e0) = acc;
    }
}

__global__ __launch_bounds__(256) void linattn_merge_kernel(const float* __restrict__ part, float* __restrict__ ctx, int splits) {
    const int bh = blockIdx.x, tid = threadIdx.x;
    const int d = tid >> 3, e0 = (tid & 7) * 4;
    const float* pp = part + (long long)bh * splits * PART;
    float M = -INFINITY;
    for (int s = 0; s < splits; ++s) M = fmaxf(M, pp[s * PART + d]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float den = 0.f;
    for (int s = 0; s < splits; ++s) {
        const float w = expf(pp[s * PART + d] - M);
        const float4 a = *reinterpret_cast<const float4*>(pp + s * PART + 2 * DH + d * DH + e0);
        den += pp[s * PART + DH + d] * w;
        acc.x += a.x * w; acc.y += a.y * w; acc.z += a.z * w; acc.w += a.w * w;
    }
    const float inv = 1.0f / den;
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    *reinterpret_cast<float4*>(ctx + ((long long)bh * DH + d) * DH + e0) = acc;
}

// Small maps (H*W <= 64: the 8x8 and 4x4 levels): context AND apply for one (b, head) in one workgroup -- k, v, q of the head
// (<= 64 x 32 each) sit in LDS, so the separate apply launch (and its re-read of q) disappears.  Same arithmetic as the two
// kernels above: column max of k, exp, ctx = (exp k)^T v / den, out = q ctx.
__global__ __launch_bounds__(256) void linattn_small_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, float* __restrict__ out,
                                                            int HW, int heads) {
    __shared__ __attribute__((aligned(16))) float ks[64 * DH];
    __shared__ __attribute__((aligned(16))) float vs[64 * DH];
    __shared__ float qs[64 * (DH + 1)];
    __shared__ float cs[DH * (DH + 4)];
    __shared__ float smax[8 * DH];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int HC = heads * DH, RS = 3 * HC;
    const float* base = qkv + (long long)b * HW * RS + h * DH;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {                       // 64 rows x 8 float4 per operand
        const int idx4 = tid + j * 256;
        const int row = idx4 >> 3, c = (idx4 & 7) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), kv = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), vv = qv;
        if (row < HW) {
            const float* r = base + (long long)row * RS + c;
            qv = *reinterpret_cast<const float4*>(r);
            kv = *reinterpret_cast<const float4*>(r + HC);
            vv = *reinterpret_cast<const float4*>(r + 2 * HC);
        }
        *reinterpret_cast<float4*>(ks + row * DH + c) = kv;
        *reinterpret_cast<float4*>(vs + row * DH + c) = vv;
        qs[row * (DH + 1) + c] = qv.x; qs[row * (DH + 1) + c + 1] = qv.y; qs[row * (DH + 1) + c + 2] = qv.z; qs[row * (DH + 1) + c + 3] = qv.w;
    }
    __syncthreads();
    {   // max_n k[n][d]  (rows >= HW hold -inf)
        const int d = tid & 31, ng = tid >> 5;
        float m = -INFINITY;
        for (int n = ng; n < 64; n += 8) m = fmaxf(m, ks[n * DH + d]);
        smax[ng * DH + d] = m;
        __syncthreads();
        if (tid < DH) {
            float mm = smax[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) mm = fmaxf(mm, smax[j * DH + tid]);
            smax[tid] = mm;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {                       // exp(k - max) in place; padding rows -> exp(-inf) = 0
        const int i = tid + j * 256;
        ks[i] = expf(ks[i] - smax[i & 31]);
    }
    __syncthreads();
    {
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float den = 0.f;
#pragma unroll 8
        for (int n = 0; n < 64; ++n) {
            const float kd = ks[n * DH + d];
            const float4 v4 = *reinterpret_cast<const float4*>(vs + n * DH + e0);
            acc.x += kd * v4.x; acc.y += kd * v4.y; acc.z += kd * v4.z; acc.w += kd * v4.w;
            den += kd;
        }
        const float inv = 1.0f / den;
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
        *reinterpret_cast<float4*>(ctx + (((long long)b * heads + h) * DH + d) * DH + e0) = acc;
        cs[d * (DH + 4) + e0] = acc.x; cs[d * (DH + 4) + e0 + 1] = acc.y; cs[d * (DH + 4) + e0 + 2] = acc.z; cs[d * (DH + 4) + e0 + 3] = acc.w;
    }
    __syncthreads();
    {   // out[n][e] = sum_d ctx[d][e] q[n][d]: thread = (pixel n, 8 consecutive e)
        const int n = tid >> 2, e0 = (tid & 3) * 8;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int d = 0; d < DH; ++d) {
            const float qd = qs[n * (DH + 1) + d];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += cs[d * (DH + 4) + e0 + e] * qd;
        }
        if (n < HW) {
            float* op = out + ((long long)b * HW + n) * HC + h * DH + e0;
            *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// out[b][n][h*32+e] = sum_d ctx[b][h][d][e] q[b][n][h*32+d]: per head a (64 px x 32 d) x (32 d x 32 e) product on the matrix
// pipe.  One workgroup = 64 pixels of one sample, one wave per head: the q tile is staged through LDS with coalesced
// float4 loads (pitch 33: conflict-free operand reads), ctx fragments come straight from global (16 values per lane),
// 2 x 16 v_mfma_f32_32x32x2_f32 per wave, rows of 128 contiguous bytes out.  (The earlier VALU version re-read ctx from LDS
// 256 times per thread and was LDS-issue bound: 15 us at 32x32.)
typedef float f32x16_att __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void linattn_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                            float* __restrict__ out, int HW, int heads, int tiles_per_sample) {
    extern __shared__ __attribute__((aligned(16))) float qs[];   // [heads][64 px][33]
    const int b = blockIdx.x / tiles_per_sample, tile = blockIdx.x % tiles_per_sample;
    const int HC = heads * DH, RS = 3 * HC;
    const int nthreads = 64 * heads;
    const int n0 = tile * 64;
    // q tile: 64 rows x HC floats, consecutive threads along the row
    const int f4_per_row = HC / 4;
    for (int i = threadIdx.x; i < 64 * f4_per_row; i += nthreads) {
        const int row = i / f4_per_row, c4 = i - row * f4_per_row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + row < HW) v = *reinterpret_cast<const float4*>(qkv + ((long long)b * HW + n0 + row) * RS + c4 * 4);
        const int hh = (c4 * 4) / DH, d = (c4 * 4) % DH;
        float* dst = qs + (hh * 64 + row) * 33 + d;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, fh = lane >> 5;
    // B fragments: ctx[h][d = 2*kk + fh][e = lane & 31]
    const float* cp = ctx + ((long long)b * heads + h) * DH * DH + l31;
    float bf[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) bf[kk] = cp[(2 * kk + fh) * DH];
    __syncthreads();
    const float* qh = qs + h * 64 * 33;
    f32x16_att acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const float a0 = qh[l31 * 33 + 2 * kk + fh], a1 = qh[(32 + l31) * 33 + 2 * kk + fh];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[kk], acc1, 0, 0, 0);
    }
    // C layout: col e = lane & 31, row px = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (n0 + px < HW) out[((long long)b * HW + n0 + px) * HC + h * DH + l31] = acc0[r];
        if (n0 + 32 + px < HW) out[((long long)b * HW + n0 + 32 + px) * HC + h * DH + l31] = acc1[r];
    }
}

// Pixel-range splits: ~1024 workgroups at most, at least one 64-pixel tile each.
static void linattn_splits(int B, int HW, int heads, int& splits, int& rows) {
    const int max_s = (int)ceil_div(HW, 64);
    int s = 1024 / (B * heads);
    if (s < 1) s = 1;
    if (s > max_s) s = max_s;
    rows = (int)(ceil_div(ceil_div(HW, s), 64) * 64);
    splits = (int)ceil_div(HW, rows);
}

size_t linattn_context_workspace_bytes(int B, int HW, int heads) {
    if (B <= 0 || HW <= 0 || heads <= 0) return 0;
    int s, rows;
    linattn_splits(B, HW, heads, s, rows);
    return s > 1 ? (size_t)B * heads * s * PART * sizeof(float) : 0;
}

int linattn_context(const float* qkv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, hipStream_t st) {
    DDK_REQUIRE(qkv && ctx && B > 0 && HW > 0 && heads > 0, "linattn_context: arguments");
    DDK_REQUIRE(aligned16(qkv) && aligned16(ctx) && aligned16(workspace), "linattn_context: alignment");
    int s, rows;
    linattn_splits(B, HW, heads, s, rows);
    if (s > 1 && (!workspace || workspace_bytes < (size_t)B * heads * s * PART * sizeof(float))) {
        if (workspace) {   // a workspace that is too small is a caller bug; none at all selects the unsplit kernel
            set_error("linattn_context: workspace too small (%zu < %zu)", workspace_bytes, (size_t)B * heads * s * PART * sizeof(float));
            return DDK_ERR_WORKSPACE;
        }
        s = 1;
        rows = HW;
    }
    hipLaunchKernelGGL(linattn_context_kernel, dim3(B * heads * s), dim3(256), 0, st, qkv, ctx, static_cast<float*>(workspace), HW, heads, s, rows);
    DDK_TRY(check_launch("linattn_context_kernel"));
    if (s > 1) {
        hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * heads), dim3(256), 0, st, static_cast<const float*>(workspace), ctx, s);
        DDK_TRY(check_launch("linattn_merge_kernel"));
    }
    return DDK_OK;
}

// context + apply in one launch when the map is small enough for one workgroup per (b, head); returns 0 and does nothing else
int linattn_fused_small(const float* qkv, float* ctx, float* out, int B, int HW, int heads, hipStream_t st) {
    DDK_REQUIRE(qkv && ctx && out && B > 0 && HW > 0 && HW <= 64 && heads > 0, "linattn_fused_small: arguments (HW <= 64)");
    DDK_REQUIRE(aligned16(qkv) && aligned16(ctx) && aligned16(out), "linattn_fused_small: alignment");
    hipLaunchKernelGGL(linattn_small_kernel, dim3(B * heads), dim3(256), 0, st, qkv, ctx, out, HW, heads);
    return check_launch("linattn_small_kernel");
}

int linattn_apply(const float* qkv, const float* ctx, float* out, int B, int HW, int heads, hipStream_t st) {
    DDK_REQUIRE(qkv && ctx && out && B > 0 && HW > 0, "linattn_apply: arguments");
    DDK_REQUIRE(heads >= 1 && heads <= 8, "linattn_apply: heads must be in 1..8");
    DDK_REQUIRE(aligned16(qkv) && aligned16(ctx) && aligned16(out), "linattn_apply: alignment");
    const int tiles = (int)ceil_div(HW, 64);
    const size_t lds = (size_t)heads * 64 * 33 * sizeof(float);
    hipLaunchKernelGGL(linattn_apply_kernel, dim3(B * tiles), dim3(64 * heads), lds, st, qkv, ctx, out, HW, heads, tiles);
    return check_launch("linattn_apply_kernel");
}

}  // namespace ddk

extern "C" {
size_t ddk_linattn_context_workspace_bytes(int B, int HW, int heads) { return ddk::linattn_context_workspace_bytes(B, HW, heads); }
int ddk_linattn_context(const float* qkv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    return ddk::linattn_context(qkv, ctx, B, HW, heads, workspace, workspace_bytes, ddk::as_stream(s));
}
int ddk_linattn_fused_small(const float* qkv, float* ctx, float* out, int B, int HW, int heads, ddk_stream_t s) {
    return ddk::linattn_fused_small(qkv, ctx, out, B, HW, heads, ddk::as_stream(s));
}
int ddk_linattn_apply(const float* qkv, const float* ctx, float* out, int B, int HW, int heads, ddk_stream_t s) {
    return ddk::linattn_apply(qkv, ctx, out, B, HW, heads, ddk::as_stream(s));
}
}
