// attention.hip -- LinearAttention core (reference models/unet/blocks.py:126-134).
//
//   q, k, v = split(to_qkv(x));  k = softmax over the n = H*W pixels;
//   ctx[d][e] = sum_n k[d][n] v[e][n]   (32 x 32 per (sample, head));   out[e][n] = sum_d ctx[d][e] q[d][n]
//
// There is no n x n matrix: per (sample, head) the state is 32x32, so the work is two streaming passes
// over the [n][3*heads*32] projection (NHWC: a pixel's q|k|v are contiguous).  The 1x1 projections
// to_qkv / to_out run on the MFMA implicit-GEMM kernel (conv_igemm.hip); this file is the part between.
#include <cstdlib>

#include "conv_common.h"

namespace ddk {

constexpr int DH = 32;  // dim_head (blocks.py:119)

// One workgroup per (b, head, split of the pixel range).  Pass 1: column max of k over its pixels.  Pass 2: tiles of
// 64 pixels -> LDS (exp(k - max), v), each thread accumulates a 1x4 strip of ctx plus the softmax denominator of its
// row.  With one split the normalised context is written directly; otherwise the split's (max, denominator,
// unnormalised ctx) go to the workspace and linattn_merge_kernel combines them in split order (deterministic):
// softmax is invariant to the subtracted constant, so partials rescale by exp(m_s - max_s m_s).
constexpr int PART = DH + DH + DH * DH;  // floats per partial: max[32], den[32], acc[32][32]

// kv_only: the rows are [k | v] (2 * heads * 32 floats) instead of [q | k | v] -- the folded-attention path projects k and v only
__global__ __launch_bounds__(256) void linattn_context_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                              float* __restrict__ part, int HW, int heads, int splits, int rows_per_split,
                                                              int kv_only) {
    __shared__ __attribute__((aligned(16))) float stage[4 * 32 * 33];   // kexp [64][32] | vs [64][32]; later the waves' partial tiles
    float* kexp = stage;
    float* vs = stage + 64 * DH;
    __shared__ float smax[8 * DH];
    const int bh = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int b = bh / heads, h = bh % heads;
    const int HC = heads * DH, RS = (kv_only ? 2 : 3) * HC;
    const float* base = qkv + (long long)b * HW * RS;
    const float* kp = base + (kv_only ? 0 : HC) + h * DH;
    const float* vp = kp + HC;
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
    // Round 4: ctx^T-tile = (exp k)^T v on the matrix pipe.  Wave w multiplies pixels [16 w, 16 w + 16) of every 64-pixel tile:
    // v_mfma_f32_32x32x2_f32 with row operand kexp[n][d] (d = lane & 31, n = 2 s + lane / 32) and column operand v[n][e] -- eight
    // MFMAs per tile and wave where every thread ran 64 x 5 FMAs behind two LDS reads each (the launch was VALU-bound: fewer or
    // more pixel splits both lose, tools/attn_ctx_bench.py).  The softmax denominators ride along as sums of the row operand.
    typedef float ctx_f32x16 __attribute__((ext_vector_type(16)));
    const int lane = tid & 63, wv = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    ctx_f32x16 macc;
#pragma unroll
    for (int r = 0; r < 16; ++r) macc[r] = 0.f;
    float dsum = 0.f;
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx4 = tid + j * 256;
            const int row = idx4 >> 3, c = (idx4 & 7) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (n0 + row < n_end) {
                kv = *reinterpret_cast<const float4*>(kp + (long long)(n0 + row) * RS + c);
                vv = *reinterpret_cast<const float4*>(vp + (long long)(n0 + row) * RS + c);
                kv.x = __expf(kv.x - smax[c]);
                kv.y = __expf(kv.y - smax[c + 1]);
                kv.z = __expf(kv.z - smax[c + 2]);
                kv.w = __expf(kv.w - smax[c + 3]);
            }
            *reinterpret_cast<float4*>(kexp + row * DH + c) = kv;
            *reinterpret_cast<float4*>(vs + row * DH + c) = vv;
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            const int n = wv * 16 + 2 * s2 + kh;
            const float a = kexp[n * DH + l31];
            macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vs[n * DH + l31], macc, 0, 0, 0);
            dsum += a;
        }
        __syncthreads();
    }
    // the four waves' partial tiles and denominators meet in LDS
    float4 acc;
    float den;
    {
        float* pt = stage;                                 // [4 waves][32 d][33] over the tile staging area (the last barrier freed it)
        static_assert(4 * 32 * 33 >= 2 * 64 * DH, "the staging array holds both tiles");
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + l31] = macc[r];
        dsum += __shfl_xor(dsum, 32, 64);
        __shared__ float dpart[4 * DH];
        if (kh == 0) dpart[wv * DH + l31] = dsum;
        __syncthreads();
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {                      // fixed order: deterministic
            const float* q = pt + (w * 32 + d) * 33 + e0;
            acc.x += q[0]; acc.y += q[1]; acc.z += q[2]; acc.w += q[3];
            den += dpart[w * DH + d];
        }
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

// ------------------------------------------------------------------------------------------------
// k, v projection + context in ONE launch (round 5), for the folded attention block on maps with HW >> C (C = 128, 4 heads):
// blocks.py:57-60 (PreNorm LayerNorm, folded into the weights), :123 (to_qkv's k and v thirds), :129-131 (softmax over the pixels of
// k, ctx = k v^T).  The two-launch form writes the [M][256] kv tensor (33.5 MB at 32x32, batch 32) and reads it back; here a
// workgroup keeps the k and v rows of TWO heads (128 rows of the LayerNorm-folded weight, all of K = 128: 64 KB) in LDS for its life,
// streams 64-pixel tiles of x through a double buffer exactly like conv1x1_ws_kernel<true, false> (weights as the MFMA row operand,
// the next tile's DMA under the 64 MFMAs of this one), and instead of storing the tile it
//   * parks k and v of the tile in the LDS buffer the tile came from (4 x [64 pixels][32], the library's XOR swizzle),
//   * keeps a running column maximum M[d] of k over its pixels (online softmax: when a tile raises M, the partial context and the
//     denominators are rescaled by exp(M_old - M_new)),
//   * accumulates ctx^T += (exp(k - M))^T v on the matrix pipe, wave = (head of the pair, 16 pixels of the tile),
// and leaves ONE partial record {M, denominator, unnormalised ctx} per (image, head, pixel split) in the format of
// linattn_context_kernel, which linattn_merge_kernel combines in split order.  grid = B x 2 head pairs x splits (256 workgroups at
// batch 32: 4 splits of 4 tiles); 512 threads, 132 KB of LDS.
constexpr int KC_K = 128, KC_BM = 64;
constexpr int KC_W_FLOATS = 128 * KC_K, KC_A_FLOATS = KC_BM * KC_K;
constexpr int KC_LDS_FLOATS = KC_W_FLOATS + 2 * KC_A_FLOATS + 2 * KC_BM + 64 + 64 + 8 * 64 + 8 * DH;

#if !defined(DDK_TUNING) || !defined(DDK_KVCTX_ABL)     /* an ablation exists in the tuning build only: the product library cannot be built with one */
#undef DDK_KVCTX_ABL
#define DDK_KVCTX_ABL 0      // tuning experiments only (wrong results): 1 no exp pass, 2 no maximum phases, 4 no context MFMAs, 8 nothing behind the projection, 16 no LayerNorm statistics
#endif
struct KvCtxParams {
    const float* x;      // [B * HW][128]
    const float* w;      // [256][128]: LayerNorm-folded k rows (4 heads x 32), then v rows
    const float* c1;     // [256] fold vectors W g and W b of those rows
    const float* c2;
    float eps;
    float* part;         // [(b * 4 + head) * splits + split][PART]
    int tiles_per_image, tiles_per_split, splits;
};

__global__ __launch_bounds__(512) void attn_kvctx_kernel(const KvCtxParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                                   // [4 chunks][128 rows][32]
    float* As = smem + KC_W_FLOATS;                     // [2 buffers][4 chunks][64 rows][32]; a consumed buffer holds the tile's k | v
    float* rowstat = As + 2 * KC_A_FLOATS;              // [64][2]: (r, r * mean) of the tile's pixels
    float* Mx = rowstat + 2 * KC_BM;                    // [64] running maximum of the k columns (head of the pair, d)
    float* Sc = Mx + 64;                                // [64] exp(M_old - M_new) of the current tile
    float* cmx = Sc + 64;                               // [8][64] column-maximum partials
    float* dpart = cmx + 8 * 64;                        // [8][32] the waves' denominators
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sp = blockIdx.x, hp = blockIdx.y, b = blockIdx.z;         // grid (splits, 2 head pairs, B): scalar without a division
    const int wm = wid & 1, wn = wid >> 1;              // projection: pixels [32 wm, +32), slice rows [32 wn, +32): k h0 | k h1 | v h0 | v h1
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const int prow = lane >> 3, ppos = lane & 7;
    const int dchunk = wid & 3, dhalf = wid >> 2;
    unsigned wvoff[8], avoff[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (dhalf * 8 + j) * 8 + prow;                                       // row of the slice
        const int src = r < 64 ? 64 * hp + r : 128 + 64 * hp + (r - 64);                // row of the [k | v] weight
        wvoff[j] = (unsigned)((src * KC_K + dchunk * 32 + ((ppos ^ ((r >> 1) & 7)) << 2)) * 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (dhalf * 4 + j) * 8 + prow;
        avoff[j] = (unsigned)((r * KC_K + dchunk * 32 + ((ppos ^ ((r >> 1) & 7)) << 2)) * 4);
    }
    {
        const unsigned dst = lds_base + (unsigned)((dchunk * (128 * 32) + dhalf * 2048) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_dma16_s(p.w, wvoff[j], dst + (unsigned)(j * 1024));
    }
    const unsigned a_dst = lds_base + (unsigned)((KC_W_FLOATS + dchunk * (KC_BM * 32) + dhalf * 1024) * 4);
    const int tile0 = b * p.tiles_per_image + sp * p.tiles_per_split;       // 32-bit: stays on the scalar unit
    {
        const float* xb = p.x + (long long)tile0 * (KC_BM * KC_K);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_s(xb, avoff[j], a_dst + (unsigned)(j * 1024));
    }
    // epilogue constants: accumulator register 4q + i = slice row 32 wn + 8 q + 4 h + i of pixel pl
    const int pl = lane & 31, h = lane >> 5;
    float4 e1[4], e2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 32 * wn + 8 * q + 4 * h;
        const int col = r < 64 ? 64 * hp + r : 128 + 64 * hp + (r - 64);
        e1[q] = *reinterpret_cast<const float4*>(p.c1 + col);
        e2[q] = *reinterpret_cast<const float4*>(p.c2 + col);
    }
    const int fsw = (pl >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = pl * 32 + (((2 * q + h) ^ fsw) << 2);
    if (tid < 64) Mx[tid] = -INFINITY;

    // context accumulators of this wave: head hh of the pair, pixels [16 qr, +16) of every tile
    const int hh = wid >> 2, qr = wid & 3, l31 = lane & 31, kh = lane >> 5;
    f32x16 macc;
#pragma unroll
    for (int r = 0; r < 16; ++r) macc[r] = 0.f;
    float dsum = 0.f;
    auto tswz = [](int n, int d) { return n * 32 + ((((d >> 2) ^ ((n >> 1) & 7)) << 2) | (d & 3)); };   // float offset of (pixel n, column d) in a parked tile

    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int buf = 0;
    for (int ti = 0; ti < p.tiles_per_split; ++ti, buf ^= 1) {
        float* Ab = As + buf * KC_A_FLOATS;
        if (!(DDK_KVCTX_ABL & 16)) {   // LayerNorm statistics of the tile's 64 pixel rows (conv1x1_ws_kernel: 8 threads per row, two passes over the resident row)
            const int row = tid >> 3, part = tid & 7;
            const float4* rp = reinterpret_cast<const float4*>(Ab + (part >> 1) * (KC_BM * 32) + row * 32);
            float4 v[4];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = rp[(4 * (part & 1) + i + (row >> 1)) & 7];
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.0f / KC_K);
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                qq += (a * a + bb * bb) + (c * c + d * d);
            }
            qq += __shfl_xor(qq, 1, 64);
            qq += __shfl_xor(qq, 2, 64);
            qq += __shfl_xor(qq, 4, 64);
            if (part == 0) {
                const float r = 1.0f / (sqrtf(qq * (1.0f / KC_K)) + p.eps);
                rowstat[2 * row] = r;
                rowstat[2 * row + 1] = r * mean;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const bool more = ti + 1 < p.tiles_per_split;
        const float* xnext = p.x + (long long)(tile0 + ti + 1) * (KC_BM * KC_K);
        const unsigned dnext = a_dst + (unsigned)((buf ^ 1) * KC_A_FLOATS * 4);
        // ---- projection: D[slice row][pixel] += W[row][k] X[pixel][k]
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* Ap = Ab + (wm * 32) * 32;
        const float* Wp = Ws + (wn * 32) * 32;
        float4 w0[2], xb[2];
        w0[0] = *reinterpret_cast<const float4*>(Wp + foff[0]);
        xb[0] = *reinterpret_cast<const float4*>(Ap + foff[0]);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int cur = it & 1, nxt = cur ^ 1;
            if (it < 15) {
                const int c = (it + 1) >> 2, q = (it + 1) & 3;
                w0[nxt] = *reinterpret_cast<const float4*>(Wp + c * (128 * 32) + foff[q]);
                xb[nxt] = *reinterpret_cast<const float4*>(Ap + c * (KC_BM * 32) + foff[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xv = e == 0 ? xb[cur].x : e == 1 ? xb[cur].y : e == 2 ? xb[cur].z : xb[cur].w;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e == 0 ? w0[cur].x : e == 1 ? w0[cur].y : e == 2 ? w0[cur].z : w0[cur].w, xv, acc, 0, 0, 0);
                if (e == 0 && it < 4 && more) {
                    __builtin_amdgcn_sched_barrier(0);
                    lds_dma16_s(xnext, avoff[it], dnext + (unsigned)(it * 1024));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const float r = rowstat[2 * (32 * wm + pl)], rm = rowstat[2 * (32 * wm + pl) + 1];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read its fragments of this tile: the buffer is free
        // ---- park k | v of the tile (LayerNorm fold applied) where x was: array wn = k h0 | k h1 | v h0 | v h1, row = pixel
        if (!(DDK_KVCTX_ABL & 8)) {
            float* T = Ab + wn * (KC_BM * 32);
            const int row = 32 * wm + pl, sw = (row >> 1) & 7;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = r * acc[4 * q] - rm * e1[q].x + e2[q].x;
                v.y = r * acc[4 * q + 1] - rm * e1[q].y + e2[q].y;
                v.z = r * acc[4 * q + 2] - rm * e1[q].z + e2[q].z;
                v.w = r * acc[4 * q + 3] - rm * e1[q].w + e2[q].w;
                *reinterpret_cast<float4*>(T + row * 32 + (((2 * q + h) ^ sw) << 2)) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(DDK_KVCTX_ABL & (8 | 2))) {   // column maxima of k over the tile's 64 pixels: thread = (column c of 64, row group of 8)
            const int c = tid & 63, ng = tid >> 6;
            const float* Tk = Ab + (c >> 5) * (KC_BM * 32);
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, Tk[tswz(ng + 8 * j, c & 31)]);
            cmx[ng * 64 + c] = m;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < 64 && !(DDK_KVCTX_ABL & (8 | 2))) {
            float mm = cmx[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) mm = fmaxf(mm, cmx[j * 64 + tid]);
            const float mo = Mx[tid], mn = fmaxf(mo, mm);
            Sc[tid] = __expf(mo - mn);                  // first tile: exp(-inf) = 0 against accumulators that are zero
            Mx[tid] = mn;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < ((DDK_KVCTX_ABL & (8 | 1)) ? 0 : 8); ++j) {                    // exp(k - M) in place
            const int idx = tid + j * 512;
            const int n = idx >> 6, c = idx & 63;
            float* q = Ab + (c >> 5) * (KC_BM * 32) + tswz(n, c & 31);
            *q = __expf(*q - Mx[c]);                     // hardware exp2 (1 ulp): the library expf is ~40 instructions, 8 per thread and tile
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(DDK_KVCTX_ABL & (8 | 4))) {   // ---- ctx^T += (exp k)^T v on the matrix pipe; a raised maximum first rescales what was accumulated
            const float* Tk = Ab + hh * (KC_BM * 32);
            const float* Tv = Ab + (2 + hh) * (KC_BM * 32);
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) macc[rr] *= Sc[hh * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh];
            dsum *= Sc[hh * 32 + l31];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int n = qr * 16 + 2 * s2 + kh;
                const float a = Tk[tswz(n, l31)];
                macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Tv[tswz(n, l31)], macc, 0, 0, 0);
                dsum += a;
            }
        }
        wait_vmcnt<0>();                                 // this wave's pieces of the next tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... everyone's; and the parked tile is consumed
    }
    // ---- the four pixel quarters of a head meet in LDS (both tile buffers are free), fixed order
    {
        float* pt = As;                                  // [8 waves][32 d][33]
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) pt[(wid * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh) * 33 + l31] = macc[rr];
        dsum += __shfl_xor(dsum, 32, 64);
        if (kh == 0) dpart[wid * DH + l31] = dsum;
        __syncthreads();
        const int oh = tid >> 8, d = (tid >> 3) & 31, e0 = (tid & 7) * 4;
        float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* q = pt + ((oh * 4 + w) * 32 + d) * 33 + e0;
            acc4.x += q[0]; acc4.y += q[1]; acc4.z += q[2]; acc4.w += q[3];
            den += dpart[(oh * 4 + w) * DH + d];
        }
        float* pp = p.part + (((long long)b * 4 + 2 * hp + oh) * p.splits + sp) * PART;
        if (e0 == 0) { pp[d] = Mx[oh * 32 + d]; pp[DH + d] = den; }
        *reinterpret_cast<float4*>(pp + 2 * DH + d * DH + e0) = acc4;
    }
}

// Core of the small-map kernels (256 threads): k (ROWS x 32, rows >= HW hold -inf), v (ROWS x 32), q (ROWS x 33 pitch) of one
// (b, head) are in LDS; column max of k, exp, ctx = (exp k)^T v / den, out = q ctx -- same arithmetic as the two kernels above.
template <int ROWS>
__device__ __forceinline__ void linattn_small_core(float* ks, float* vs, float* qs, float* cs, float* smax, float* __restrict__ ctx,
                                                   float* __restrict__ out, int b, int h, int HW, int heads) {
    const int HC = heads * DH;
    const int tid = threadIdx.x;
    {   // max_n k[n][d]  (rows >= HW hold -inf)
        const int d = tid & 31, ng = tid >> 5;
        float m = -INFINITY;
        for (int n = ng; n < ROWS; n += 8) m = fmaxf(m, ks[n * DH + d]);
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
#pragma unroll 8
    for (int j = 0; j < ROWS / 8; ++j) {                // exp(k - max) in place; padding rows -> exp(-inf) = 0
        const int i = tid + j * 256;
        ks[i] = __expf(ks[i] - smax[i & 31]);
    }
    __syncthreads();
    {
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float den = 0.f;
#pragma unroll 8
        for (int n = 0; n < ROWS; ++n) {
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
    // out[n][e] = sum_d ctx[d][e] q[n][d]: thread = (pixel n, 8 consecutive e), 64 pixels per pass
    for (int n = tid >> 2; n < ROWS; n += 64) {
        const int e0 = (tid & 3) * 8;
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

// The same core on the matrix pipe, for the 256-row case (16x16 maps) where the VALU form above is LDS-bound (it re-reads k, v,
// ctx per multiply: 2.5 MB of LDS traffic per workgroup): ctx = (exp k)^T v is a [32 x ROWS] x [ROWS x 32] product -- each of
// the 4 waves takes ROWS/4 rows (v_mfma_f32_32x32x2_f32, operands read from LDS one float per lane), the partial 32 x 32 blocks
// meet in LDS in wave order; out = q ctx is ROWS/32 blocks of [32 x 32] x [32 x 32], two per wave.  `part` = 4 x 32 x 33 floats.
typedef float f32x16_sm __attribute__((ext_vector_type(16)));
template <int ROWS>
__device__ __forceinline__ void linattn_core_mfma(float* ks, float* vs, float* qs, float* cs, float* smax, float* part,
                                                  float* __restrict__ ctx, float* __restrict__ out, int b, int h, int HW, int heads) {
    static_assert(ROWS % 128 == 0, "4 waves x multiples of 32 rows");
    const int HC = heads * DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, kk = lane >> 5;
    {   // max_n k[n][d]  (rows >= HW hold -inf)
        const int d = tid & 31, ng = tid >> 5;
        float m = -INFINITY;
        for (int n = ng; n < ROWS; n += 8) m = fmaxf(m, ks[n * DH + d]);
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
#pragma unroll 8
    for (int j = 0; j < ROWS / 8; ++j) {                // exp(k - max) in place; padding rows -> exp(-inf) = 0
        const int i = tid + j * 256;
        ks[i] = __expf(ks[i] - smax[i & 31]);
    }
    __syncthreads();
    {   // softmax denominators: den[d] = sum_n exp k[n][d], 8 strided partials per column summed in fixed order
        const int d = tid & 31, ng = tid >> 5;
        float t = 0.f;
        for (int n = ng; n < ROWS; n += 8) t += ks[n * DH + d];
        smax[DH + ng * DH + d] = t;                     // smax holds 8 * DH floats... use the area behind the maxima
    }
    // raw context of this wave's rows
    {
        constexpr int RPW = ROWS / 4;
        f32x16_sm acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* kp = ks + (wave * RPW + kk) * DH + c32;
        const float* vp = vs + (wave * RPW + kk) * DH + c32;
#pragma unroll 8
        for (int j = 0; j < RPW / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[j * 2 * DH], vp[j * 2 * DH], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * 33 + c32] = acc[r];
    }
    __syncthreads();
    {   // ctx[d][e] = (sum of the 4 partials) / den[d]: thread = (d, 4 consecutive e)
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        float den = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) den += smax[DH + g * DH + d];
        const float inv = 1.0f / den;
        float a4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = part[d * 33 + e0 + e];
#pragma unroll
            for (int w = 1; w < 4; ++w) t += part[(w * 32 + d) * 33 + e0 + e];
            a4[e] = t * inv;
            cs[d * (DH + 4) + e0 + e] = a4[e];
        }
        *reinterpret_cast<float4*>(ctx + (((long long)b * heads + h) * DH + d) * DH + e0) = make_float4(a4[0], a4[1], a4[2], a4[3]);
    }
    __syncthreads();
    // out[n][e] = sum_d q[n][d] ctx[d][e]: blocks of 32 pixels, ROWS / 128 blocks per wave
#pragma unroll
    for (int mb = 0; mb < ROWS / 128; ++mb) {
        const int n0 = (wave * (ROWS / 128) + mb) * 32;
        f32x16_sm acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* qp = qs + (n0 + c32) * (DH + 1) + kk;
        const float* cp = cs + kk * (DH + 4) + c32;
#pragma unroll
        for (int j = 0; j < DH / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * j], cp[2 * j * (DH + 4)], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (n < HW) out[((long long)b * HW + n) * HC + h * DH + c32] = acc[r];
        }
    }
}

// Small maps: context AND apply for one (b, head) in one workgroup -- k, v, q of the head (ROWS x 32 each) sit in LDS, so the
// separate merge and apply launches (and the re-read of q) disappear.  ROWS = 64 for the 8x8 and 4x4 levels (30 KB of LDS),
// 256 for the 16x16 level (100 KB).
template <int ROWS>
__global__ __launch_bounds__(256) void linattn_small_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, float* __restrict__ out,
                                                            int HW, int heads) {
    extern __shared__ __align__(16) float lsm[];
    float* ks = lsm;
    float* vs = ks + ROWS * DH;
    float* qs = vs + ROWS * DH;
    float* cs = qs + ROWS * (DH + 1);
    float* smax = cs + DH * (DH + 4);                   // 8 * DH maxima partials (+ 8 * DH denominator partials, + the context partials,
    float* part = smax + 16 * DH;                       //  for the matrix-pipe core)
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int HC = heads * DH, RS = 3 * HC;
    const float* base = qkv + (long long)b * HW * RS + h * DH;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS / 32; ++j) {               // ROWS rows x 8 float4 per operand
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
    if constexpr (ROWS >= 128) linattn_core_mfma<ROWS>(ks, vs, qs, cs, smax, part, ctx, out, b, h, HW, heads);
    else linattn_small_core<ROWS>(ks, vs, qs, cs, smax, ctx, out, b, h, HW, heads);
}

static size_t small_lds_bytes(int rows) {
    return ((size_t)rows * DH * 2 + (size_t)rows * (DH + 1) + DH * (DH + 4) + 16 * DH + (rows >= 128 ? 4 * 32 * 33 : 0)) * 4;
}

// The same with the projection inside: to_qkv (1x1 conv with the channel LayerNorm folded in, blocks.py:57-60, 123) for ONE head
// of ONE image is a [HW x C] x [C x 96] product -- small enough to run in the workgroup that consumes it, so the small maps need
// no to_qkv launch and no qkv tensor.  The image (<= 64 x C) sits in LDS; per-pixel mean and 1/(std + eps) come from it; the 4
// waves split the channel chunks (v_mfma_f32_16x16x4_f32, weights in operand order streamed L2 -> registers: qkv_operand_pack_kernel)
// and meet in LDS; r (W o g) x - r mean (W g) + W b lands in the k / v / q arrays of the core.
typedef float f32x4_att __attribute__((ext_vector_type(4)));
constexpr int QP_PITCH = 100;                           // partial-sum row pitch: 96 columns + 4

__host__ __device__ static inline size_t small_qkv_lds_bytes(int MB, int C) {
    const size_t xs = (size_t)16 * MB * (C + 4), part = (size_t)4 * 16 * MB * QP_PITCH;
    const size_t core = 64 * DH * 2 + 64 * (DH + 1) + DH * (DH + 4) + 8 * DH + 128 + 192;   // ks, vs, qs, cs, smax, rowstat, c1 | c2
    return ((xs > part ? xs : part) + core) * 4;
}

template <int MB>
__global__ __launch_bounds__(256) void linattn_small_qkv_kernel(const float* __restrict__ x, const float* __restrict__ wop,
                                                                const float* __restrict__ c1, const float* __restrict__ c2, float ln_eps,
                                                                float* __restrict__ ctx, float* __restrict__ out, int HW, int C, int heads) {
    extern __shared__ __align__(16) float sm[];
    constexpr int ROWS = 16 * MB;
    const int pitch = C + 4, nch = C >> 5;
    const size_t big = (size_t)ROWS * pitch > (size_t)4 * ROWS * QP_PITCH ? (size_t)ROWS * pitch : (size_t)4 * ROWS * QP_PITCH;
    float* xs = sm;                                     // the image; later the 4 waves' partial sums
    float* ks = sm + big;
    float* vs = ks + 64 * DH;
    float* qs = vs + 64 * DH;
    float* cs = qs + 64 * (DH + 1);
    float* smax = cs + DH * (DH + 4);
    float* rowstat = smax + 8 * DH;                     // [64][2]: r, r * mean
    float* cfold = rowstat + 128;                       // [96] W g and [96] W b of this head's q | k | v columns
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int HC = heads * DH;

    // weights of this wave's first chunk while the image loads
    const float* wl = wop + (size_t)h * nch * 3072 + lane * 4;          // a chunk = 6 n blocks x 512 floats
    float4 bA[6][2], bB[6][2];
    auto load_b = [&](int chunk, float4 (&bq)[6][2]) {
        chunk = chunk < nch ? chunk : nch - 1;
        const float* wp = wl + (size_t)chunk * 3072;
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) {
            bq[nb][0] = *reinterpret_cast<const float4*>(wp + nb * 512);
            bq[nb][1] = *reinterpret_cast<const float4*>(wp + nb * 512 + 256);
        }
    };
    load_b(wave, bA);
    load_b(wave + 4, bB);
    if (tid < 192) {                                    // the fold vectors of this head's 96 columns: one global load each, up front
        const int n = tid % 96;
        const int col = (n >> 5) * HC + h * DH + (n & 31);
        cfold[tid] = tid < 96 ? c1[col] : c2[col];
    }
    {
        const int q4 = C >> 2;
        for (int i = tid; i < ROWS * q4; i += 256) {
            const int row = i / q4, c = (i - row * q4) << 2;
            const float4 v = row < HW ? *reinterpret_cast<const float4*>(x + ((long long)b * HW + row) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(xs + row * pitch + c) = v;
        }
    }
    __syncthreads();
    {   // LayerNorm statistics of every pixel row: TPR threads per row (biased variance, eps added to the std: blocks.py:57-60)
        constexpr int TPR = 256 / ROWS;
        const int row = tid / TPR, sub = tid % TPR;
        // two passes over the row (it is resident in LDS): E[x^2] - mean^2 cancels catastrophically where |mean| >> std, which the
        // residual stream at the 4x4 bottleneck can reach; torch.var(unbiased=False) of the reference is two-pass too
        float s1 = 0.f;
        for (int c = sub * 4; c < C; c += TPR * 4) {
            const float4 v = *reinterpret_cast<const float4*>(xs + row * pitch + c);
            s1 += (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) s1 += __shfl_xor(s1, o, 64);
        const float inv_c = 1.0f / (float)C;
        const float mean = s1 * inv_c;
        float s2 = 0.f;
        for (int c = sub * 4; c < C; c += TPR * 4) {
            const float4 v = *reinterpret_cast<const float4*>(xs + row * pitch + c);
            const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
            s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) s2 += __shfl_xor(s2, o, 64);
        if (sub == 0) {
            const float var = s2 * inv_c;
            const float r = 1.0f / (sqrtf(var) + ln_eps);
            rowstat[2 * row] = r;
            rowstat[2 * row + 1] = r * mean;
        }
    }

    // ---- [ROWS x C] x [C x 96]: wave w takes chunks w, w + 4, ...
    f32x4_att acc[MB][6];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) acc[i][nb] = f32x4_att{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int chunk, const float4 (&bq)[6][2]) {
        float4 a[MB][2];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const float* ap = xs + (i * 16 + m) * pitch + (chunk << 5) + kq * 8;
            a[i][0] = *reinterpret_cast<const float4*>(ap);
            a[i][1] = *reinterpret_cast<const float4*>(ap + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const float av = reinterpret_cast<const float*>(&a[i][0])[kk];
#pragma unroll
                for (int nb = 0; nb < 6; ++nb)
                    acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, reinterpret_cast<const float*>(&bq[nb][0])[kk], acc[i][nb], 0, 0, 0);
            }
    };
    for (int c = wave; c < nch; c += 8) {
        compute(c, bA);
        load_b(c + 8, bA);
        if (c + 4 < nch) compute(c + 4, bB);
        load_b(c + 12, bB);
    }
    __syncthreads();                                    // image consumed, rowstat written
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int nb = 0; nb < 6; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[((wave * ROWS) + i * 16 + kq * 4 + r) * QP_PITCH + nb * 16 + m] = acc[i][nb][r];
    __syncthreads();
    // ---- sum of the 4 partials (fixed order) + folded LayerNorm -> q | k | v of the core (rows >= HW: k = -inf, q = v = 0)
    // (only the ROWS rows the core reads: on a 4x4 map that is 16, not the arrays' 64 -- every phase of this kernel is a dependent
    //  LDS pass behind a barrier, so padding rows are time, not just work)
    for (int e = tid; e < ROWS * 96; e += 256) {
        const int row = e / 96, n = e - row * 96;
        const int sel = n >> 5, d = n & 31;             // 0: q, 1: k, 2: v
        float v = sel == 1 ? -INFINITY : 0.f;
        if (row < HW) {
            float s = xs[row * QP_PITCH + n];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += xs[(w * ROWS + row) * QP_PITCH + n];
            v = rowstat[2 * row] * s - rowstat[2 * row + 1] * cfold[n] + cfold[96 + n];
        }
        if (sel == 0) qs[row * (DH + 1) + d] = v;
        else if (sel == 1) ks[row * DH + d] = v;
        else vs[row * DH + d] = v;
    }
    __syncthreads();
    linattn_small_core<ROWS>(ks, vs, qs, cs, smax, ctx, out, b, h, HW, heads);
}

// wop[head][chunk][n block 0..5][k half][lane][j] = lnw[o][i]: o = (nb / 2) * HC + head * 32 + (nb % 2) * 16 + lane % 16 (q, k, v
// columns of the head), i = 32 chunk + 8 (lane / 16) + 4 half + j; lnw = [3 HC][cp] (the LayerNorm-folded to_qkv weight)
__global__ __launch_bounds__(256) void qkv_operand_pack_kernel(const float* __restrict__ lnw, float* __restrict__ wop, int heads, int cp,
                                                               long long total) {
    const int nch = cp >> 5, HC = heads * DH;
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63), half = (int)((idx >> 8) & 1);
        long long r = idx >> 9;
        const int nb = (int)(r % 6); r /= 6;
        const int chunk = (int)(r % nch);
        const int head = (int)(r / nch);
        const int o = (nb >> 1) * HC + head * DH + (nb & 1) * 16 + (lane & 15);
        const int i = chunk * 32 + (lane >> 4) * 8 + half * 4 + j;
        wop[idx] = lnw[(long long)o * cp + i];
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
#ifdef DDK_TUNING
    const char* cap_env = getenv("DDK_LINATTN_WGS");
    const int cap = cap_env ? atoi(cap_env) : 1024;
#else
    const int cap = 1024;
#endif
    int s = cap / (B * heads);
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

int linattn_context(const float* qkv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, hipStream_t st,
                    bool kv_only) {
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
    hipLaunchKernelGGL(linattn_context_kernel, dim3(B * heads * s), dim3(256), 0, st, qkv, ctx, static_cast<float*>(workspace), HW, heads, s, rows,
                       kv_only ? 1 : 0);
    DDK_TRY(check_launch("linattn_context_kernel"));
    if (s > 1) {
        hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * heads), dim3(256), 0, st, static_cast<const float*>(workspace), ctx, s);
        DDK_TRY(check_launch("linattn_merge_kernel"));
    }
    return DDK_OK;
}

// context + apply in one launch when the map is small enough (<= 256 pixels) for one workgroup per (b, head)
int linattn_fused_small(const float* qkv, float* ctx, float* out, int B, int HW, int heads, hipStream_t st) {
    DDK_REQUIRE(qkv && ctx && out && B > 0 && HW > 0 && HW <= 256 && heads > 0, "linattn_fused_small: arguments (HW <= 256)");
    DDK_REQUIRE(aligned16(qkv) && aligned16(ctx) && aligned16(out), "linattn_fused_small: alignment");
    DDK_TRY(ensure_device_init());
    if (HW <= 64)
        hipLaunchKernelGGL(linattn_small_kernel<64>, dim3(B * heads), dim3(256), small_lds_bytes(64), st, qkv, ctx, out, HW, heads);
    else
        hipLaunchKernelGGL(linattn_small_kernel<256>, dim3(B * heads), dim3(256), small_lds_bytes(256), st, qkv, ctx, out, HW, heads);
    return check_launch("linattn_small_kernel");
}

bool linattn_small_qkv_ok(int HW, int C) { return HW > 0 && HW <= 64 && C % 32 == 0 && C >= 32 && small_qkv_lds_bytes(HW <= 16 ? 1 : 4, C) <= 160 * 1024; }

// ------------------------------------------------------------------------------------------------
// Folded attention output for maps with HW >> C (blocks.py:126-134 + to_out + the PreNorm LayerNorm of :57-60, C = hidden = 128).
// q is LINEAR in this attention (only k is soft-maxed), so with the LayerNorm folded into W_q (Wqg = W_q o g, c1q = W_q g,
// c2q = W_q b; r = 1 / (std + eps) per pixel):
//   y_n = W_out ctx^T q_n + b_out + x_n = r_n (A x_n) - r_n mean_n a1 + a2 + x_n,
//   A = W_out T,  T[h 32 + e][c] = sum_d ctx[h][d][e] Wqg[h 32 + d][c]            (one C x C matrix per IMAGE)
//   a1 = W_out (ctx^T c1q),  a2 = W_out (ctx^T c2q) + b_out
// i.e. to_qkv's q third, the apply kernel and to_out collapse into a 1x1 conv of x with per-image weights (2.6 M MACs per image
// to build A instead of (128 + 32) x 128 per PIXEL); conv1x1_ws_kernel<LN, RES> with per-image weights evaluates it.
// grid (B, 4) x 4 waves: workgroup (b, quarter) builds T (redundantly) and rows [32 quarter, +32) of A, both on the matrix pipe.
constexpr int FOLD_C = 128;
constexpr int FOLD_LDS_FLOATS = FOLD_C * FOLD_C + 32 * (FOLD_C + 1) + 2 * FOLD_C;
typedef float f32x16_att __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void attn_fold_kernel(const float* __restrict__ ctx, const float* __restrict__ wqg,
                                                        const float* __restrict__ c1q, const float* __restrict__ c2q,
                                                        const float* __restrict__ wout, const float* __restrict__ bout,
                                                        float* __restrict__ A, float* __restrict__ a1, float* __restrict__ a2) {
    extern __shared__ __align__(16) float sm[];
    float* T = sm;                              // [128 (h, e)][128 c]
    float* Wq = T + FOLD_C * FOLD_C;            // W_out rows of this quarter, [32 n][128 he + 1]: the A-phase row operand
    float* t1 = Wq + 32 * (FOLD_C + 1);         // ctx^T c1q, ctx^T c2q
    float* t2 = t1 + FOLD_C;
    const int b = blockIdx.x, nq = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    // Both products on the matrix pipe (v_mfma_f32_32x32x2_f32; VALU forms of this kernel took 19-34 us: one shared-operand read
    // per FMA).  T phase: wave = head h, T_h [32 e][128 c] = ctx_h^T [32 e][32 d] . Wqg_h [32 d][128 c]: 4 column blocks x 16 k-pairs.
    {
        const float* cp = ctx + ((long long)b * 4 + wave) * DH * DH;          // [d][e]: row operand (e = lane & 31, d = 2 s + kh)
        const float* wp = wqg + (long long)wave * DH * FOLD_C;                // [d][c]: column operand
        f32x16_att acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // all 80 operand words of the lane first (one memory latency for the phase, not one per unrolled batch), then the 64 MFMAs
        float av[DH / 2], bv[DH / 2][4];
#pragma unroll
        for (int s2 = 0; s2 < DH / 2; ++s2) {
            const int d = 2 * s2 + kh;
            av[s2] = cp[d * DH + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[s2][j] = wp[d * FOLD_C + j * 32 + l31];
        }
        __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise sinks every load to its MFMA: 4 in flight, one latency per k-pair
#pragma unroll
        for (int s2 = 0; s2 < DH / 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(wave * DH + (r & 3) + 8 * (r >> 2) + 4 * kh) * FOLD_C + j * 32 + l31] = acc[j][r];
    }
    for (int i = tid; i < 32 * FOLD_C; i += 256) Wq[(i >> 7) * (FOLD_C + 1) + (i & 127)] = wout[(long long)nq * 32 * FOLD_C + i];
    if (tid < FOLD_C) {
        const int h = tid >> 5, e = tid & 31;
        const float* ch = ctx + ((long long)b * 4 + h) * DH * DH;
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float cv = ch[d * DH + e];
            u1 += cv * c1q[h * DH + d];
            u2 += cv * c2q[h * DH + d];
        }
        t1[tid] = u1;
        t2[tid] = u2;
    }
    __syncthreads();
    {   // A phase: rows [32 nq, +32) x column block `wave`: A = W_out [32 n][128 he] . T [128 he][32 c], 64 k-pairs
        f32x16_att acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
        for (int s2 = 0; s2 < FOLD_C / 2; ++s2) {
            const int he = 2 * s2 + kh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wq[l31 * (FOLD_C + 1) + he], T[he * FOLD_C + wave * 32 + l31], acc, 0, 0, 0);
        }
        float* Ab = A + ((long long)b * FOLD_C + nq * 32) * FOLD_C + wave * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) Ab[((r & 3) + 8 * (r >> 2) + 4 * kh) * FOLD_C] = acc[r];
        if (tid < 32) {
            float u1 = 0.f, u2 = 0.f;
#pragma unroll 16
            for (int he = 0; he < FOLD_C; ++he) {
                u1 += Wq[tid * (FOLD_C + 1) + he] * t1[he];
                u2 += Wq[tid * (FOLD_C + 1) + he] * t2[he];
            }
            const int n = nq * 32 + tid;
            a1[(long long)b * FOLD_C + n] = u1;
            a2[(long long)b * FOLD_C + n] = u2 + (bout ? bout[n] : 0.f);
        }
    }
}

int linattn_small_qkv_init_device() {
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kvctx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(KC_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FOLD_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_small_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_small_qkv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_small_qkv_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return DDK_OK;
}

int qkv_operand_pack(const float* lnw, float* wop, int heads, int cp, hipStream_t st) {
    DDK_REQUIRE(lnw && wop && heads > 0 && cp > 0 && cp % 32 == 0, "qkv_operand_pack: arguments");
    const long long total = (long long)3 * heads * DH * cp;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(qkv_operand_pack_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, lnw, wop, heads, cp, total);
    return check_launch("qkv_operand_pack_kernel");
}

// to_qkv (LayerNorm folded) + context + apply of the small maps in one launch; x [B][HW][C], wop from qkv_operand_pack
int linattn_small_qkv(const float* x, const float* wop, const float* c1, const float* c2, float ln_eps, float* ctx, float* out, int B, int HW,
                      int C, int heads, hipStream_t st) {
    DDK_REQUIRE(x && wop && c1 && c2 && ctx && out && B > 0 && heads > 0, "linattn_small_qkv: arguments");
    DDK_REQUIRE(linattn_small_qkv_ok(HW, C), "linattn_small_qkv: needs H*W <= 64 and C % 32 == 0 (LDS budget)");
    DDK_REQUIRE(aligned16(x) && aligned16(wop) && aligned16(ctx) && aligned16(out), "linattn_small_qkv: alignment");
    DDK_TRY(ensure_device_init());
    if (HW <= 16)
        hipLaunchKernelGGL(linattn_small_qkv_kernel<1>, dim3(B * heads), dim3(256), small_qkv_lds_bytes(1, C), st, x, wop, c1, c2, ln_eps, ctx, out,
                           HW, C, heads);
    else
        hipLaunchKernelGGL(linattn_small_qkv_kernel<4>, dim3(B * heads), dim3(256), small_qkv_lds_bytes(4, C), st, x, wop, c1, c2, ln_eps, ctx, out,
                           HW, C, heads);
    return check_launch("linattn_small_qkv_kernel");
}

// ---- k, v projection + context in one launch (attn_kvctx_kernel)
bool attn_kvctx_ok(int B, int HW, int C, int heads) { return B > 0 && C == KC_K && heads == 4 && HW >= 4 * KC_BM && HW % KC_BM == 0; }
int attn_kvctx_splits(int B, int HW) {
    const int tiles = HW / KC_BM;
    int s = 1;
    while (s * 2 <= tiles && tiles % (s * 2) == 0 && (long long)B * 2 * (s * 2) <= 256) s *= 2;     // fill the chip, whole tiles per split
    return s;
}
size_t attn_kvctx_workspace_bytes(int B, int HW) { return (size_t)B * 4 * attn_kvctx_splits(B, HW) * PART * sizeof(float); }

// x [B*HW][128]; w_kv [256][128] = the k and v rows of the LayerNorm-folded to_qkv weight (W o g), c1 / c2 [256] = (W g, W b) of those
// rows -> ctx [B][4][32][32] (softmax over the pixels of k, then k v^T), bit-stable; workspace: attn_kvctx_workspace_bytes
int attn_kvctx(const float* x, const float* w_kv, const float* c1, const float* c2, float ln_eps, float* ctx, int B, int HW, void* workspace,
               size_t workspace_bytes, hipStream_t st) {
    DDK_REQUIRE(x && w_kv && c1 && c2 && ctx && workspace, "attn_kvctx: null pointer");
    DDK_REQUIRE(attn_kvctx_ok(B, HW, KC_K, 4), "attn_kvctx: needs 128 channels, 4 heads, H*W a multiple of 64 and >= 256");
    DDK_REQUIRE(aligned16(x) && aligned16(w_kv) && aligned16(c1) && aligned16(c2) && aligned16(ctx) && aligned16(workspace), "attn_kvctx: alignment");
    DDK_REQUIRE((long long)B * HW * KC_K * 4 < (1LL << 32) && B <= 65535, "attn_kvctx: input of 4 GiB or more, or more than 65535 images");
    const int s = attn_kvctx_splits(B, HW);
    if (workspace_bytes < attn_kvctx_workspace_bytes(B, HW)) {
        set_error("attn_kvctx: workspace too small (%zu < %zu)", workspace_bytes, attn_kvctx_workspace_bytes(B, HW));
        return DDK_ERR_WORKSPACE;
    }
    DDK_TRY(ensure_device_init());
    KvCtxParams p{x, w_kv, c1, c2, ln_eps, static_cast<float*>(workspace), HW / KC_BM, HW / KC_BM / s, s};
    hipLaunchKernelGGL(attn_kvctx_kernel, dim3((unsigned)s, 2, (unsigned)B), dim3(512), KC_LDS_FLOATS * sizeof(float), st, p);
    DDK_TRY(check_launch("attn_kvctx_kernel"));
    hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * 4), dim3(256), 0, st, static_cast<const float*>(workspace), ctx, s);
    return check_launch("linattn_merge_kernel");
}

bool attn_fold_ok(int C, int heads) { return C == FOLD_C && heads * DH == FOLD_C; }
size_t attn_fold_out_floats(int B) { return (size_t)B * (FOLD_C * FOLD_C + 2 * FOLD_C); }

// ctx [B][4][32][32]; wqg [128][128] = rows 0..127 of the LayerNorm-folded to_qkv weight, c1q / c2q its fold vectors; wout [128][128]
// the packed to_out weight, bout its bias -> A [B][128][128], a1, a2 [B][128]
int attn_fold(const float* ctx, const float* wqg, const float* c1q, const float* c2q, const float* wout, const float* bout, float* A,
              float* a1, float* a2, int B, int C, int heads, hipStream_t st) {
    DDK_REQUIRE(ctx && wqg && c1q && c2q && wout && A && a1 && a2 && B > 0, "attn_fold: arguments");
    DDK_REQUIRE(attn_fold_ok(C, heads), "attn_fold: C == 128 and heads * 32 == 128 only");
    DDK_REQUIRE(aligned16(ctx) && aligned16(wout) && aligned16(A), "attn_fold: alignment");
    DDK_TRY(ensure_device_init());
    hipLaunchKernelGGL(attn_fold_kernel, dim3(B, 4), dim3(256), FOLD_LDS_FLOATS * sizeof(float), st, ctx, wqg, c1q, c2q, wout, bout, A, a1, a2);
    return check_launch("attn_fold_kernel");
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
    return ddk::linattn_context(qkv, ctx, B, HW, heads, workspace, workspace_bytes, ddk::as_stream(s), false);
}
int ddk_linattn_context_kv(const float* kv, float* ctx, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    return ddk::linattn_context(kv, ctx, B, HW, heads, workspace, workspace_bytes, ddk::as_stream(s), true);
}
size_t ddk_attention_kv_context_workspace_bytes(int B, int HW) { return ddk::attn_kvctx_workspace_bytes(B, HW); }
int ddk_attention_kv_context_ok(int B, int HW, int C, int heads) { return ddk::attn_kvctx_ok(B, HW, C, heads) ? 1 : 0; }
int ddk_attention_kv_context(const float* x, const float* w_kv, const float* c1, const float* c2, float ln_eps, float* ctx, int B, int HW,
                             void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    return ddk::attn_kvctx(x, w_kv, c1, c2, ln_eps, ctx, B, HW, workspace, workspace_bytes, ddk::as_stream(s));
}
int ddk_attention_fold(const float* ctx, const float* wqg, const float* c1q, const float* c2q, const float* wout, const float* bout, float* A,
                       float* a1, float* a2, int B, int C, int heads, ddk_stream_t s) {
    return ddk::attn_fold(ctx, wqg, c1q, c2q, wout, bout, A, a1, a2, B, C, heads, ddk::as_stream(s));
}
int ddk_linattn_fused_small(const float* qkv, float* ctx, float* out, int B, int HW, int heads, ddk_stream_t s) {
    return ddk::linattn_fused_small(qkv, ctx, out, B, HW, heads, ddk::as_stream(s));
}
int ddk_pack_qkv_operand(const float* folded_w, float* dst, int heads, int c_pad, ddk_stream_t s) {
    return ddk::qkv_operand_pack(folded_w, dst, heads, c_pad, ddk::as_stream(s));
}
int ddk_linattn_small_from_x(const float* x, const float* w_operand, const float* c1, const float* c2, float ln_eps, float* ctx, float* out,
                             int B, int HW, int C, int heads, ddk_stream_t s) {
    return ddk::linattn_small_qkv(x, w_operand, c1, c2, ln_eps, ctx, out, B, HW, C, heads, ddk::as_stream(s));
}
int ddk_linattn_apply(const float* qkv, const float* ctx, float* out, int B, int HW, int heads, ddk_stream_t s) {
    return ddk::linattn_apply(qkv, ctx, out, B, HW, heads, ddk::as_stream(s));
}
}
