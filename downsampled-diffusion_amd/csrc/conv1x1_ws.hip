// conv1x1_ws.hip -- 1x1 conv with K = 128 input channels as a WEIGHTS-STATIONARY, pixel-streaming GEMM (gfx950).
//
// Replaces the nn.Conv2d(k=1) dispatches of reference models/unet/blocks.py:103 (res_conv), :123 (to_qkv, behind the channel
// LayerNorm of :57-60 -- folded in as in conv_igemm.hip) and :124 (to_out, with the Residual add of :13-14) on the large maps
// (32x32, 16x16) where the input has 128 channels.
//
// Why: these layers have a short contraction (K = 128 = 4 chunks of 32).  The im2col kernel gives every 64x64 output tile its own
// workgroup: 3072 workgroups for to_qkv at 32x32, each loading its 16 KB slice of the weights, running 4 chunks between barriers
// and transposing its tile through LDS -- 42 us where the MFMAs need 20.5 (profiles/r03_sampler_step_breakdown.txt).  Here a
// workgroup keeps a 128-channel slice of W (all of K: 64 KB) in LDS for its whole life and streams 64-pixel tiles of the input
// through a double buffer: no barrier inside the contraction, 128 back-to-back MFMAs per wave and tile, the next tile's DMA
// under them.
//
//   workgroup g of slice s: output channels [128 s, 128 s + 128), pixel tiles g, g + G, g + 2G, ... (G workgroups per slice)
//   LDS: W 4 chunk images [128 n][32 k] (64 KB) + 2 x A 4 chunk images [64 pixels][32 k] (2 x 32 KB) + row statistics;
//        128-byte rows, k-chunk position XOR-swizzled by (row >> 1) & 7 on the DMA source address (as everywhere in this library)
//   MFMA: v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (rows = channels) and the pixels as B: a lane ends up with 4
//        consecutive output channels of its pixel per accumulator quad -> float4 NHWC stores straight from registers, no LDS
//        transpose in the epilogue (conv_first.hip's orientation); wave = 32 pixels x 32 channels, 8 waves = 64 x 128, two per SIMD
//   vmcnt: the DMA pieces and the residual loads are inline asm, counted by hand (hipcc cannot see them): per tile a wave issues
//        [4 residual loads] [4 DMA pieces of the next tile] ... [4 stores]; "residual landed" = vmcnt(4), "next tile landed" = vmcnt(4)
//        behind the stores.
#include "conv_common.h"

namespace ddk {

constexpr int WS_K = 128, WS_BN = 128, WS_BM = 64;
constexpr int WS_W_FLOATS = WS_BN * WS_K, WS_A_FLOATS = WS_BM * WS_K;
constexpr int WS_LDS_FLOATS = WS_W_FLOATS + 2 * WS_A_FLOATS + 2 * WS_BM;

struct WsParams {
    const float* x;       // [M][128]
    const float* w;       // [N][128]
    const float* bias;    // [N] or null
    const float* resid;   // [M][N] or null
    const float* ln_c1;   // LayerNorm fold (conv_igemm.hip): w holds W o g, ln_c1 = W g, ln_c2 = W b; null = plain conv
    const float* ln_c2;
    float ln_eps;
    float* out;           // [M][N]
    int N, MT, G;         // output channels, 64-pixel tiles, workgroups per 128-channel slice
    long long resid_bytes;
    int tpi;              // > 0: PER-IMAGE weights (N == 128): w is [images][128][128], ln_c1 / ln_c2 / bias [images][128], an image is
                          // tpi consecutive tiles and "slice" below is the image index (the folded attention output, attention.hip)
};

typedef int ws_i32x4 __attribute__((ext_vector_type(4)));
typedef float ws_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ws_buf_load16(ws_f32x4& d, unsigned voff, const ws_i32x4& srd, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}

template <bool LN, bool RES>
__global__ __launch_bounds__(512) void conv1x1_ws_kernel(const WsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                                   // [4 chunks][128 rows][32]
    float* As = smem + WS_W_FLOATS;                     // [2 buffers][4 chunks][64 rows][32]
    float* rowstat = As + 2 * WS_A_FLOATS;              // [64][2]: (r, r * mean) of the tile's pixels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x / p.G, g = blockIdx.x - slice * p.G;
    const bool per_img = p.tpi > 0;
    const int n0 = per_img ? 0 : slice * WS_BN;
    const int vec0 = per_img ? slice * WS_BN : 0;       // offset of this workgroup's per-channel vectors (bias, LayerNorm fold)
    const int tile_end = per_img ? (slice + 1) * p.tpi : p.MT;
    const int wm = wid & 1, wn = wid >> 1;              // this wave: pixels [32 wm, +32), channels [32 wn, +32) of the slice
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;

    // ---- DMA: wave w moves half (w >> 2) of k-chunk (w & 3) of both operands; a piece = 8 rows x 128 bytes, lane = (row lane / 8,
    // 16-byte position lane % 8)
    const int prow = lane >> 3, ppos = lane & 7;
    const int dchunk = wid & 3, dhalf = wid >> 2;
    unsigned wvoff[8], avoff[4];                        // byte offset of this lane's 16 bytes in its pieces of a [rows][128] operand
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (dhalf * 8 + j) * 8 + prow;
        wvoff[j] = (unsigned)((r * WS_K + dchunk * 32 + ((ppos ^ ((r >> 1) & 7)) << 2)) * 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (dhalf * 4 + j) * 8 + prow;
        avoff[j] = (unsigned)((r * WS_K + dchunk * 32 + ((ppos ^ ((r >> 1) & 7)) << 2)) * 4);
    }
    {   // the weight slice: 8 pieces per wave, once
        const float* wb = p.w + (long long)(per_img ? slice * WS_BN : n0) * WS_K;
        const unsigned dst = lds_base + (unsigned)((dchunk * (WS_BN * 32) + dhalf * 2048) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_dma16_s(wb, wvoff[j], dst + (unsigned)(j * 1024));
    }
    const unsigned a_dst = lds_base + (unsigned)((WS_W_FLOATS + dchunk * (WS_BM * 32) + dhalf * 1024) * 4);
    auto issue_a = [&](int tile, int buf) {             // 4 pieces per wave
        const float* xb = p.x + (long long)tile * WS_BM * WS_K;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_s(xb, avoff[j], a_dst + (unsigned)(buf * WS_A_FLOATS * 4 + j * 1024));
    };
    int tile = (per_img ? slice * p.tpi : 0) + g;
    if (tile < tile_end) issue_a(tile, 0);

    // ---- per-wave constants of the epilogue: this lane's 4 channel quads: channel n0 + 32 wn + 8 q + 4 h
    const int pl = lane & 31, h = lane >> 5;
    const int cbase = n0 + 32 * wn + 4 * h;
    float4 e1[4], e2[4];                                // LN: (c1, c2 + bias); plain: (unused, bias)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = vec0 + cbase + 8 * q;
        float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (LN) {
            e1[q] = *reinterpret_cast<const float4*>(p.ln_c1 + c);
            const float4 c2 = *reinterpret_cast<const float4*>(p.ln_c2 + c);
            e2[q] = make_float4(c2.x + b4.x, c2.y + b4.y, c2.z + b4.z, c2.w + b4.w);
        } else {
            e1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            e2[q] = b4;
        }
    }
    ws_i32x4 rsrd;                                       // residual: raw buffer descriptor over [M][N]
    {
        const unsigned long long a = reinterpret_cast<unsigned long long>(RES ? p.resid : p.x);
        rsrd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rsrd.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFFu));
        rsrd.z = __builtin_amdgcn_readfirstlane((int)(unsigned)p.resid_bytes);
        rsrd.w = 0x00020000;
    }
    // fragment addressing: row = block base + (lane & 31); k-chunk position 2q + (lane >> 5), swizzled by the row
    const int fsw = (pl >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = pl * 32 + (((2 * q + h) ^ fsw) << 2);

    wait_vmcnt<0>();                                    // weights + first tile landed (and the epilogue constants)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int buf = 0;
    for (; tile < tile_end; tile += p.G, buf ^= 1) {
        const float* Ab = As + buf * WS_A_FLOATS;
        const long long m0 = (long long)tile * WS_BM;
        if (LN) {
            // LayerNorm statistics of the tile's 64 pixel rows, two passes over the resident row (4 threads per row, one k-chunk
            // image each; the swizzle only permutes a row's 16-byte positions)
            const int row = tid >> 3, part = tid & 7;   // 8 threads per row: half (part & 1) of k-chunk image (part >> 1)
            const float4* rp = reinterpret_cast<const float4*>(Ab + (part >> 1) * (WS_BM * 32) + row * 32);
            float4 v[4];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {            // position rotated by the row pair: the 64 lanes of one read cover all banks
                v[i] = rp[(4 * (part & 1) + i + (row >> 1)) & 7];
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.0f / WS_K);
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                qq += (a * a + b * b) + (c * c + d * d);
            }
            qq += __shfl_xor(qq, 1, 64);
            qq += __shfl_xor(qq, 2, 64);
            qq += __shfl_xor(qq, 4, 64);
            if (part == 0) {
                const float r = 1.0f / (sqrtf(qq * (1.0f / WS_K)) + p.ln_eps);   // eps on the std (blocks.py:58-60)
                rowstat[2 * row] = r;
                rowstat[2 * row + 1] = r * mean;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // residual of this lane's 4 output quads, requested first (older than the DMA pieces below)
        const unsigned row_off = (unsigned)(((m0 + 32 * wm + pl) * p.N + cbase) * 4);
        ws_f32x4 rr[4];
        if (RES) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ws_buf_load16(rr[q], row_off, rsrd, (unsigned)(8 * q * 4));
        }
        const bool more = tile + p.G < tile_end;         // wave-uniform
        const float* xnext = p.x + (long long)(tile + p.G) * WS_BM * WS_K;
        const unsigned dnext = a_dst + (unsigned)((buf ^ 1) * WS_A_FLOATS * 4);

        // ---- 64 MFMAs per wave: D[channel][pixel] += W[channel][k] X[pixel][k], no barrier inside.  The fragments of step
        // it + 1 are requested before the MFMAs of step it, and the next tile's 4 DMA pieces ride behind the first MFMA of the
        // first 4 steps; two waves per SIMD cover each other's waits.
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
                w0[nxt] = *reinterpret_cast<const float4*>(Wp + c * (WS_BN * 32) + foff[q]);
                xb[nxt] = *reinterpret_cast<const float4*>(Ap + c * (WS_BM * 32) + foff[q]);
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
        // ---- epilogue: accumulator register 4q + i = channel cbase + 8 q + i of pixel pl
        float r = 1.f, rm = 0.f;
        if (LN) { r = rowstat[2 * (32 * wm + pl)]; rm = rowstat[2 * (32 * wm + pl) + 1]; }
        if (RES) {
            if (more) wait_vmcnt<4>(); else wait_vmcnt<0>();          // the 4 residual loads (the next tile's 4 pieces may still fly)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(rr[q]));
        }
        float* orow = p.out + (m0 + 32 * wm + pl) * p.N + cbase;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            if (LN) {
                v.x = r * v.x - rm * e1[q].x + e2[q].x; v.y = r * v.y - rm * e1[q].y + e2[q].y;
                v.z = r * v.z - rm * e1[q].z + e2[q].z; v.w = r * v.w - rm * e1[q].w + e2[q].w;
            } else {
                v.x += e2[q].x; v.y += e2[q].y; v.z += e2[q].z; v.w += e2[q].w;
            }
            if (RES) { v.x += rr[q].x; v.y += rr[q].y; v.z += rr[q].z; v.w += rr[q].w; }
            *reinterpret_cast<float4*>(orow + 8 * q) = v;
        }
        // the next tile has landed (this wave's pieces; the 4 stores just issued are younger) -- then everyone's, and everyone is
        // done with this tile's buffer and row statistics
        if (more) {
            wait_vmcnt<4>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
}

bool conv1x1_ws_ok(long long M, int K, int N) {
    return K == WS_K && N > 0 && N % WS_BN == 0 && M >= 2048 && M % WS_BM == 0 && M * (long long)(N > K ? N : K) * 4 < (1LL << 31);
}

int conv1x1_ws_init_device() {
    const int bytes = (int)(WS_LDS_FLOATS * sizeof(float));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return DDK_OK;
}

// x [M][128], w [N][128] (the packed 1x1 weight: row pitch 128), out [M][N].  ln != nullptr: w holds W o g and (c1, c2) = (W g, W b).
int conv1x1_ws(const float* x, const float* w, const float* bias, const float* resid, float* out, long long M, int N, const ConvLnFold* ln,
               hipStream_t st, int images) {
    DDK_REQUIRE(x && w && out, "conv1x1_ws: null pointer");
    DDK_REQUIRE(conv1x1_ws_ok(M, WS_K, N), "conv1x1_ws: needs K == 128, N % 128 == 0, M % 64 == 0, M >= 2048");
    DDK_REQUIRE(images == 0 || (images > 0 && N == WS_BN && M % images == 0 && (M / images) % WS_BM == 0),
                "conv1x1_ws: per-image weights need N == 128 and whole 64-pixel tiles per image");
    DDK_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out) && aligned16(bias) && aligned16(resid), "conv1x1_ws: alignment");
    DDK_REQUIRE(!ln || (ln->c1 && ln->c2 && aligned16(ln->c1) && aligned16(ln->c2)), "conv1x1_ws: LayerNorm folding vectors");
    DDK_TRY(ensure_device_init());
    WsParams p{};
    p.x = x; p.w = w; p.bias = bias; p.resid = resid; p.out = out;
    p.ln_c1 = ln ? ln->c1 : nullptr; p.ln_c2 = ln ? ln->c2 : nullptr; p.ln_eps = ln ? ln->eps : 0.f;
    p.N = N; p.MT = (int)(M / WS_BM);
    const int NS = images > 0 ? images : N / WS_BN;     // weight slices: 128-channel slices of one filter, or one filter per image
    p.tpi = images > 0 ? (int)(M / images / WS_BM) : 0;
    int G = 256 / NS;                                   // one workgroup per CU (132 KB of LDS)
    if (G < 1) G = 1;
    if (G > (images > 0 ? p.tpi : p.MT)) G = images > 0 ? p.tpi : p.MT;
    p.G = G;
    p.resid_bytes = M * N * 4;
    const dim3 grid((unsigned)(NS * G));
    const size_t lds = WS_LDS_FLOATS * sizeof(float);
    if (ln && resid) hipLaunchKernelGGL((conv1x1_ws_kernel<true, true>), grid, dim3(512), lds, st, p);
    else if (ln) hipLaunchKernelGGL((conv1x1_ws_kernel<true, false>), grid, dim3(512), lds, st, p);
    else if (resid) hipLaunchKernelGGL((conv1x1_ws_kernel<false, true>), grid, dim3(512), lds, st, p);
    else hipLaunchKernelGGL((conv1x1_ws_kernel<false, false>), grid, dim3(512), lds, st, p);
    return check_launch("conv1x1_ws_kernel");
}

}  // namespace ddk

extern "C" int ddk_conv1x1_ws_ok(long long M, int K, int N) { return ddk::conv1x1_ws_ok(M, K, N) ? 1 : 0; }

extern "C" int ddk_conv1x1_ws(const float* x, const float* w, const float* bias, const float* resid, float* out, long long M, int N,
                              const float* ln_c1, const float* ln_c2, float ln_eps, ddk_stream_t s) {
    const ddk::ConvLnFold ln{ln_c1, ln_c2, ln_eps};
    return ddk::conv1x1_ws(x, w, bias, resid, out, M, N, ln_c1 ? &ln : nullptr, ddk::as_stream(s));
}

extern "C" int ddk_conv1x1_ws_images(const float* x, const float* w, const float* bias, const float* resid, float* out, long long M, int N,
                                     const float* ln_c1, const float* ln_c2, float ln_eps, int images, ddk_stream_t s) {
    const ddk::ConvLnFold ln{ln_c1, ln_c2, ln_eps};
    return ddk::conv1x1_ws(x, w, bias, resid, out, M, N, ln_c1 ? &ln : nullptr, ddk::as_stream(s), images);
}
