// conv_local.hip -- conv3x3 + GroupNorm + Mish (+ time shift) (+ residual) in ONE launch for the small maps (4x4, 8x8).
//
// Reference: Block = Conv2d(3, padding=1) -> GroupNorm(8) -> Mish (models/unet/blocks.py:75-84), ResnetBlock adds the time
// shift after the first Block and the residual after the second (blocks.py:105-115).
//
// Why a separate kernel: on a 4x4 map a batch of 32 images is 512 pixels.  Split over channel chunks the conv fills the
// chip but leaves slabs that a second launch (GroupNorm) has to sum -- and one dependent launch costs ~6 us whatever it
// does (profiles/r02_batch_scaling.txt: GroupNorm takes the same time at batch 16 and 32).  Here the tiling is chosen
// so that the GroupNorm reduction domain lies INSIDE one workgroup: a workgroup owns one image x 32 output channels
// (whole groups of 8, 16 or 32 channels), its 8 waves split k = 9 taps x C_in between them, the partial accumulators
// meet in LDS, and statistics, affine, Mish, shift and residual happen on the way out.  No slabs, no second launch.
//
// Arithmetic: direct convolution on v_mfma_f32_16x16x4_f32 (a 16-pixel image is exactly one M block).  The activated
// input image (HW x C_in, <= 134 KB) sits in LDS with one all-zero row that out-of-image taps read; weights stream
// straight into registers, three (tap, chunk) units in flight per wave.  Their layout (ddk_pack_conv_weight_local) is the
// MFMA B-operand order itself: [n tile][tap][chunk][n block][k half][lane][4], so one load instruction reads 1 KB of
// consecutive memory.  (First version: the [o][tap][i] layout of the other kernels, lane = 16 B out of a row 9*C_in*4 bytes
// from its neighbour's -- every quarter-wave touched 16 cache lines and the CU's vector-memory pipe, not the MFMA, set the
// pace: 35 GB/s per CU, 0.9 us per unit.)
// grid = B * N/32 with the n tile fastest: workgroup ids that share an n tile's 295 KB of weights land on the same XCD
// (ids round-robin over the 8 XCDs), so each L2 holds 1/8 of the filter.
#include <type_traits>

#include "ddk_internal.h"

namespace ddk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LocalParams {
    const float* src0;
    const float* src1;
    int c0, c1;
    const float* w;       // ddk_pack_conv_weight_local layout
    const float* bias;    // [N] or null
    const float* gamma;
    const float* beta;
    const float* temb;    // [rows][temb_stride] shift added after Mish, or null
    int temb_stride;
    const long long* temb_rows;
    const float* addend;  // [B][HW][N] residual added last, or null
    float* out;           // [B][HW][N]
    int H, W, N, cpg;
    float eps;
    int addend_slabs;            // > 1: `addend` is still in split-K form: that many slabs, `addend_stride` floats apart, summed in
    long long addend_stride;     // order, plus addend_bias[c] (the skip conv's reduce pass folded into this load)
    const float* addend_bias;
    int ipb;                     // images per 16-row block: 1 (4x4 maps) or 4 (2x2 maps: four 4-pixel images share an M block)
    int src_slabs;               // > 1: `src0` is still in split-K form (the stride-2 conv in front left its slabs): that many slabs,
    long long src_stride;        // `src_stride` floats apart, summed in slab order, plus src_bias[c] -- the arithmetic and the order of
    const float* src_bias;       // splitk_reduce_kernel, done while the image is staged (no reduce launch, no reduced tensor)
};

// One float4 of a staged source row: channels [c, c + 4) of pixel row `r` (global row index), from src0 -- plain, or the fixed-order
// sum of its split-K slabs plus the producing conv's bias -- or from the concat's second source.
template <typename P>
__device__ __forceinline__ float4 staged_src4(const P& p, long long r, int c) {
    if (c >= p.c0) return *reinterpret_cast<const float4*>(p.src1 + r * p.c1 + (c - p.c0));
    const float* s0 = p.src0 + r * p.c0 + c;
    float4 v = *reinterpret_cast<const float4*>(s0);
    if (p.src_slabs > 1) {
        // four slab loads in flight at a time (clamped index, no load under a condition), added in slab order: a workgroup has nothing else
        // to hide these behind -- issued one by one (first version) eight slabs cost the 4x4 Block 5.9 us, more than the reduce launch
        for (int sl = 1; sl < p.src_slabs; sl += 4) {
            float4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = sl + j < p.src_slabs ? sl + j : p.src_slabs - 1;
                t[j] = *reinterpret_cast<const float4*>(s0 + q * p.src_stride);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sl + j < p.src_slabs) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
        }
        if (p.src_bias) {
            const float4 bb = *reinterpret_cast<const float4*>(p.src_bias + c);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
    }
    return v;
}

constexpr int LOC_PP = 36;   // pitch (floats) of a partial-accumulator row: 4 rows apart = 16 banks apart

__host__ __device__ static inline size_t local_lds_bytes(int MT, int cin) {
    const size_t a = (size_t)(MT + 1) * (cin + 4) * 4;
    const size_t p = (size_t)8 * MT * LOC_PP * 4;
    return (a > p ? a : p) + 256;
}

// Shared tail of the image-local kernels.  Every thread holds NV conv outputs (bias added) of ONE output channel c = n0 + col
// of image b; lane = (other index bit) * 32 + col.  GroupNorm statistics of (image, group of col): two passes over the
// registers, like torch's native_group_norm.  Wave level: lanes that share the group (xor masks below cpg, and 32); workgroup
// level: the 8 waves' sums in fixed order.  Then affine, Mish, time shift, residual, store.
// What the tail needs from global memory besides the conv result -- affine, time shift, residual -- requested at the START of
// the kernel (tail_prefetch), so that their latency (two dependent loads for the shift: temb_rows[b], then the row) hides behind
// the k loop instead of standing at the very end of every launch.
template <int NV>
struct TailPre {
    float ga, be, sh;
    float add[NV];
};
template <int NV, typename P>
__device__ __forceinline__ TailPre<NV> tail_prefetch(const long long (&o)[NV], int c, int b, const P& p) {
    TailPre<NV> t;
    t.ga = p.gamma[c];
    t.be = p.beta[c];
    t.sh = 0.f;
    if (p.temb) {
        const long long tr = p.temb_rows ? p.temb_rows[b] : b;
        t.sh = p.temb[tr * p.temb_stride + c];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float r = 0.f;
        if (p.addend) {
            r = p.addend[o[i]];
            for (int sl = 1; sl < p.addend_slabs; sl += 4) {       // four loads in flight, added in slab order
                float u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = p.addend[(sl + j < p.addend_slabs ? sl + j : p.addend_slabs - 1) * p.addend_stride + o[i]];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (sl + j < p.addend_slabs) r += u[j];
            }
            if (p.addend_bias) r += p.addend_bias[c];
        }
        t.add[i] = r;
    }
    return t;
}

template <int NV, typename P>
__device__ __forceinline__ void gn_mish_tail(const float (&v)[NV], const long long (&o)[NV], int col, int c, int b, int hw, int lane, int wave,
                                             float* red, const P& p, const TailPre<NV>& pre, int ipb = 1) {
    const int gl = col / p.cpg;                 // group within the tile: 0 .. 32/cpg - 1
    // Sum over the lanes of the group (cpg consecutive columns, both 32-lane halves), then over the 8 waves through `rd` (two
    // disjoint regions for the two passes, so one barrier per pass).  Inside a 16-lane row the tree runs on DPP (quad permutes,
    // row_half_mirror, row_mirror: any pairing that crosses the halves adds the same numbers in every lane), only the steps
    // across rows go through the LDS crossbar.
    auto dpp_add = [](float v, auto ctrl) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    auto group_sum = [&](float s, float* rd) {
        s = dpp_add(s, std::integral_constant<int, 0xB1>{});                          // lanes 1 apart
        s = dpp_add(s, std::integral_constant<int, 0x4E>{});                          // 2 apart
        if (p.cpg >= 8) s = dpp_add(s, std::integral_constant<int, 0x141>{});         // the other quad of the 8-lane group
        if (p.cpg >= 16) s = dpp_add(s, std::integral_constant<int, 0x140>{});        // the other half of the 16-lane row
        if (p.cpg >= 32) s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if ((lane & 32) == 0 && (col & (p.cpg - 1)) == 0) rd[wave * 4 + gl] = s;
        __syncthreads();
        if (ipb == 4) return rd[(wave & ~1) * 4 + gl] + rd[(wave | 1) * 4 + gl];   // 4 images x 4 rows: an image = 2 waves' rows
        float t = rd[gl];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += rd[w * 4 + gl];
        return t;
    };
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    const float inv_n = 1.0f / (float)(hw * p.cpg);
    const float mean = group_sum(s, red) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float var = group_sum(q, red + 32) * inv_n;
    const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float y = mish_f((v[i] - mean) * rstd * pre.ga + pre.be) + pre.sh;
        if (p.addend) y += pre.add[i];
        p.out[o[i]] = y;
    }
}

#ifdef DDK_TUNING
// Diagnostic stamps of conv3x3_gn_wlocal_kernel (tuning build only): per workgroup, in shader cycles (s_memtime) --
// [0] entry, [1] image staged, [2] k loop done, [3] end of kernel, [4] cycles matrix wave 0 spent parked at the chunk barriers,
// [5] cycles transform wave 8 spent transforming, [6] chunks, [7] valid.  Written to a buffer nothing else reads.
__device__ unsigned long long g_wl_stamps[8 * 512];
#define WL_STAMP(x) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); } while (0)
#else
#define WL_STAMP(x) do { } while (0)
#endif

// NU > 0: C_in = 256 NU, so that wave w's units are (tap k / NU, chunk w + 8 (k % NU)), k = 0 .. 9 NU - 1 -- the unit loop unrolls
// with compile-time taps and the LDS row of a tap's source pixel is one of nine per-lane registers computed once (per unit those selects
// were ~10 VALU instructions per 16 MFMAs, and VALU work between fp32 MFMAs is matrix time: tools/local_clock.py, k loop 78 % busy)
template <int MT, int IPB = 1, int NU = 0>     // IPB images per block: 1, or 4 on 2x2 maps (MT == 16)
__global__ __launch_bounds__(512) void conv3x3_gn_local_kernel(const LocalParams p) {
    extern __shared__ __align__(16) float lds[];
    constexpr int MB = MT / 16;
    unsigned long long st_entry = 0, st_entry_r = 0, st_img = 0, st_loop = 0, st_red = 0;
    (void)st_entry; (void)st_entry_r; (void)st_img; (void)st_loop; (void)st_red;
    WL_STAMP(st_entry);
#ifdef DDK_TUNING
    st_entry_r = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: unit indices and loop control stay scalar
    const int m = lane & 15, kq = lane >> 4;
    const int NT = p.N >> 5;
    const int nt = blockIdx.x % NT, b = blockIdx.x / NT;
    const int n0 = nt << 5;
    const int cin = p.c0 + p.c1;
    const int pitch = cin + 4;
    const int nch = cin >> 5;
    float* red = lds + (local_lds_bytes(MT, cin) - 256) / 4;   // 64 floats of reduction scratch behind the tile

    // ---- weights first: they do not depend on the previous kernel's output being in this CU's reach, and their latency
    //      hides behind the image load
    const float* wl = p.w + (size_t)nt * 9 * nch * 1024 + lane * 4;   // a unit = 1024 floats: [n block][k half][lane][4]
    float4 bA[2][2], bB[2][2], bC[2][2];
    // (tap, chunk) of the next unit to LOAD, advanced by 8 units per call without a division (scalar registers)
    int ltap = 0, lchunk = wave;
    auto norm = [&](int& tap, int& chunk) {
        while (chunk >= nch) { chunk -= nch; ++tap; }
    };
    norm(ltap, lchunk);
    auto load_b = [&](float4 (&bq)[2][2]) {
        const int unit = ltap < 9 ? ltap * nch + lchunk : 9 * nch - 1;
        const float* wp = wl + (size_t)unit * 1024;
        bq[0][0] = *reinterpret_cast<const float4*>(wp);
        bq[0][1] = *reinterpret_cast<const float4*>(wp + 256);
        bq[1][0] = *reinterpret_cast<const float4*>(wp + 512);
        bq[1][1] = *reinterpret_cast<const float4*>(wp + 768);
        lchunk += 8;
        norm(ltap, lchunk);
    };
    load_b(bA);
    load_b(bB);
    load_b(bC);
    // ... and so do the tail's operands (affine, time shift, residual) of this thread's outputs: (column tid % 32, rows tid / 32 + 16 i)
    const int col = tid & 31, row = tid >> 5;
    const int c = n0 + col;
    long long o_t[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) o_t[i] = ((long long)b * MT + row + 16 * i) * p.N + c;
    constexpr int hwi = MT / IPB;                                // pixels per image
    const TailPre<MB> pre = tail_prefetch<MB>(o_t, c, IPB == 1 ? b : b * IPB + row / hwi, p);

    // ---- the image(s): [MT rows][cin] into LDS, row MT = zeros
    {
        const int q4 = cin >> 2;
        const long long row0 = (long long)b * MT;
        for (int i = tid; i < MT * q4; i += 512) {
            const int row = i / q4, c = (i - row * q4) << 2;
            const float4 v = staged_src4(p, row0 + row, c);
            *reinterpret_cast<float4*>(lds + row * pitch + c) = v;
        }
        for (int i = tid; i < q4; i += 512) *reinterpret_cast<float4*>(lds + MT * pitch + (i << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    WL_STAMP(st_img);

    // ---- k loop: this wave's units u = wave, wave + 8, ...
    f32x4 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int py[MB], px[MB], pbase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int r = i * 16 + m;
        pbase[i] = IPB == 1 ? 0 : (r / hwi) * hwi;               // first row of this pixel's image inside the block
        const int pr = r - pbase[i];
        py[i] = pr / p.W;
        px[i] = pr - py[i] * p.W;
    }
    int ctap = 0, cchunk = wave;      // (tap, chunk) of the next unit to COMPUTE
    norm(ctap, cchunk);
    auto compute = [&](const float4 (&bq)[2][2]) {
        const int t3 = ctap / 3;
        const int dy = t3 - 1, dx = ctap - t3 * 3 - 1;
        float4 a[MB][2];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int yy = py[i] + dy, xx = px[i] + dx;
            const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            const int srow = ok ? pbase[i] + yy * p.W + xx : MT;
            const float* ap = lds + srow * pitch + (cchunk << 5) + kq * 8;
            a[i][0] = *reinterpret_cast<const float4*>(ap);
            a[i][1] = *reinterpret_cast<const float4*>(ap + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b0 = reinterpret_cast<const float*>(&bq[0][0])[kk];
            const float b1 = reinterpret_cast<const float*>(&bq[1][0])[kk];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const float av = reinterpret_cast<const float*>(&a[i][0])[kk];
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[i][1], 0, 0, 0);
            }
        }
        cchunk += 8;
        norm(ctap, cchunk);
    };
    if constexpr (NU > 0) {
        int a_tap[9][MB];             // float offset of (source pixel row of tap t, this lane's k quarter, this wave's first chunk)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int yy = py[i] + t / 3 - 1, xx = px[i] + t % 3 - 1;
                const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                a_tap[t][i] = (ok ? pbase[i] + yy * p.W + xx : MT) * pitch + kq * 8 + (wave << 5);
            }
        auto compute_at = [&](int t, int sub, const float4 (&bq)[2][2]) {      // t, sub: compile-time after unrolling
            float4 a[MB][2];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const float* ap = lds + a_tap[t][i] + sub * 256;
                a[i][0] = *reinterpret_cast<const float4*>(ap);
                a[i][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float b0 = reinterpret_cast<const float*>(&bq[0][0])[kk];
                const float b1 = reinterpret_cast<const float*>(&bq[1][0])[kk];
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const float av = reinterpret_cast<const float*>(&a[i][0])[kk];
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[i][1], 0, 0, 0);
                }
            }
        };
#pragma unroll
        for (int k = 0; k < 9 * NU; k += 3) {
            compute_at(k / NU, k % NU, bA);
            load_b(bA);
            if (k + 1 < 9 * NU) compute_at((k + 1) / NU, (k + 1) % NU, bB);
            load_b(bB);
            if (k + 2 < 9 * NU) compute_at((k + 2) / NU, (k + 2) % NU, bC);
            load_b(bC);
        }
    } else {
        while (ctap < 9) {            // scalar loop control; loads past the end re-read the last unit (no load under a condition)
            compute(bA);
            load_b(bA);
            if (ctap < 9) compute(bB);
            load_b(bB);
            if (ctap < 9) compute(bC);
            load_b(bC);
        }
    }

    // ---- the 8 waves' partial tiles meet in LDS (the image is no longer needed)
    WL_STAMP(st_loop);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                lds[(wave * MT + i * 16 + kq * 4 + r) * LOC_PP + nb * 16 + m] = acc[i][nb][r];
    __syncthreads();

    // ---- thread = (column tid % 32, rows tid / 32 + 16 i): fixed-order sum of the partials + bias
    float v[MB];
    const float cb = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        float s = lds[(row + 16 * i) * LOC_PP + col];
#pragma unroll
        for (int w = 1; w < 8; ++w) s += lds[(w * MT + row + 16 * i) * LOC_PP + col];
        v[i] = s + cb;
    }

    WL_STAMP(st_red);
    gn_mish_tail<MB>(v, o_t, col, c, b, hwi, lane, wave, red, p, pre, IPB);
#ifdef DDK_TUNING
    if (tid == 0) {     // [0] entry, [1] image staged, [2] k loop done (wave 0), [3] end, [4] partials summed, [5] entry tick, [6] exit tick, [7] valid
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long st_end;
        WL_STAMP(st_end);
        unsigned long long* o = g_wl_stamps + (blockIdx.x & 511) * 8;
        o[0] = st_entry; o[1] = st_img; o[2] = st_loop; o[3] = st_end; o[4] = st_red; o[5] = st_entry_r; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = 2;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same idea for the 64-pixel maps (8x8), where the direct form would be MFMA-bound (9.4 MFLOP per workgroup): Winograd
// F(2x2, 3x3) inside the image-local tiling.  An image is 16 tiles = exactly one M block of v_mfma_f32_16x16x4_f32; per
// 32-channel chunk four TRANSFORM waves build V = B^T d B for (tile, channel) from the image in LDS (16 reads, 32 adds, 16
// writes per item) one chunk ahead, while eight MATRIX waves multiply: wave w takes positions 2w and 2w+1 (32 MFMAs) against
// U streamed from L2 in operand order (ddk_pack_conv_weight_wino_local: [n tile][chunk][position][n block][k half][lane][4]).
// V is double-buffered, one barrier per chunk.  Output transform A^T M A per (tile, channel) through LDS, then the shared
// GroupNorm tail.
struct WLocalParams {
    const float* src0;
    const float* src1;
    int c0, c1;
    const float* w;
    const float* bias;
    const float* gamma;
    const float* beta;
    const float* temb;
    int temb_stride;
    const long long* temb_rows;
    const float* addend;
    float* out;
    int H, W, N, cpg;
    float eps;
    int addend_slabs;
    long long addend_stride;
    const float* addend_bias;
    int src_slabs;               // as in LocalParams: src0 in split-K form, summed (+ bias) while the image is staged
    long long src_stride;
    const float* src_bias;
};

constexpr int WL_VP = 36;                       // V / M row pitch (floats): 16 rows cover the 64 banks once
constexpr int WL_VBUF = 16 * 16 * WL_VP;        // one V buffer: [position][tile][36]

__host__ __device__ static inline size_t wlocal_lds_bytes(int cin) { return ((size_t)65 * (cin + 4) + 2 * WL_VBUF) * 4 + 256; }

__global__ __launch_bounds__(1024) void conv3x3_gn_wlocal_kernel(const WLocalParams p) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7 matrix waves, 8..15 transform waves
    const int m = lane & 15, kq = lane >> 4;
    const int NT = p.N >> 5;
    const int nt = blockIdx.x % NT, b = blockIdx.x / NT;
    const int n0 = nt << 5;
    const int cin = p.c0 + p.c1;
    const int pitch = cin + 4;
    const int nch = cin >> 5;
    float* V = lds + 65 * pitch;
    float* red = V + 2 * WL_VBUF;
    auto lds_barrier = [] {   // LDS traffic only: __syncthreads() would also wait (vmcnt) for the weight loads deliberately in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    unsigned long long st_entry = 0, st_img = 0, st_loop = 0, st_a = 0, st_b = 0, st_wait = 0, st_tr = 0;
    (void)st_entry; (void)st_img; (void)st_loop; (void)st_a; (void)st_b; (void)st_wait; (void)st_tr;
    WL_STAMP(st_entry);

    // ---- weights of a matrix wave's two positions, one chunk ahead (two register sets, one per chunk parity).  Measured at
    //      batch 32: 1.85 us per chunk where the MFMAs need 0.95.  Round 3 (tools/share_rate.hip, profiles/r03_share_rate.txt,
    //      r03_wlocal8_pmc.json): the stream itself is NOT the limit -- 256 workgroups pulling 512 KB each, 32 per XCD on the same
    //      bytes, finish in 5.3-6.3 us from registers or LDS-DMA alike (90-98 GB/s per CU); with the weight loads ablated the kernel
    //      still takes 19.7 of 20.8 us, with the transform ablated 18.7.  What is left is the per-chunk barrier with two matrix waves
    //      per SIMD parked at it together, and ~8 us of launch + image staging + output transform + GroupNorm tail around the loop.
    //      (Rotating the n tiles over the XCDs so that an XCD's workgroups stream different bytes is slower: 23.1 us.)
    const float* wl = p.w + ((size_t)nt * nch * 16 + 2 * (wave & 7)) * 1024 + lane * 4;
    float4 bA[2][2][2], bB[2][2][2];            // [position of the pair][n block][k half]
    auto load_b = [&](int chunk, float4 (&bq)[2][2][2]) {
        chunk = chunk < nch ? chunk : nch - 1;  // past the end: harmless re-read, no load under a condition
        const float* wp = wl + (size_t)chunk * 16 * 1024;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            bq[pp][0][0] = *reinterpret_cast<const float4*>(wp + pp * 1024);
            bq[pp][0][1] = *reinterpret_cast<const float4*>(wp + pp * 1024 + 256);
            bq[pp][1][0] = *reinterpret_cast<const float4*>(wp + pp * 1024 + 512);
            bq[pp][1][1] = *reinterpret_cast<const float4*>(wp + pp * 1024 + 768);
        }
    };
    if (wave < 8) {
        load_b(0, bA);
        load_b(1, bB);
    }

    // ---- the image: [64 rows][cin] into LDS, row 64 = zeros (all 16 waves)
    {
        const int q4 = cin >> 2;
        const long long row0 = (long long)b * 64;
        for (int i = tid; i < 64 * q4; i += 1024) {
            const int row = i / q4, c = (i - row * q4) << 2;
            const float4 v = staged_src4(p, row0 + row, c);
            *reinterpret_cast<float4*>(lds + row * pitch + c) = v;
        }
        for (int i = tid; i < q4; i += 1024) *reinterpret_cast<float4*>(lds + 64 * pitch + (i << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                            // image complete
    WL_STAMP(st_img);

    const int TW = p.W >> 1;
    if (wave >= 8) {
        // ================================================================ transform waves: V = B^T d B, one chunk ahead of the
        // matrix waves.  512 threads: thread = (tile t2 / 32, position row i = (t2 / 8) % 4, channel quad t2 % 8) -- row i of B^T d
        // needs two patch rows only (0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3): 8 ds_read_b128, 8 float4 operations and
        // 4 ds_write_b128 per chunk and thread; the 8 lanes of a write group cover one tile's 128 contiguous bytes (no conflicts).
        // History (tools/wl_clock.py): one channel per item on 4 waves (32 + 32 four-byte LDS operations per thread and chunk) kept the
        // transform busy 2961 cycles per chunk where the MFMAs need 2048 -- it, not weight delivery, was this kernel's critical
        // path; float4 items on 4 waves 2720; 8 waves: see profiles/r03_wl_clock.txt.
        __builtin_amdgcn_s_setprio(3);          // a short burst the matrix waves wait for: it goes first on its SIMD
        const int t2 = tid - 512;
        const int tt = t2 >> 5, ri = (t2 >> 3) & 3, q4 = t2 & 7;
        const int tty = tt / TW, ttx = tt - tty * TW;
        const int row_a = ri == 0 ? 0 : ri == 2 ? 2 : 1, row_b = ri == 0 ? 2 : ri == 1 ? 2 : ri == 2 ? 1 : 3;
        int poff[8];                            // LDS offsets of the two patch rows (the zero row where the patch leaves the image)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * tty - 1 + (h == 0 ? row_a : row_b), x = 2 * ttx - 1 + j;
                const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                poff[h * 4 + j] = (ok ? y * p.W + x : 64) * pitch + q4 * 4;
            }
        const int voff = (4 * ri * 16 + tt) * WL_VP + q4 * 4;
        const float sgn = ri == 1 ? 1.0f : -1.0f;  // row = a + sgn * b: one exact FMA per float instead of add, subtract and select
        auto f4 = [](const float* q) { return *reinterpret_cast<const float4*>(q); };
        auto sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
        auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
        auto transform = [&](int chunk, int buf) {
            float4 r[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float4 a = f4(lds + poff[x] + (chunk << 5)), b = f4(lds + poff[4 + x] + (chunk << 5));
                r[x] = make_float4(fmaf(sgn, b.x, a.x), fmaf(sgn, b.y, a.y), fmaf(sgn, b.z, a.z), fmaf(sgn, b.w, a.w));
            }
            float* vb = V + buf * WL_VBUF + voff;
            *reinterpret_cast<float4*>(vb + 0 * (16 * WL_VP)) = sub(r[0], r[2]);
            *reinterpret_cast<float4*>(vb + 1 * (16 * WL_VP)) = add(r[1], r[2]);
            *reinterpret_cast<float4*>(vb + 2 * (16 * WL_VP)) = sub(r[2], r[1]);
            *reinterpret_cast<float4*>(vb + 3 * (16 * WL_VP)) = sub(r[1], r[3]);
        };
        transform(0, 0);
        for (int c = 0; c < nch; ++c) {
            lds_barrier();                      // V[c & 1] complete, V[(c + 1) & 1] consumed
            WL_STAMP(st_a);
            if (c + 1 < nch) transform(c + 1, (c + 1) & 1);
            WL_STAMP(st_b);
            st_tr += st_b - st_a;
        }
#ifdef DDK_TUNING
        if (tid == 512) g_wl_stamps[(blockIdx.x & 511) * 8 + 5] = st_tr;
#endif
        return;                                 // a finished wave no longer counts at the workgroup's barriers
    }

    // ==================================================================== matrix waves (512 threads)
    // the tail's operands of this thread's four outputs (tile tid / 32, channel tid % 32) are requested now: their latency hides
    // behind the k loop
    const int tt = tid >> 5, col = tid & 31, c = n0 + col;
    const int tty = tt / TW, ttx = tt - tty * TW;
    long long o_t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o_t[k] = ((long long)b * 64 + (2 * tty + (k >> 1)) * p.W + 2 * ttx + (k & 1)) * p.N + c;
    const TailPre<4> pre = tail_prefetch<4>(o_t, c, b, p);
    f32x4 acc[2][2];
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto chunk_step = [&](int chunk, const float4 (&bq)[2][2][2]) {
        WL_STAMP(st_a);
        lds_barrier();
        WL_STAMP(st_b);
        st_wait += st_b - st_a;
        const float* vb = V + (chunk & 1) * WL_VBUF + ((2 * wave) * 16 + m) * WL_VP + kq * 8;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            float4 a[2];
            a[0] = *reinterpret_cast<const float4*>(vb + pp * (16 * WL_VP));
            a[1] = *reinterpret_cast<const float4*>(vb + pp * (16 * WL_VP) + 4);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float av = reinterpret_cast<const float*>(&a[0])[kk];
#if defined(DDK_TUNING) && defined(DDK_WL_NO_MFMA)     /* ablation, tuning build only (wrong results): how long does the transform take when the matrix pipe is idle? */
                acc[pp][0][0] += av * reinterpret_cast<const float*>(&bq[pp][0][0])[kk];
                acc[pp][1][0] += av * reinterpret_cast<const float*>(&bq[pp][1][0])[kk];
#else
                acc[pp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, reinterpret_cast<const float*>(&bq[pp][0][0])[kk], acc[pp][0], 0, 0, 0);
                acc[pp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, reinterpret_cast<const float*>(&bq[pp][1][0])[kk], acc[pp][1], 0, 0, 0);
#endif
            }
        }
    };
    for (int c = 0; c < nch; c += 2) {
        chunk_step(c, bA);
        load_b(c + 2, bA);
        if (c + 1 < nch) chunk_step(c + 1, bB);
        load_b(c + 3, bB);
    }

    // ---- M[position][tile][n] of the 8 waves into LDS (over the V buffers; the transform waves have finished)
    WL_STAMP(st_loop);
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) V[((2 * wave + pp) * 16 + kq * 4 + r) * WL_VP + nb * 16 + m] = acc[pp][nb][r];
    __syncthreads();

    // ---- output transform: thread = (tile tid / 32, channel tid % 32), Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]
    float mm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) mm[k] = V[(k * 16 + tt) * WL_VP + col];
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t0[j] = (mm[0 + j] + mm[4 + j]) + mm[8 + j];
        t1[j] = (mm[4 + j] - mm[8 + j]) - mm[12 + j];
    }
    const float cb = p.bias ? p.bias[c] : 0.f;
    float v[4];
    v[0] = ((t0[0] + t0[1]) + t0[2]) + cb;
    v[1] = ((t0[1] - t0[2]) - t0[3]) + cb;
    v[2] = ((t1[0] + t1[1]) + t1[2]) + cb;
    v[3] = ((t1[1] - t1[2]) - t1[3]) + cb;
    gn_mish_tail<4>(v, o_t, col, c, b, 64, lane, wave, red, p, pre);
#ifdef DDK_TUNING
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long st_end;
        WL_STAMP(st_end);
        unsigned long long* o = g_wl_stamps + (blockIdx.x & 511) * 8;
        o[0] = st_entry; o[1] = st_img; o[2] = st_loop; o[3] = st_end; o[4] = st_wait; o[6] = (unsigned long long)nch; o[7] = 1;
    }
#endif
}

// dst[n tile][tap][chunk][n block][k half][lane = kq * 16 + n][j] = w[o = 32 nt + 16 nb + n][i = 32 chunk + 8 kq + 4 half + j][tap]
// (taps = 9: a 3x3 filter [O][I][3][3]; taps = 1: a 1x1 filter [O][I])
__global__ __launch_bounds__(256) void pack_conv_weight_local_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I,
                                                                     int i_pad, long long total, int taps) {
    const int nch = i_pad >> 5;
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63), half = (int)((idx >> 8) & 1), nb = (int)((idx >> 9) & 1);
        long long r = idx >> 10;
        const int chunk = (int)(r % nch); r /= nch;
        const int tap = (int)(r % taps);
        const int nt = (int)(r / taps);
        const int o = nt * 32 + nb * 16 + (lane & 15);
        const int i = chunk * 32 + (lane >> 4) * 8 + half * 4 + j;
        dst[idx] = i < I ? w[((long long)o * I + i) * taps + tap] : 0.f;
    }
}

// ConvTranspose2d(4, stride 2, padding 1) filter w_T [I][O][4][4] in the operand order of the level chain's CH_UPT op (level_chain.hip):
// dst[n tile][phase = 2 py + px][tap = 2 ty + tx][chunk][n block][k half][lane][j] = w_T[i][o][ky][kx] with (o, i) decoded as above and
// ky = py == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 0 : 2) (the input row offsets 0, -1 / +1, 0 of chain_upt), kx likewise from (px, tx)
__global__ __launch_bounds__(256) void pack_convT_weight_local_kernel(const float* __restrict__ w, float* __restrict__ dst, int I, int O,
                                                                      long long total) {
    const int nch = I >> 5;
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63), half = (int)((idx >> 8) & 1), nb = (int)((idx >> 9) & 1);
        long long r = idx >> 10;
        const int chunk = (int)(r % nch); r /= nch;
        const int tp = (int)(r & 3); r >>= 2;
        const int ph = (int)(r & 3);
        const int nt = (int)(r >> 2);
        const int o = nt * 32 + nb * 16 + (lane & 15);
        const int i = chunk * 32 + (lane >> 4) * 8 + half * 4 + j;
        const int py = ph >> 1, px = ph & 1, ty = tp >> 1, tx = tp & 1;
        const int ky = py == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 0 : 2), kx = px == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 0 : 2);
        dst[idx] = w[(((long long)i * O + o) * 4 + ky) * 4 + kx];
    }
}

// Winograd-domain filter U = G g G^T in the operand order of conv3x3_gn_wlocal_kernel:
// dst[n tile][chunk][position][n block][k half][lane][j], same (o, i) decoding as above.  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void pack_conv_weight_wlocal_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I,
                                                                      int i_pad, long long total) {
    const int nch = i_pad >> 5;
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63), half = (int)((idx >> 8) & 1), nb = (int)((idx >> 9) & 1);
        long long r = idx >> 10;
        const int pos = (int)(r & 15); r >>= 4;
        const int chunk = (int)(r % nch);
        const int nt = (int)(r / nch);
        const int o = nt * 32 + nb * 16 + (lane & 15);
        const int i = chunk * 32 + (lane >> 4) * 8 + half * 4 + j;
        float u = 0.f;
        if (i < I) {
            const float* g = w + ((long long)o * I + i) * 9;
            const int pi = pos >> 2, pj = pos & 3;
            float t[3];
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
                t[bb] = pi == 0 ? g[bb] : pi == 1 ? 0.5f * ((g[bb] + g[3 + bb]) + g[6 + bb]) : pi == 2 ? 0.5f * ((g[bb] - g[3 + bb]) + g[6 + bb])
                                                                                                        : g[6 + bb];
            u = pj == 0 ? t[0] : pj == 1 ? 0.5f * ((t[0] + t[1]) + t[2]) : pj == 2 ? 0.5f * ((t[0] - t[1]) + t[2]) : t[2];
        }
        dst[idx] = u;
    }
}

bool conv_gn_wlocal_ok(int H, int W, int cin, int c0, int N, int groups) {
    if (H * W != 64 || H % 2 || W % 2) return false;
    if (cin % 32 || c0 % 4 || (cin - c0) % 4 || N % 32 || N % groups) return false;
    const int cpg = N / groups;
    if (cpg != 8 && cpg != 16 && cpg != 32) return false;
    return wlocal_lds_bytes(cin) <= 160 * 1024;
}

int conv_gn_wlocal(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* gamma,
                   const float* beta, const float* temb, int temb_stride, const long long* temb_rows, const float* addend, float* out,
                   int B, int H, int W, int N, int groups, float eps, hipStream_t st, const AddendSlabs& as, const AddendSlabs& ss) {
    DDK_REQUIRE(src0 && w && gamma && beta && out, "conv_gn_wlocal: null pointer");
    DDK_REQUIRE(as.n >= 1 && (as.n == 1 || addend), "conv_gn_wlocal: addend slabs");
    DDK_REQUIRE(ss.n >= 1 && (ss.n == 1 || (ss.stride > 0 && ss.stride % 4 == 0 && aligned16(ss.bias))), "conv_gn_wlocal: source slabs");
    DDK_REQUIRE(B > 0 && groups > 0, "conv_gn_wlocal: B and groups must be positive");
    DDK_REQUIRE(c1 == 0 || src1, "conv_gn_wlocal: second source missing");
    DDK_REQUIRE(conv_gn_wlocal_ok(H, W, c0 + c1, c0, N, groups), "conv_gn_wlocal: shape not eligible (needs H*W == 64 with even H, W; "
                "cin % 32 == 0 and <= 320; N % 32 == 0; channels per group in {8, 16, 32})");
    DDK_REQUIRE(aligned16(src0) && aligned16(src1) && aligned16(w), "conv_gn_wlocal: sources and weights must be 16-byte aligned");
    DDK_TRY(ensure_device_init());
    WLocalParams p{src0, src1, c0, c1, w, bias, gamma, beta, temb, temb_stride, temb_rows, addend, out, H, W, N, N / groups, eps,
                   as.n, as.stride, as.bias, ss.n, ss.stride, ss.bias};
    hipLaunchKernelGGL(conv3x3_gn_wlocal_kernel, dim3((unsigned)((long long)B * (N / 32))), dim3(1024), wlocal_lds_bytes(c0 + c1), st, p);
    return check_launch("conv3x3_gn_wlocal_kernel");
}

bool conv_gn_local_ok(int H, int W, int cin, int c0, int N, int groups) {
    const int HW = H * W;
    if (HW != 16 && HW != 64 && !(H == 2 && W == 2)) return false;      // 2x2: four images per 16-row block (needs B % 4 == 0)
    if (cin % 32 || c0 % 4 || (cin - c0) % 4 || N % 32 || N % groups) return false;
    const int cpg = N / groups;
    if (cpg != 8 && cpg != 16 && cpg != 32) return false;
    return local_lds_bytes(HW < 16 ? 16 : HW, cin) <= 160 * 1024;
}

int conv_gn_local_init_device() {
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<16, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<16, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_wlocal_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    return DDK_OK;
}

int conv_gn_local(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* gamma,
                  const float* beta, const float* temb, int temb_stride, const long long* temb_rows, const float* addend, float* out,
                  int B, int H, int W, int N, int groups, float eps, hipStream_t st, const AddendSlabs& as, const AddendSlabs& ss) {
    DDK_REQUIRE(src0 && w && gamma && beta && out, "conv_gn_local: null pointer");
    DDK_REQUIRE(as.n >= 1 && (as.n == 1 || addend), "conv_gn_local: addend slabs");
    DDK_REQUIRE(ss.n >= 1 && (ss.n == 1 || (ss.stride > 0 && ss.stride % 4 == 0 && aligned16(ss.bias))), "conv_gn_local: source slabs");
    DDK_REQUIRE(B > 0 && groups > 0, "conv_gn_local: B and groups must be positive");
    DDK_REQUIRE(c1 == 0 || src1, "conv_gn_local: second source missing");
    DDK_REQUIRE(conv_gn_local_ok(H, W, c0 + c1, c0, N, groups), "conv_gn_local: shape not eligible (needs H*W in {16, 64}, "
                "cin % 32 == 0, N % 32 == 0, channels per group in {8, 16, 32})");
    DDK_REQUIRE(aligned16(src0) && aligned16(src1) && aligned16(w), "conv_gn_local: sources and weights must be 16-byte aligned");
    DDK_TRY(ensure_device_init());
    const int HW = H * W;
    const int ipb = HW == 4 ? 4 : 1;
    DDK_REQUIRE(B % ipb == 0, "conv_gn_local: 2x2 maps need a batch that is a multiple of 4 (four images share an M block)");
    LocalParams p{src0, src1, c0, c1, w, bias, gamma, beta, temb, temb_stride, temb_rows, addend, out, H, W, N, N / groups, eps,
                  as.n, as.stride, as.bias, ipb, ss.n, ss.stride, ss.bias};
    const size_t ldsb = local_lds_bytes(HW < 16 ? 16 : HW, c0 + c1);
    const dim3 grid((unsigned)((long long)(B / ipb) * (N / 32)));
    if (HW == 4)
        hipLaunchKernelGGL((conv3x3_gn_local_kernel<16, 4>), grid, dim3(512), ldsb, st, p);
    else if (HW == 16)
        if (c0 + c1 == 256) hipLaunchKernelGGL((conv3x3_gn_local_kernel<16, 1, 1>), grid, dim3(512), ldsb, st, p);
        else if (c0 + c1 == 512) hipLaunchKernelGGL((conv3x3_gn_local_kernel<16, 1, 2>), grid, dim3(512), ldsb, st, p);
        else hipLaunchKernelGGL(conv3x3_gn_local_kernel<16>, grid, dim3(512), ldsb, st, p);
    else
        hipLaunchKernelGGL(conv3x3_gn_local_kernel<64>, grid, dim3(512), ldsb, st, p);
    return check_launch("conv3x3_gn_local_kernel");
}

}  // namespace ddk

extern "C" {

int ddk_pack_conv_weight_local(const float* w_oihw, float* dst, int O, int I, int i_pad, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_oihw && dst && O > 0 && I > 0 && O % 32 == 0 && i_pad >= I && i_pad % 32 == 0,
                "pack_conv_weight_local: arguments (O % 32 == 0, i_pad % 32 == 0)");
    const long long total = (long long)O * 9 * i_pad;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_conv_weight_local_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_oihw,
                       dst, O, I, i_pad, total, 9);
    return check_launch("pack_conv_weight_local_kernel");
}

int ddk_pack_convT_weight_local(const float* w_iohw, float* dst, int I, int O, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_iohw && dst && I > 0 && O > 0 && O % 32 == 0 && I % 32 == 0, "pack_convT_weight_local: arguments (I % 32 == 0, O % 32 == 0)");
    const long long total = (long long)O * 16 * I;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_convT_weight_local_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_iohw,
                       dst, I, O, total);
    return check_launch("pack_convT_weight_local_kernel");
}

int ddk_pack_conv1x1_weight_local(const float* w_oi, float* dst, int O, int I, int i_pad, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_oi && dst && O > 0 && I > 0 && O % 32 == 0 && i_pad >= I && i_pad % 32 == 0,
                "pack_conv1x1_weight_local: arguments (O % 32 == 0, i_pad % 32 == 0)");
    const long long total = (long long)O * i_pad;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_conv_weight_local_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_oi,
                       dst, O, I, i_pad, total, 1);
    return check_launch("pack_conv_weight_local_kernel");
}

int ddk_pack_conv_weight_wino_local(const float* w_oihw, float* dst, int O, int I, int i_pad, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_oihw && dst && O > 0 && I > 0 && O % 32 == 0 && i_pad >= I && i_pad % 32 == 0,
                "pack_conv_weight_wino_local: arguments (O % 32 == 0, i_pad % 32 == 0)");
    const long long total = (long long)O * 16 * i_pad;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_conv_weight_wlocal_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_oihw,
                       dst, O, I, i_pad, total);
    return check_launch("pack_conv_weight_wlocal_kernel");
}

int ddk_conv3x3_gn_mish_slabs(const float* src, int src_slabs, long long src_stride, const float* src_bias, int c0, const float* weight,
                              const float* bias, const float* gamma, const float* beta, const float* temb, int temb_stride,
                              const float* addend, int addend_slabs, long long addend_stride, const float* addend_bias, float* out, int B,
                              int H, int W, int N, int groups, float eps, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(src_slabs >= 1 && addend_slabs >= 1, "conv3x3_gn_mish_slabs: slab counts start at 1");
    AddendSlabs ss, as;
    ss.n = src_slabs; ss.stride = src_stride; ss.bias = src_slabs > 1 ? src_bias : nullptr;
    as.n = addend_slabs; as.stride = addend_stride; as.bias = addend_slabs > 1 ? addend_bias : nullptr;
    if (H * W == 64)
        return conv_gn_wlocal(src, c0, nullptr, 0, weight, bias, gamma, beta, temb, temb_stride, nullptr, addend, out, B, H, W, N, groups, eps,
                              as_stream(s), as, ss);
    return conv_gn_local(src, c0, nullptr, 0, weight, bias, gamma, beta, temb, temb_stride, nullptr, addend, out, B, H, W, N, groups, eps,
                         as_stream(s), as, ss);
}

int ddk_conv3x3_gn_mish_wino_ok(int H, int W, int cin, int c0, int N, int groups) {
    return ddk::conv_gn_wlocal_ok(H, W, cin, c0, N, groups) ? 1 : 0;
}

int ddk_conv3x3_gn_mish_wino(const float* src0, int c0, const float* src1, int c1, const float* weight, const float* bias,
                             const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend, float* out,
                             int B, int H, int W, int N, int groups, float eps, ddk_stream_t s) {
    return ddk::conv_gn_wlocal(src0, c0, src1, c1, weight, bias, gamma, beta, temb, temb_stride, nullptr, addend, out, B, H, W, N, groups,
                               eps, ddk::as_stream(s));
}

int ddk_conv3x3_gn_mish_ok(int H, int W, int cin, int c0, int N, int groups) { return ddk::conv_gn_local_ok(H, W, cin, c0, N, groups) ? 1 : 0; }

int ddk_conv3x3_gn_mish(const float* src0, int c0, const float* src1, int c1, const float* weight, const float* bias,
                        const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend, float* out,
                        int B, int H, int W, int N, int groups, float eps, ddk_stream_t s) {
    return ddk::conv_gn_local(src0, c0, src1, c1, weight, bias, gamma, beta, temb, temb_stride, nullptr, addend, out, B, H, W, N, groups,
                              eps, ddk::as_stream(s));
}

}  // extern "C"

#ifdef DDK_TUNING
extern "C" int ddk_debug_read_wl_stamps(unsigned long long* host_out) {   // tuning build only (not in include/ddk.h)
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ddk::g_wl_stamps), sizeof(unsigned long long) * 8 * 512) != hipSuccess) return -2;
    static unsigned long long zeros[8 * 512];
    return hipMemcpyToSymbol(HIP_SYMBOL(ddk::g_wl_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -2;
}
#endif
