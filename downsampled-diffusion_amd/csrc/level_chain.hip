// level_chain.hip -- the whole 4x4 level of the UNet in ONE persistent launch.
//
// Reference: models/unet/unet.py:83-101 (the level's module sequence: downs[-1] ResnetBlock x 2 + attention, mid_block1, mid_attn,
// mid_block2, ups[0] ResnetBlock x 2 + attention), models/unet/blocks.py:74-84 (Block), :105-115 (ResnetBlock), :8-14, 50-71,
// 116-134 (Residual(PreNorm(LinearAttention))).
//
// Why: at batch 32 a 4x4 map is 512 pixels.  Each of the level's 19 launches (12 conv3x3 + GroupNorm + Mish, one with a 512-channel
// concat input, 3 x {to_qkv + attention core, to_out}, one 1x1 skip conv) did ~4-5 us of matrix work inside ~10 us: what stands
// between two dependent launches is a kernel boundary (2.9 us at this geometry, tools/chain_hop.hip) plus the restaging of an image
// that never leaves its eight workgroups.  Here the level is one launch: workgroup (image b, channel slice nt) walks the level's op
// list; after an op it publishes its 16 x 32 slice (sc1 stores, drained, one arrival on the image's counter) and before the next
// it waits for the image's eight arrivals and stages the 16 x C image (sc1 loads) -- the hand-off form of row 1 of
// MI355X_MICROARCH.md's sc1 table, 1.8 us per hop instead of 2.9 (profiles/r06_chain_hop.txt).  The next op's first weight units,
// affine and time shift are requested BEFORE the wait; the residual stream never goes through memory at all: thread (row, col)
// of a workgroup produces the same output element in every op, so the ResnetBlock / attention residual is a register.
//
// Geometry: grid = 8 x min(B, 32) workgroups of 512 threads, block = image slot * 8 + slice, so that the 32 workgroups of a slice
// share one XCD's L2 for that slice's filter (as in conv_local.hip); images beyond 32 are walked in rounds.  All workgroups of the
// grid are resident at once on a whole, otherwise idle MI355X (<= 256 workgroups, one per CU: 96 KB of LDS); the waits are bounded
// by wall time anyway, a give-up moves the caller's sticky fail word (ddk_unet_cluster_check) and is never silent.
#include "level_chain.h"
#include "conv_common.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ddk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LC_BIG = 18432;              // floats: image [17][cin + 4] (cin <= 512) or the 8x8 source of a stride-2 conv [65][260] | conv partials
                                           // 8 x 16 x 36 (transpose conv: 8 x 64 x 36) | attention partials 8 x 16 x 100
constexpr int LC_PP = 36;                  // conv partial row pitch
constexpr int LC_QP = 100;                 // attention partial row pitch (96 columns + 4)
constexpr int LC_RED = LC_BIG;             // 64
constexpr int LC_TILE = LC_RED + 64;       // [16][36]: the output slice on its way to 16-byte stores
constexpr int LC_KS = LC_TILE + 576;       // [16][32]
constexpr int LC_VS = LC_KS + 512;         // [16][32]
constexpr int LC_QS = LC_VS + 512;         // [16][33]
constexpr int LC_CS = LC_QS + 528;         // [32][36]
constexpr int LC_SMAX = LC_CS + 1152;      // [32]
constexpr int LC_ROWSTAT = LC_SMAX + 32;   // [16][2]
constexpr int LC_CFOLD = LC_ROWSTAT + 32;  // [192]
constexpr int LC_MISC = LC_CFOLD + 192;    // [0] this workgroup gave up waiting, [1] waves counted in at the current publish
constexpr int LC_PF = LC_MISC + 4;         // 256 floats nobody reads: where the L2 prefetch's LDS-DMA pieces land
constexpr int LC_FLOATS = LC_PF + 256;

size_t level_chain_lds_bytes() { return 100 * 1024; }    // > 80 KB: one workgroup per CU
static_assert(LC_FLOATS * 4 <= 100 * 1024, "LDS carve-up");

// ---- memory-side helpers: 16-byte sc1 loads -- issue and wait in ONE asm statement, so that no compiler-made copy can ever sit
//      between a load and the wait that makes its destination valid -- and stores
__device__ __forceinline__ void ld_sc1_x1(f32x4& a, const float* pa) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(pa) : "memory");
}
__device__ __forceinline__ void ld_sc1_x2(f32x4& a, f32x4& b, const float* pa, const float* pb) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
}
__device__ __forceinline__ void ld_sc1_x4(f32x4& a, f32x4& b, f32x4& c, f32x4& d, const float* pa, const float* pb, const float* pc, const float* pd) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(pa), "v"(pb), "v"(pc), "v"(pd) : "memory");
}
// (s_nop 1: a store of more than 8 bytes reads its upper data registers up to two cycles after issue; the compiler's hazard recogniser does
//  not see through the asm, so the wait states that keep a following VALU write out of those registers are spelled out here)
__device__ __forceinline__ void st_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

#ifdef DDK_TUNING
// Diagnostic stamps (tuning build only): per op and watched workgroup (blocks 0, 100, 255), s_memrealtime ticks (10 ns) at
// [0] op entry, [1] wait passed, [2] image staged, [3] k loop / projection done, [4] op left (after the signal).
// Three launches per step share the buffer: the 4x4 chain (19 ops) writes rows 0..2, the 8x8 chains rows 3..5 (6 ops) and 6..8 (7 ops).
__device__ unsigned long long g_lc_stamps[9 * CH_MAX_OPS * 8];
#define LC_STAMP(c, k, i)                                                                                                   \
    do {                                                                                                                    \
        if ((c).tid == 0 && (c).watch >= 0) g_lc_stamps[((c).watch * CH_MAX_OPS + (k)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define LC_STAMP(c, k, i) do { } while (0)
#endif

// Sum over the 64 lanes, the same value in every lane, on DPP + readlane (wave_sum's __shfl_xor steps compile to six dependent
// ds_bpermute_b32, ~0.25 us per reduction on an otherwise idle CU; the op's tail runs two of them back to back)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp_add = [](float x, auto ctrl) {
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});    // row_mirror: every lane holds its 16-lane row's sum
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int... I, class F>
__device__ __forceinline__ void static_for_lc_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for_lc(F&& f) { static_for_lc_impl(std::make_integer_sequence<int, N>{}, f); }

struct ChainCtx {
    float* lds;
    int tid, lane, wave, b, nt;
    unsigned signals;            // signalling ops of this image so far (uniform)
    float keep, keep2;           // the residual stream's / the skip conv's element of this thread (row tid / 32, column tid % 32 of the slice)
    float* misc;                 // LDS: [0] this workgroup gave up waiting, [1] waves counted in at the current publish
    unsigned pf_dst;             // LDS byte address of the scratch kilobyte the prefetch pieces land in
    int nwaves, units3;          // waves in the workgroup; filter units per 32-channel chunk of a CH_CONV3 op (9 taps, or 16 Winograd positions)
    int slot, slots;             // image slot of this workgroup and slots in the grid (the 32 workgroups of a slice share an XCD)
    int watch, k;                // tuning build: which stamp row this workgroup writes (-1: none), current op index
};

// every arrival of the image so far (8 per signalling op); bounded: 20 ms of wall time, then this workgroup stops waiting for good
__device__ __forceinline__ void chain_wait(const ChainParams& p, ChainCtx& c) {
    int* dead = reinterpret_cast<int*>(c.misc);
    if (c.tid == 0 && !*dead) {
        const unsigned need = 8u * c.signals;
        unsigned* cnt = p.cnt + c.b * 32;
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            const long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000LL) {       // 100 MHz ticks
                    *dead = 1;
                    atomicAdd(p.fail, 1u);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// The NEXT op's filter slice into this XCD's L2 while this op runs (the level's filters are 33 MB, read once per step: without
// this every op's k loop starts on far-memory latency -- the 512-channel conv ran at 13-16 us where its MFMAs need 8.4,
// profiles/r06_chain_clock_v1.txt).  The 32 workgroups of a slice share the XCD: workgroup `slot` touches every slots-th part, one
// 16-byte piece per 128-byte line and thread.  The pieces are LDS-DMA loads into a scratch kilobyte nobody reads: a load with a
// register destination would have to keep that register reserved until it lands, and the compiler cannot be told (a first version
// did that through an asm output operand; a register copy the compiler inserted behind the load freed the real destination, the
// late data landed in an address register and the kernel faulted).
__device__ __forceinline__ void chain_prefetch(const ChainOp& o, const ChainCtx& c) {
    const int cin = o.c0 + o.c1;
    const float* base;
    int lines;                                           // 128-byte lines of this workgroup's slice
    if (o.kind == CH_ATTN) {
        if (c.nt >= 4) return;
        base = o.w + (size_t)c.nt * (cin >> 5) * 3072;
        lines = (cin >> 5) * 3072 / 32;
    } else {
        const int taps = o.kind == CH_CONV3 ? c.units3 : o.kind == CH_UPT ? 16 : 1;
        base = o.w + (size_t)c.nt * taps * (cin >> 5) * 1024;
        lines = taps * (cin >> 5) * 1024 / 32;
    }
    const int share = (lines + c.slots - 1) / c.slots;
    const int first = c.slot * share;
    const int n = lines - first < share ? lines - first : share;      // this workgroup's lines: [first, first + n)
    for (int i = c.wave * 64; i < n; i += c.nwaves * 64) {            // wave-uniform trip count; lanes past the end re-touch the last line
        int line = first + i + c.lane;
        line = line < first + n ? line : first + n - 1;
        lds_dma16(base + (size_t)line * 32, c.pf_dst);
    }
}

// The end of an op.  Thread (row = tid / 32, col = tid % 32) holds output element y of the workgroup's 16 x 32 slice -- the same
// element in every op, so the residuals are registers.  The four lanes of a quad hold four consecutive channels of one pixel: they
// are gathered with three DPP moves (no LDS, no barrier) and every fourth lane stores 16 bytes.  Publishing: each wave drains its
// OWN stores and counts itself in on an LDS word; the wave whose count completes the eight arrives on the image's counter for all
// of them (MI355X_MICROARCH.md, hand-off table, row 1 with the LDS-counter form of its condition (3)) -- no workgroup barrier, the
// waves run ahead into the next op's prologue as they finish.
__device__ __forceinline__ void chain_finish(const ChainParams& p, const ChainOp& o, ChainCtx& c, float y, int n0) {
    if (o.flags & CHF_ADD_KEEP) y += c.keep;
    if (o.flags & CHF_ADD_KEEP2) y += c.keep2;
    if (o.flags & CHF_SAVE_KEEP) c.keep = y;
    if (o.flags & CHF_SAVE_KEEP2) c.keep2 = y;
    if (!(o.flags & CHF_NO_OUT)) {
        const int yb = __builtin_bit_cast(int, y);
        const float y0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
        const float y1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0x55, 0xF, 0xF, true));   // [1,1,1,1]
        const float y2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0xAA, 0xF, 0xF, true));   // [2,2,2,2]
        const float y3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0xFF, 0xF, 0xF, true));   // [3,3,3,3]
        if ((c.lane & 3) == 0) {
            float* dst = o.out + ((long long)c.b * 16 + (c.tid >> 5)) * o.n_out + n0 + (c.tid & 31);
            const f32x4 v{y0, y1, y2, y3};
            if (o.flags & CHF_SIGNAL) st_sc1(dst, v);
            else *reinterpret_cast<f32x4*>(dst) = v;
        }
    }
    if (o.flags & CHF_SIGNAL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c.lane == 0) {
            unsigned* lcnt = reinterpret_cast<unsigned*>(c.misc) + 1;
            if ((atomicAdd(lcnt, 1u) & 7u) == 7u) __hip_atomic_fetch_add(p.cnt + c.b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// [16 rows][cin] of image b (src0 | src1 along the channels) -> LDS rows of pitch cin + 4; 512 threads, cin / 128 float4 each
__device__ __forceinline__ void chain_stage(const ChainOp& o, ChainCtx& c, int cin, int pitch) {
    const int q4 = cin >> 2;
    auto addr = [&](int idx) -> const float* {
        const int row = idx / q4, ch = (idx - row * q4) << 2;
        const long long r = (long long)c.b * 16 + row;
        return ch < o.c0 ? o.src0 + r * o.c0 + ch : o.src1 + r * o.c1 + (ch - o.c0);
    };
    auto put = [&](int idx, f32x4 v) {
        const int row = idx / q4, ch = (idx - row * q4) << 2;
        *reinterpret_cast<f32x4*>(c.lds + row * pitch + ch) = v;
    };
    if (cin == 512) {
        f32x4 v0, v1, v2, v3;
        ld_sc1_x4(v0, v1, v2, v3, addr(c.tid), addr(c.tid + 512), addr(c.tid + 1024), addr(c.tid + 1536));
        put(c.tid, v0); put(c.tid + 512, v1); put(c.tid + 1024, v2); put(c.tid + 1536, v3);
    } else if (cin == 256) {
        f32x4 v0, v1;
        ld_sc1_x2(v0, v1, addr(c.tid), addr(c.tid + 512));
        put(c.tid, v0); put(c.tid + 512, v1);
    } else {    // 128
        f32x4 v0;
        ld_sc1_x1(v0, addr(c.tid));
        put(c.tid, v0);
    }
    if (c.tid < q4) *reinterpret_cast<f32x4*>(c.lds + 16 * pitch + (c.tid << 2)) = f32x4{0.f, 0.f, 0.f, 0.f};   // the row out-of-image taps read
}

// the 8x8 source map of a stride-2 conv: [64 rows][256] of image b -> LDS rows of pitch 260, row 64 = zeros; 8 x 16 bytes per thread
__device__ __forceinline__ void chain_stage64(const ChainOp& o, ChainCtx& c) {
    const float* src = o.src0 + (long long)c.b * 64 * 256;
#pragma unroll
    for (int r0 = 0; r0 < 4096; r0 += 2048) {
        f32x4 v0, v1, v2, v3;
        ld_sc1_x4(v0, v1, v2, v3, src + (size_t)(r0 + c.tid) * 4, src + (size_t)(r0 + c.tid + 512) * 4, src + (size_t)(r0 + c.tid + 1024) * 4,
                  src + (size_t)(r0 + c.tid + 1536) * 4);
        auto put = [&](int idx, f32x4 v) { *reinterpret_cast<f32x4*>(c.lds + (idx >> 6) * 260 + ((idx & 63) << 2)) = v; };
        put(r0 + c.tid, v0); put(r0 + c.tid + 512, v1); put(r0 + c.tid + 1024, v2); put(r0 + c.tid + 1536, v3);
    }
    if (c.tid < 64) *reinterpret_cast<f32x4*>(c.lds + 64 * 260 + (c.tid << 2)) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// CH_CONV3 / CH_CONV1.  The arithmetic of conv3x3_gn_local_kernel<16, 1, NU> (conv_local.hip): direct conv on v_mfma_f32_16x16x4_f32,
// the 8 waves split k = taps x 32-channel chunks, weights straight from L2 into registers in MFMA operand order (three units in
// flight), partial tiles meet in LDS, GroupNorm over the slice (32 channels x 16 pixels = one group) in two passes.
// NU > 0: cin = 256 NU and 9 taps (compile-time taps, the nine source rows precomputed); NU = 0: generic (the 1x1 convs).
template <int NU>
__device__ __forceinline__ void chain_conv(const ChainParams& p, const ChainOp& o, ChainCtx& c) {
    float* lds = c.lds;
    const int tid = c.tid, lane = c.lane, wave = c.wave;
    const int m = lane & 15, kq = lane >> 4;
    const int n0 = c.nt << 5;
    const int cin = o.c0 + o.c1, pitch = cin + 4, nch = cin >> 5;
    const int taps = o.kind == CH_CONV3 ? 9 : 1;

    // ---- what does not depend on the other workgroups: weights, affine, shift, bias -- requested before the wait
    const float* wl = o.w + (size_t)c.nt * taps * nch * 1024 + lane * 4;
    float4 bA[2][2], bB[2][2], bC[2][2];
    int ltap = 0, lchunk = wave;
    auto norm = [&](int& tap, int& chunk) {
        while (chunk >= nch) { chunk -= nch; ++tap; }
    };
    norm(ltap, lchunk);
    auto load_b = [&](float4 (&bq)[2][2]) {
        const int unit = ltap < taps ? ltap * nch + lchunk : taps * nch - 1;
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
    const int col = tid & 31, row = tid >> 5;
    const int ch = n0 + col;
    float ga = 1.f, be = 0.f, sh = 0.f;
    if (o.kind == CH_CONV3 && !(o.flags & CHF_NO_GN)) { ga = o.gamma[ch]; be = o.beta[ch]; }
    if (o.temb_off >= 0) {
        const long long tr = p.temb_rows ? p.temb_rows[c.b] : c.b;
        sh = p.temb[tr * p.temb_stride + o.temb_off + ch];
    }
    const float cb = o.bias ? o.bias[ch] : 0.f;
    if (o.flags & CHF_KEEP_FROM_SRC) c.keep = o.src0[((long long)c.b * 16 + row) * o.c0 + ch];
    if (c.k + 1 < p.n_ops) chain_prefetch(p.op[c.k + 1], c);
    const bool down = NU == 1 && (o.flags & CHF_DOWN);      // stride 2: the source is the 8x8 map

    LC_STAMP(c, c.k, 0);
    if (o.flags & CHF_WAIT) chain_wait(p, c);
    else __syncthreads();                 // a wave may still be reading the previous op's partial tiles: the image lands on them
    LC_STAMP(c, c.k, 1);
    if (down) chain_stage64(o, c);
    else chain_stage(o, c, cin, pitch);   // (its vmcnt(0) also drains the prefetch pieces)
    __syncthreads();
    LC_STAMP(c, c.k, 2);

    // ---- k loop
    f32x4 acc[2];
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int py = m >> 2, px = m & 3;
    if constexpr (NU > 0) {
        int a_tap[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (down) {                   // output pixel (py, px) of the 4x4 map reads input pixel (2 py + dy, 2 px + dx) of the 8x8 one
                const int yy = 2 * py + t / 3 - 1, xx = 2 * px + t % 3 - 1;
                const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
                a_tap[t] = (ok ? yy * 8 + xx : 64) * pitch + kq * 8 + (wave << 5);
            } else {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                const bool ok = (unsigned)yy < 4u && (unsigned)xx < 4u;
                a_tap[t] = (ok ? yy * 4 + xx : 16) * pitch + kq * 8 + (wave << 5);
            }
        }
        auto compute_at = [&](int t, int sub, const float4 (&bq)[2][2]) {
            const float* ap = lds + a_tap[t] + sub * 256;
            const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[0][0])[kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[1][0])[kk], acc[1], 0, 0, 0);
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
        int ctap = 0, cchunk = wave;
        norm(ctap, cchunk);
        auto compute = [&](const float4 (&bq)[2][2]) {
            int srow = m;
            if (taps == 9) {
                const int t3 = ctap / 3;
                const int yy = py + t3 - 1, xx = px + ctap - t3 * 3 - 1;
                srow = ((unsigned)yy < 4u && (unsigned)xx < 4u) ? yy * 4 + xx : 16;
            }
            const float* ap = lds + srow * pitch + (cchunk << 5) + kq * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[0][0])[kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[1][0])[kk], acc[1], 0, 0, 0);
            }
            cchunk += 8;
            norm(ctap, cchunk);
        };
        while (ctap < taps) {
            compute(bA);
            load_b(bA);
            if (ctap < taps) compute(bB);
            load_b(bB);
            if (ctap < taps) compute(bC);
            load_b(bC);
        }
    }

    // ---- the 8 waves' partial tiles meet in LDS (the image is no longer needed)
    LC_STAMP(c, c.k, 3);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(wave * 16 + kq * 4 + r) * LC_PP + nb * 16 + m] = acc[nb][r];
    __syncthreads();
    float v = lds[row * LC_PP + col];
#pragma unroll
    for (int w = 1; w < 8; ++w) v += lds[(w * 16 + row) * LC_PP + col];
    v += cb;
    float y = v;
    if (o.kind == CH_CONV3 && !(o.flags & CHF_NO_GN)) {
        // GroupNorm of the slice = one group of 32 channels x 16 pixels.  Each wave: mean and M2 of its own 64 values (two passes inside
        // the wave); the eight {mean, M2} records merge exactly (Chan et al.) behind ONE barrier.
        float* red = lds + LC_RED;
        const float mw = wave_sum_dpp(v) * (1.0f / 64.0f);
        const float dw = v - mw;
        const float m2w = wave_sum_dpp(dw * dw);
        if (lane == 0) { red[2 * wave] = mw; red[2 * wave + 1] = m2w; }
        __syncthreads();
        float msum = red[0], m2 = red[1];
#pragma unroll
        for (int w = 1; w < 8; ++w) { msum += red[2 * w]; m2 += red[2 * w + 1]; }
        const float mean = msum * 0.125f;
        float dev = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const float dm = red[2 * w] - mean; dev += dm * dm; }
        const float var = (m2 + 64.0f * dev) * (1.0f / 512.0f);
        const float rstd = 1.0f / sqrtf(var + p.gn_eps);
        y = mish_f((v - mean) * rstd * ga + be) + sh;
    }
    chain_finish(p, o, c, y, n0);
    if (o.flags & CHF_SIGNAL) ++c.signals;
    LC_STAMP(c, c.k, 4);
}

// CH_UPT: ConvTranspose2d(C, C, 4, stride 2, padding 1) + bias on the 4x4 map -> 8x8 (blocks.py:32-38).  out[2y + py][2x + px] only gets
// input rows y + dy with ky = 2y + py + 1 - 2 (y + dy) in [0, 4): py = 0 -> (dy 0, ky 1), (dy -1, ky 3); py = 1 -> (dy +1, ky 0), (dy 0, ky 2)
// -- and the same along x: four output phases, each a 2 x 2-tap stride-1 conv of the 4x4 map = a [16 x 4 C] x [4 C x 32] product per
// workgroup.  Wave w takes chunk w of every (phase, tap): 16 units of 16 MFMAs; filter in operand order [slice][phase][tap][chunk][1024]
// (ddk_pack_convT_weight_local).  Thread (pixel, channel) ends up with the pixel's four phase outputs = a 2 x 2 block of the 8x8 map.
__device__ __forceinline__ void chain_upt(const ChainParams& p, const ChainOp& o, ChainCtx& c) {
    float* lds = c.lds;
    const int tid = c.tid, lane = c.lane, wave = c.wave;
    const int m = lane & 15, kq = lane >> 4;
    const int n0 = c.nt << 5;
    const int cin = o.c0, pitch = cin + 4, nch = cin >> 5;          // cin = 256: 8 chunks = one per wave
    const float* wl = o.w + ((size_t)c.nt * 16 * nch + wave) * 1024 + lane * 4;      // unit u = (phase * 4 + tap) * nch + chunk
    float4 bA[2][2], bB[2][2], bC[2][2];
    int lu = 0;
    auto load_b = [&](float4 (&bq)[2][2]) {
        const float* wp = wl + (size_t)(lu < 16 ? lu : 15) * nch * 1024;
        bq[0][0] = *reinterpret_cast<const float4*>(wp);
        bq[0][1] = *reinterpret_cast<const float4*>(wp + 256);
        bq[1][0] = *reinterpret_cast<const float4*>(wp + 512);
        bq[1][1] = *reinterpret_cast<const float4*>(wp + 768);
        ++lu;
    };
    load_b(bA);
    load_b(bB);
    load_b(bC);
    const int col = tid & 31, row = tid >> 5;
    const float cb = o.bias ? o.bias[n0 + col] : 0.f;
    if (c.k + 1 < p.n_ops) chain_prefetch(p.op[c.k + 1], c);
    LC_STAMP(c, c.k, 0);
    if (o.flags & CHF_WAIT) chain_wait(p, c);
    else __syncthreads();
    LC_STAMP(c, c.k, 1);
    chain_stage(o, c, cin, pitch);
    __syncthreads();
    LC_STAMP(c, c.k, 2);
    f32x4 acc[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[ph][0] = acc[ph][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int y = m >> 2, x = m & 3;
    auto unit = [&](auto phc, auto tpc, const float4 (&bq)[2][2]) {
        constexpr int ph = decltype(phc)::value, tp = decltype(tpc)::value;
        constexpr int pyy = ph >> 1, pxx = ph & 1, ty = tp >> 1, tx = tp & 1;
        constexpr int dy = pyy == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 1 : 0);
        constexpr int dx = pxx == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 1 : 0);
        const int yy = y + dy, xx = x + dx;
        const bool ok = (unsigned)yy < 4u && (unsigned)xx < 4u;
        const float* ap = lds + (ok ? yy * 4 + xx : 16) * pitch + (wave << 5) + kq * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            acc[ph][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[0][0])[kk], acc[ph][0], 0, 0, 0);
            acc[ph][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[1][0])[kk], acc[ph][1], 0, 0, 0);
        }
    };
    static_for_lc<16>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        using PH = std::integral_constant<int, u / 4>;
        using TP = std::integral_constant<int, u % 4>;
        if constexpr (u % 3 == 0) { unit(PH{}, TP{}, bA); load_b(bA); }
        else if constexpr (u % 3 == 1) { unit(PH{}, TP{}, bB); load_b(bB); }
        else { unit(PH{}, TP{}, bC); load_b(bC); }
    });
    LC_STAMP(c, c.k, 3);
    __syncthreads();
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[((wave * 4 + ph) * 16 + kq * 4 + r) * LC_PP + nb * 16 + m] = acc[ph][nb][r];
    __syncthreads();
    float yv[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        float s = lds[(ph * 16 + row) * LC_PP + col];
#pragma unroll
        for (int w = 1; w < 8; ++w) s += lds[((w * 4 + ph) * 16 + row) * LC_PP + col];
        yv[ph] = s + cb;
    }
    {   // the four lanes of a quad hold four consecutive channels: lane j stores phase j's pixel as 16 bytes
        float g[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yb = __builtin_bit_cast(int, yv[k]);
            g[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0x00, 0xF, 0xF, true));
            g[k][1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0x55, 0xF, 0xF, true));
            g[k][2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0xAA, 0xF, 0xF, true));
            g[k][3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0xFF, 0xF, 0xF, true));
        }
        const int j = lane & 3;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = j == 0 ? g[0][q] : j == 1 ? g[1][q] : j == 2 ? g[2][q] : g[3][q];
        const int oy = 2 * (row >> 2) + (j >> 1), ox = 2 * (row & 3) + (j & 1);
        float* dst = o.out + ((long long)c.b * 64 + oy * 8 + ox) * o.n_out + n0 + (col & ~3);
        if (o.flags & CHF_SIGNAL) st_sc1(dst, v);
        else *reinterpret_cast<f32x4*>(dst) = v;
    }
    if (o.flags & CHF_SIGNAL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            unsigned* lcnt = reinterpret_cast<unsigned*>(c.misc) + 1;
            if ((atomicAdd(lcnt, 1u) & 7u) == 7u) __hip_atomic_fetch_add(p.cnt + c.b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++c.signals;
    }
    LC_STAMP(c, c.k, 4);
}

// CH_ATTN (blocks.py:116-134 behind PreNorm, :57-71): head h = nt < 4 of image b.  q | k | v of the head = LayerNorm-folded to_qkv
// ([16 x C] x [C x 96], the 8 waves split the 32-channel chunks, weights in operand order: qkv_operand_pack_kernel), k soft-maxed
// over the 16 pixels, ctx = k^T v, out = q ctx -- linattn_small_qkv_kernel's arithmetic on 512 threads.  o.gamma / o.beta: W g, W b.
__device__ __forceinline__ void chain_attn(const ChainParams& p, const ChainOp& o, ChainCtx& c) {
    float* lds = c.lds;
    const int tid = c.tid, lane = c.lane, wave = c.wave;
    // the to_out behind this op is short: warm the L2 for the conv behind it as well
    if (c.k + 1 < p.n_ops) chain_prefetch(p.op[c.k + 1], c);
    if (c.k + 2 < p.n_ops) chain_prefetch(p.op[c.k + 2], c);
    if (c.nt < 4) {
        const int h = c.nt;
        const int m = lane & 15, kq = lane >> 4;
        const int C = o.c0, pitch = C + 4, nch = C >> 5;
        float* xs = lds;
        float* ks = lds + LC_KS;
        float* vs = lds + LC_VS;
        float* qs = lds + LC_QS;
        float* cs = lds + LC_CS;
        float* smax = lds + LC_SMAX;
        float* rowstat = lds + LC_ROWSTAT;
        float* cfold = lds + LC_CFOLD;
        const float* wl = o.w + (size_t)h * nch * 3072 + lane * 4;         // a chunk = 6 n blocks x 512 floats
        float4 bq[6][2];
        {
            const int chunk = wave < nch ? wave : nch - 1;
            const float* wp = wl + (size_t)chunk * 3072;
#pragma unroll
            for (int nb = 0; nb < 6; ++nb) {
                bq[nb][0] = *reinterpret_cast<const float4*>(wp + nb * 512);
                bq[nb][1] = *reinterpret_cast<const float4*>(wp + nb * 512 + 256);
            }
        }
        if (tid < 192) {                                   // the fold vectors of this head's 96 columns (q | k | v)
            const int n = tid % 96;
            const int colw = (n >> 5) * 128 + h * 32 + (n & 31);
            cfold[tid] = tid < 96 ? o.gamma[colw] : o.beta[colw];
        }
        LC_STAMP(c, c.k, 0);
        if (o.flags & CHF_WAIT) chain_wait(p, c);
        else __syncthreads();
        LC_STAMP(c, c.k, 1);
        chain_stage(o, c, C, pitch);
        __syncthreads();
        LC_STAMP(c, c.k, 2);
        {   // LayerNorm statistics of the 16 pixel rows: 32 threads per row, two passes over the resident row (biased variance, eps on the std)
            const int row = tid >> 5, sub = tid & 31;
            float s1 = 0.f;
            for (int ch = sub * 4; ch < C; ch += 128) {
                const float4 v = *reinterpret_cast<const float4*>(xs + row * pitch + ch);
                s1 += (v.x + v.y) + (v.z + v.w);
            }
#pragma unroll
            for (int of = 1; of < 32; of <<= 1) s1 += __shfl_xor(s1, of, 64);
            const float inv_c = 1.0f / (float)C;
            const float mean = s1 * inv_c;
            float s2 = 0.f;
            for (int ch = sub * 4; ch < C; ch += 128) {
                const float4 v = *reinterpret_cast<const float4*>(xs + row * pitch + ch);
                const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
#pragma unroll
            for (int of = 1; of < 32; of <<= 1) s2 += __shfl_xor(s2, of, 64);
            if (sub == 0) {
                const float r = 1.0f / (sqrtf(s2 * inv_c) + p.ln_eps);
                rowstat[2 * row] = r;
                rowstat[2 * row + 1] = r * mean;
            }
        }
        // ---- [16 x C] x [C x 96]: wave w takes chunks w, w + 8, ...
        f32x4 acc[6];
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = wave; chunk < nch; chunk += 8) {
            const float* ap = xs + m * pitch + (chunk << 5) + kq * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int nb = 0; nb < 6; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[nb][0])[kk], acc[nb], 0, 0, 0);
            if (chunk + 8 < nch) {
                const float* wp = wl + (size_t)(chunk + 8) * 3072;
#pragma unroll
                for (int nb = 0; nb < 6; ++nb) {
                    bq[nb][0] = *reinterpret_cast<const float4*>(wp + nb * 512);
                    bq[nb][1] = *reinterpret_cast<const float4*>(wp + nb * 512 + 256);
                }
            }
        }
        LC_STAMP(c, c.k, 3);
        __syncthreads();                                    // image consumed, rowstat written
#pragma unroll
        for (int nb = 0; nb < 6; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[(wave * 16 + kq * 4 + r) * LC_QP + nb * 16 + m] = acc[nb][r];
        __syncthreads();
        // ---- the 8 partials in fixed order + the folded LayerNorm -> q | k | v
        for (int e = tid; e < 16 * 96; e += 512) {
            const int row = e / 96, n = e - row * 96;
            float s = xs[row * LC_QP + n];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += xs[(w * 16 + row) * LC_QP + n];
            const float v = rowstat[2 * row] * s - rowstat[2 * row + 1] * cfold[n] + cfold[96 + n];
            const int sel = n >> 5, d = n & 31;
            if (sel == 0) qs[row * 33 + d] = v;
            else if (sel == 1) ks[row * 32 + d] = v;
            else vs[row * 32 + d] = v;
        }
        __syncthreads();
        // ---- softmax over the pixels (dim=-1 of the reference's [b, heads, c, n]), context, apply
        if (tid < 32) {
            float mx = ks[tid];
#pragma unroll
            for (int n = 1; n < 16; ++n) mx = fmaxf(mx, ks[n * 32 + tid]);
            smax[tid] = mx;
        }
        __syncthreads();
        ks[tid] = __expf(ks[tid] - smax[tid & 31]);
        __syncthreads();
        {
            const int d = tid >> 4, e0 = (tid & 15) << 1;
            float a0 = 0.f, a1 = 0.f, den = 0.f;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const float kd = ks[n * 32 + d];
                a0 += kd * vs[n * 32 + e0];
                a1 += kd * vs[n * 32 + e0 + 1];
                den += kd;
            }
            const float inv = 1.0f / den;
            cs[d * 36 + e0] = a0 * inv;
            cs[d * 36 + e0 + 1] = a1 * inv;
        }
        __syncthreads();
        const int n = tid >> 5, e = tid & 31;      // out[n][e] = sum_d q[n][d] ctx[d][e]: the thread's own element, as in chain_conv
        float y = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) y += qs[n * 33 + d] * cs[d * 36 + e];
        chain_finish(p, o, c, y, h * 32);
    } else {
        // heads are workgroups 0..3; 4..7 have nothing to compute here and only warm their L2 for the ops behind.  They WAIT like the
        // others before they arrive: the image's counter is one monotonic count of arrivals, and "8 x signalling ops so far" only means
        // "everybody has published the previous op" if nobody can arrive for op k + 1 before every arrival of op k is in (a first version
        // let them arrive at once: a reader could then pass its wait with one slice of the previous op still unpublished -- rare on the
        // 4x4 maps, every first forward on the 8x8 ones)
        if (o.flags & CHF_WAIT) chain_wait(p, c);
        if ((o.flags & CHF_SIGNAL) && tid == 0) __hip_atomic_fetch_add(p.cnt + c.b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (o.flags & CHF_SIGNAL) ++c.signals;
    LC_STAMP(c, c.k, 4);
}

__global__ __launch_bounds__(512) void level_chain_kernel(const ChainParams p) {
    extern __shared__ __align__(16) float lds[];
    ChainCtx c;
    c.lds = lds;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
    c.nt = blockIdx.x & 7;
    c.misc = lds + LC_MISC;
    c.pf_dst = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + LC_PF);
    c.nwaves = 8;
    c.units3 = 9;
    c.watch = blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : blockIdx.x == 255 ? 2 : -1;
    c.k = 0;
    if (c.tid == 0) { *reinterpret_cast<int*>(lds + LC_MISC) = 0; reinterpret_cast<unsigned*>(lds + LC_MISC)[1] = 0u; }
    __syncthreads();
    const int slots = gridDim.x >> 3;
    c.slot = blockIdx.x >> 3;
    c.slots = slots;
    for (int b = blockIdx.x >> 3; b < p.B; b += slots) {
        c.b = b;
        c.signals = 0;
        c.keep = c.keep2 = 0.f;
        // the first op's filter is cold too: its lines are requested now, the later part of its k loop hits L2 (retired at its stage)
        chain_prefetch(p.op[0], c);
        for (int k = 0; k < p.n_ops; ++k) {
            const ChainOp& o = p.op[k];
            c.k = k;
            if (o.kind == CH_ATTN) chain_attn(p, o, c);
            else if (o.kind == CH_UPT) chain_upt(p, o, c);
            else if (o.kind == CH_CONV1) chain_conv<0>(p, o, c);
            else if (o.c0 + o.c1 == 256) chain_conv<1>(p, o, c);
            else if (o.c0 + o.c1 == 512) chain_conv<2>(p, o, c);
            else chain_conv<0>(p, o, c);
        }
        // the image's last departure re-arms its counters for the next launch: every one of the eight has passed its last wait
        __syncthreads();              // (every wave has left the last op)
        if (c.tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.done + b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == 7u) {
                __hip_atomic_store(p.cnt + b * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.done + b * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ===========================================================================================================================
// The 8x8 levels (downs[-2] and ups[1] of the cfg4 UNet: 64 pixels x 256 channels per image) the same way, 1024 threads:
// CH_CONV3 is conv3x3_gn_wlocal_kernel's arithmetic (conv_local.hip) -- Winograd F(2x2, 3x3), an image = 16 tiles = one M block of
// v_mfma_f32_16x16x4_f32; eight TRANSFORM waves (8..15) build V = B^T d B of the next 32-channel chunk while eight MATRIX waves (0..7)
// multiply positions 2w, 2w+1 of this one against U streamed from L2 in operand order (ddk_pack_conv_weight_wino_local), V
// double-buffered, one LDS barrier per chunk -- with a 512-channel concat input walked as two staged halves.  CH_CONV1 / CH_ATTN:
// direct products on the matrix waves.  Thread (tile tid / 32, channel tid % 32) of the matrix waves owns the 2 x 2 pixels of its
// tile in every op: the residuals are four registers.
constexpr int L8_P = 260;                      // image row pitch (256 channels per staged pass + 4)
constexpr int L8_IMG = 65 * L8_P;              // rows 0..63 + the zero row out-of-image taps read; later: attention / 1x1 partial tiles
constexpr int L8_VP = 36;
constexpr int L8_VBUF = 16 * 16 * L8_VP;       // one V buffer [position][tile][36]
constexpr int L8_V = L8_IMG;                   // two V buffers; later M [position][tile][36]; attention: k, v, q, ctx
constexpr int L8_KS = L8_V;                    // [64][32]
constexpr int L8_VS = L8_KS + 2048;            // [64][32]
constexpr int L8_QS = L8_VS + 2048;            // [64][33]
constexpr int L8_CS = L8_QS + 2112;            // [32][36]
constexpr int L8_RED = L8_V + 2 * L8_VBUF;     // 64
constexpr int L8_SMAX = L8_RED + 64;           // [8][32] partial maxima, [32] maxima
constexpr int L8_ROWSTAT = L8_SMAX + 320;      // [64][2]
constexpr int L8_CFOLD = L8_ROWSTAT + 128;     // [192]
constexpr int L8_MISC = L8_CFOLD + 192;
constexpr int L8_PF = L8_MISC + 4;
constexpr int L8_FLOATS = L8_PF + 256;
static_assert(L8_CS + 1152 <= L8_RED, "attention arrays fit the V buffers");
static_assert(L8_FLOATS * 4 <= 160 * 1024, "LDS carve-up (8x8)");
static_assert(2 * 64 * LC_QP <= L8_IMG && 8 * 64 * LC_PP <= 2 * L8_VBUF, "partial tiles fit");

// [64 rows][ncols] of image b, channels [ch_lo, ch_lo + ncols) of (src0 | src1) -> LDS rows of pitch ncols + 4 (+ the zero row)
__device__ __forceinline__ void c8_stage(const ChainOp& o, ChainCtx& c, int ch_lo, int ncols) {
    const int q4 = ncols >> 2, pitch = ncols + 4;
    auto addr = [&](int idx) -> const float* {
        const int row = idx / q4, ch = ch_lo + ((idx - row * q4) << 2);
        const long long r = (long long)c.b * 64 + row;
        return ch < o.c0 ? o.src0 + r * o.c0 + ch : o.src1 + r * o.c1 + (ch - o.c0);
    };
    auto put = [&](int idx, f32x4 v) {
        const int row = idx / q4, ch = (idx - row * q4) << 2;
        *reinterpret_cast<f32x4*>(c.lds + row * pitch + ch) = v;
    };
    // two 16-byte pieces per thread and round (a 1024-thread workgroup leaves 128 registers per thread: four pieces and their
    // addresses in flight next to the matrix waves' 64 filter registers spilled)
    for (int r0 = 0; r0 < 64 * q4; r0 += 2048) {
        f32x4 v0, v1;
        ld_sc1_x2(v0, v1, addr(r0 + c.tid), addr(r0 + c.tid + 1024));
        put(r0 + c.tid, v0); put(r0 + c.tid + 1024, v1);
    }
    if (c.tid < q4) *reinterpret_cast<f32x4*>(c.lds + 64 * pitch + (c.tid << 2)) = f32x4{0.f, 0.f, 0.f, 0.f};
}

struct C8Keep { float keep[4], keep2[4]; };

// An op's thread indices pass through an empty asm: everything derived from them is recomputed inside the op.  Without it the
// compiler hoists every op's loop-invariant address arithmetic (hundreds of per-thread LDS offsets) in front of the op loop and
// spills it (200 registers of scratch at the 128 a 1024-thread workgroup leaves each thread).
__device__ __forceinline__ int c8_fresh(int v) { asm volatile("" : "+v"(v)); return v; }

// the end of an 8x8 op: matrix-wave thread (tile tt = tid / 32, channel col = tid % 32) holds y[k], k = 2 dy + dx of its tile's 2 x 2
// pixels.  A quad's lanes hold four consecutive channels: sixteen DPP moves give every lane all of the quad's values, lane j of the
// quad stores pixel j's four channels as 16 bytes.  Publishing as in chain_finish (waves 0..7 store and count themselves in).
__device__ __forceinline__ void c8_finish(const ChainParams& p, const ChainOp& o, ChainCtx& c, C8Keep& kp, float (&y)[4], int n0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (o.flags & CHF_ADD_KEEP) y[k] += kp.keep[k];
        if (o.flags & CHF_ADD_KEEP2) y[k] += kp.keep2[k];
        if (o.flags & CHF_SAVE_KEEP) kp.keep[k] = y[k];
        if (o.flags & CHF_SAVE_KEEP2) kp.keep2[k] = y[k];
    }
    if (!(o.flags & CHF_NO_OUT)) {
        float g[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yb = __builtin_bit_cast(int, y[k]);
            g[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0x00, 0xF, 0xF, true));
            g[k][1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0x55, 0xF, 0xF, true));
            g[k][2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0xAA, 0xF, 0xF, true));
            g[k][3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yb, 0xFF, 0xF, 0xF, true));
        }
        const int j = c.lane & 3;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = j == 0 ? g[0][q] : j == 1 ? g[1][q] : j == 2 ? g[2][q] : g[3][q];
        const int tt = c.tid >> 5, ty = tt >> 2, tx = tt & 3;
        const int pix = (2 * ty + (j >> 1)) * 8 + 2 * tx + (j & 1);
        float* dst = o.out + ((long long)c.b * 64 + pix) * o.n_out + n0 + ((c.tid & 31) & ~3);
        if (o.flags & CHF_SIGNAL) st_sc1(dst, v);
        else *reinterpret_cast<f32x4*>(dst) = v;
    }
    if (o.flags & CHF_SIGNAL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c.lane == 0) {
            unsigned* lcnt = reinterpret_cast<unsigned*>(c.misc) + 1;
            if ((atomicAdd(lcnt, 1u) & 7u) == 7u) __hip_atomic_fetch_add(p.cnt + c.b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ void c8_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// CH_CONV3 on an 8x8 map: Winograd, see the section header.  cin = 256, or 512 as two staged halves.
__device__ __forceinline__ void c8_conv3(const ChainParams& p, const ChainOp& o, ChainCtx& c, C8Keep& kp) {
    float* lds = c.lds;
    const int tid = c.tid, lane = c.lane, wave = c.wave;
    const int m = lane & 15, kq = lane >> 4;
    const int n0 = c.nt << 5;
    const int cin = o.c0 + o.c1, nch = cin >> 5, halves = cin >> 8;
    float* V = lds + L8_V;
    const bool matrix = wave < 8;

    // ---- before the wait: the first two chunks' filter units (matrix waves), the tail's operands, the next op's filter into L2
    const float* wl = o.w + ((size_t)c.nt * nch * 16 + 2 * (wave & 7)) * 1024 + lane * 4;
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
    const int tt = (tid >> 5) & 15, col = tid & 31, ch = n0 + col;
    const int tty = tt >> 2, ttx = tt & 3;
    float ga = 1.f, be = 0.f, sh = 0.f, cb = 0.f;
    if (matrix) {
        load_b(0, bA);
        load_b(1, bB);
        ga = o.gamma[ch]; be = o.beta[ch];
        if (o.temb_off >= 0) {
            const long long tr = p.temb_rows ? p.temb_rows[c.b] : c.b;
            sh = p.temb[tr * p.temb_stride + o.temb_off + ch];
        }
        if (o.bias) cb = o.bias[ch];
        if (o.flags & CHF_KEEP_FROM_SRC) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                kp.keep[k] = o.src0[((long long)c.b * 64 + (2 * tty + (k >> 1)) * 8 + 2 * ttx + (k & 1)) * o.c0 + ch];
        }
    }
    if (c.k + 1 < p.n_ops) chain_prefetch(p.op[c.k + 1], c);

    LC_STAMP(c, c.k, 0);
    if (o.flags & CHF_WAIT) chain_wait(p, c);
    else __syncthreads();
    LC_STAMP(c, c.k, 1);

    f32x4 acc[2][2];
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The two roles walk the same barriers (stage, one per chunk) in two separate loops: in one loop the register allocator would
    // keep both roles' state alive in every wave (348 spilled registers at the 128 a 1024-thread workgroup leaves each thread).
    if (!matrix) {
        // transform waves: thread = (tile t2 / 32, position row i = (t2 / 8) % 4, channel quad t2 % 8) -- conv3x3_gn_wlocal_kernel's item
        const int t2 = tid - 512;
        const int ftt = (t2 >> 5) & 15, ri = (t2 >> 3) & 3, q4i = t2 & 7;
        const int fty = ftt >> 2, ftx = ftt & 3;
        const int row_a = ri == 0 ? 0 : ri == 2 ? 2 : 1, row_b = ri == 0 ? 2 : ri == 1 ? 2 : ri == 2 ? 1 : 3;
        int poff[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = 2 * fty - 1 + (hh == 0 ? row_a : row_b), xx = 2 * ftx - 1 + j;
                const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
                poff[hh * 4 + j] = (ok ? yy * 8 + xx : 64) * L8_P + q4i * 4;
            }
        const int voff = (4 * ri * 16 + ftt) * L8_VP + q4i * 4;
        const float sgn = ri == 1 ? 1.0f : -1.0f;
        auto f4 = [](const float* q) { return *reinterpret_cast<const float4*>(q); };
        auto sub4 = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
        auto add4 = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
        auto transform = [&](int lchunk, int buf) {       // lchunk: chunk inside the staged half
            float4 r[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float4 a = f4(lds + poff[x] + (lchunk << 5)), b = f4(lds + poff[4 + x] + (lchunk << 5));
                r[x] = make_float4(fmaf(sgn, b.x, a.x), fmaf(sgn, b.y, a.y), fmaf(sgn, b.z, a.z), fmaf(sgn, b.w, a.w));
            }
            float* vb = V + buf * L8_VBUF + voff;
            *reinterpret_cast<float4*>(vb + 0 * (16 * L8_VP)) = sub4(r[0], r[2]);
            *reinterpret_cast<float4*>(vb + 1 * (16 * L8_VP)) = add4(r[1], r[2]);
            *reinterpret_cast<float4*>(vb + 2 * (16 * L8_VP)) = sub4(r[2], r[1]);
            *reinterpret_cast<float4*>(vb + 3 * (16 * L8_VP)) = sub4(r[1], r[3]);
        };
        for (int hf = 0; hf < halves; ++hf) {
            if (hf > 0) __syncthreads();                   // every reader of the first half's image and V buffers is through
            c8_stage(o, c, hf << 8, 256);
            __syncthreads();
            const int c0 = hf << 3;                        // first chunk of this half
            __builtin_amdgcn_s_setprio(3);
            transform(0, c0 & 1);
#pragma unroll 1
            for (int lc = 0; lc < 8; ++lc) {
                c8_lds_barrier();                          // V[(c0 + lc) & 1] complete, the other buffer consumed
                if (lc + 1 < 8) transform(lc + 1, (c0 + lc + 1) & 1);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    } else {
        auto chunk_mfma = [&](int chunk, const float4 (&bq)[2][2][2]) {
            const float* vb = V + (chunk & 1) * L8_VBUF + ((2 * wave) * 16 + m) * L8_VP + kq * 8;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const float4 a0 = *reinterpret_cast<const float4*>(vb + pp * (16 * L8_VP)), a1 = *reinterpret_cast<const float4*>(vb + pp * (16 * L8_VP) + 4);
                const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    acc[pp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[pp][0][0])[kk], acc[pp][0], 0, 0, 0);
                    acc[pp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], reinterpret_cast<const float*>(&bq[pp][1][0])[kk], acc[pp][1], 0, 0, 0);
                }
            }
        };
        for (int hf = 0; hf < halves; ++hf) {
            if (hf > 0) __syncthreads();
            c8_stage(o, c, hf << 8, 256);
            __syncthreads();
            if (hf == 0) LC_STAMP(c, c.k, 2);
            const int c0 = hf << 3;
#pragma unroll 1
            for (int lc = 0; lc < 8; lc += 2) {       // (not unrolled: with all eight chunks' loads hoisted the matrix waves spill)
                c8_lds_barrier();
                chunk_mfma(c0 + lc, bA);
                load_b(c0 + lc + 2, bA);
                c8_lds_barrier();
                chunk_mfma(c0 + lc + 1, bB);
                load_b(c0 + lc + 3, bB);
            }
        }
    }
    LC_STAMP(c, c.k, 3);

    // ---- M[position][tile][n] of the 8 matrix waves into LDS (over the V buffers), output transform, GroupNorm, finish
    __syncthreads();
    if (matrix) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) V[((2 * wave + pp) * 16 + kq * 4 + r) * L8_VP + nb * 16 + m] = acc[pp][nb][r];
    }
    __syncthreads();
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    float* red = lds + L8_RED;
    if (matrix) {
        float mm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) mm[k] = V[(k * 16 + tt) * L8_VP + col];
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = (mm[0 + j] + mm[4 + j]) + mm[8 + j];
            t1[j] = (mm[4 + j] - mm[8 + j]) - mm[12 + j];
        }
        v[0] = ((t0[0] + t0[1]) + t0[2]) + cb;
        v[1] = ((t0[1] - t0[2]) - t0[3]) + cb;
        v[2] = ((t1[0] + t1[1]) + t1[2]) + cb;
        v[3] = ((t1[1] - t1[2]) - t1[3]) + cb;
        // GroupNorm of the slice (32 channels x 64 pixels = the 8 matrix waves' 2048 values): per wave {mean, M2} of its 256, merged exactly
        const float mw = wave_sum_dpp((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) q += (v[k] - mw) * (v[k] - mw);
        const float m2w = wave_sum_dpp(q);
        if (lane == 0) { red[2 * wave] = mw; red[2 * wave + 1] = m2w; }
    }
    __syncthreads();
    if (matrix) {
        float msum = red[0], m2 = red[1];
#pragma unroll
        for (int w = 1; w < 8; ++w) { msum += red[2 * w]; m2 += red[2 * w + 1]; }
        const float mean = msum * 0.125f;
        float dev = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const float dm = red[2 * w] - mean; dev += dm * dm; }
        const float var = (m2 + 256.0f * dev) * (1.0f / 2048.0f);
        const float rstd = 1.0f / sqrtf(var + p.gn_eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = mish_f((v[k] - mean) * rstd * ga + be) + sh;
        c8_finish(p, o, c, kp, v, n0);
    }
    if (o.flags & CHF_SIGNAL) ++c.signals;
    LC_STAMP(c, c.k, 4);
}

// CH_CONV1 on an 8x8 map: [64 x cin] x [cin x 32] direct on the matrix waves, wave w = chunks w, w + 8, ...; four 16-row M blocks per
// wave, partial tiles [wave][64][36] meet in the V buffers
__device__ __forceinline__ void c8_conv1(const ChainParams& p, const ChainOp& o, ChainCtx& c, C8Keep& kp) {
    float* lds = c.lds;
    const int tid = c.tid, lane = c.lane, wave = c.wave;
    const int m = lane & 15, kq = lane >> 4;
    const int n0 = c.nt << 5;
    const int cin = o.c0 + o.c1, nch = cin >> 5, halves = cin > 256 ? 2 : 1, hcols = cin > 256 ? 256 : cin, hch = hcols >> 5;
    const bool matrix = wave < 8;
    float* V = lds + L8_V;
    const float* wl = o.w + (size_t)c.nt * nch * 1024 + lane * 4;
    const int tt = (tid >> 5) & 15, col = tid & 31, ch = n0 + col;
    float cb = 0.f;
    if (matrix && o.bias) cb = o.bias[ch];
    if (c.k + 1 < p.n_ops) chain_prefetch(p.op[c.k + 1], c);
    LC_STAMP(c, c.k, 0);
    if (o.flags & CHF_WAIT) chain_wait(p, c);
    else __syncthreads();
    LC_STAMP(c, c.k, 1);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int hf = 0; hf < halves; ++hf) {
        if (hf > 0) __syncthreads();
        c8_stage(o, c, hf << 8, hcols);
        __syncthreads();
        if (hf == 0) LC_STAMP(c, c.k, 2);
        if (matrix) {
            const int pitch = hcols + 4;
            for (int lc = wave; lc < hch; lc += 8) {
                const float* wp = wl + (size_t)(hf * 8 + lc) * 1024;
                const float4 b00 = *reinterpret_cast<const float4*>(wp), b01 = *reinterpret_cast<const float4*>(wp + 256);
                const float4 b10 = *reinterpret_cast<const float4*>(wp + 512), b11 = *reinterpret_cast<const float4*>(wp + 768);
                const float bv0[8] = {b00.x, b00.y, b00.z, b00.w, b01.x, b01.y, b01.z, b01.w};
                const float bv1[8] = {b10.x, b10.y, b10.z, b10.w, b11.x, b11.y, b11.z, b11.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* ap = lds + (i * 16 + m) * pitch + (lc << 5) + kq * 8;
                    const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv0[kk], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv1[kk], acc[i][1], 0, 0, 0);
                    }
                }
            }
        }
    }
    LC_STAMP(c, c.k, 3);
    if (matrix) {     // the V buffers are free: nobody reads them in this op
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) V[(wave * 64 + i * 16 + kq * 4 + r) * LC_PP + nb * 16 + m] = acc[i][nb][r];
    }
    __syncthreads();
    if (matrix) {
        const int tty = tt >> 2, ttx = tt & 3;
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = (2 * tty + (k >> 1)) * 8 + 2 * ttx + (k & 1);
            float sacc = V[row * LC_PP + col];
#pragma unroll
            for (int w = 1; w < 8; ++w) sacc += V[(w * 64 + row) * LC_PP + col];
            y[k] = sacc + cb;
        }
        c8_finish(p, o, c, kp, y, n0);
    }
    if (o.flags & CHF_SIGNAL) ++c.signals;
    LC_STAMP(c, c.k, 4);
}

// CH_ATTN on an 8x8 map: head h = nt < 4.  Projection [64 x C] x [C x 96] on the matrix waves (wave w: M block w % 4, chunks of K half
// w / 4), softmax over the 64 pixels, context, apply -- linattn_small_qkv_kernel<4>'s arithmetic on 1024 threads.
__device__ __forceinline__ void c8_attn(const ChainParams& p, const ChainOp& o, ChainCtx& c, C8Keep& kp) {
    float* lds = c.lds;
    const int tid = c.tid, lane = c.lane, wave = c.wave;
    if (c.k + 1 < p.n_ops) chain_prefetch(p.op[c.k + 1], c);
    if (c.k + 2 < p.n_ops) chain_prefetch(p.op[c.k + 2], c);
    if (c.nt < 4) {
        const int h = c.nt;
        const int m = lane & 15, kq = lane >> 4;
        const int C = o.c0, nch = C >> 5;
        const bool matrix = wave < 8;
        float* xs = lds;
        float* ks = lds + L8_KS;
        float* vs = lds + L8_VS;
        float* qs = lds + L8_QS;
        float* cs = lds + L8_CS;
        float* smax = lds + L8_SMAX;
        float* rowstat = lds + L8_ROWSTAT;
        float* cfold = lds + L8_CFOLD;
        const float* wl = o.w + (size_t)h * nch * 3072 + lane * 4;
        if (tid < 192) {
            const int n = tid % 96;
            const int colw = (n >> 5) * 128 + h * 32 + (n & 31);
            cfold[tid] = tid < 96 ? o.gamma[colw] : o.beta[colw];
        }
        LC_STAMP(c, c.k, 0);
        if (o.flags & CHF_WAIT) chain_wait(p, c);
        else __syncthreads();
        LC_STAMP(c, c.k, 1);
        c8_stage(o, c, 0, C);
        __syncthreads();
        LC_STAMP(c, c.k, 2);
        {   // LayerNorm statistics of the 64 pixel rows: 16 threads per row, two passes over the resident row
            const int row = tid >> 4, sub = tid & 15;
            float s1 = 0.f;
            for (int chn = sub * 4; chn < C; chn += 64) {
                const float4 v = *reinterpret_cast<const float4*>(xs + row * L8_P + chn);
                s1 += (v.x + v.y) + (v.z + v.w);
            }
#pragma unroll
            for (int of = 1; of < 16; of <<= 1) s1 += __shfl_xor(s1, of, 64);
            const float inv_c = 1.0f / (float)C;
            const float mean = s1 * inv_c;
            float s2 = 0.f;
            for (int chn = sub * 4; chn < C; chn += 64) {
                const float4 v = *reinterpret_cast<const float4*>(xs + row * L8_P + chn);
                const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
#pragma unroll
            for (int of = 1; of < 16; of <<= 1) s2 += __shfl_xor(s2, of, 64);
            if (sub == 0) {
                const float r = 1.0f / (sqrtf(s2 * inv_c) + p.ln_eps);
                rowstat[2 * row] = r;
                rowstat[2 * row + 1] = r * mean;
            }
        }
        f32x4 acc[6];
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mb = wave & 3, khalf = (wave >> 2) & 1;
        if (matrix) {
            for (int chunk = khalf; chunk < nch; chunk += 2) {
                const float* wp = wl + (size_t)chunk * 3072;
                const float* ap = xs + (mb * 16 + m) * L8_P + (chunk << 5) + kq * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int nb = 0; nb < 6; ++nb) {
                    const float4 b0 = *reinterpret_cast<const float4*>(wp + nb * 512), b1 = *reinterpret_cast<const float4*>(wp + nb * 512 + 256);
                    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc[nb], 0, 0, 0);
                    if (nb == 2) __builtin_amdgcn_sched_barrier(0);     // three n blocks' filter fragments in flight, not six
                }
            }
        }
        LC_STAMP(c, c.k, 3);
        __syncthreads();                                    // image consumed, rowstat written
        if (matrix) {
#pragma unroll
            for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[(khalf * 64 + mb * 16 + kq * 4 + r) * LC_QP + nb * 16 + m] = acc[nb][r];
        }
        __syncthreads();
        for (int e = tid; e < 64 * 96; e += 1024) {
            const int row = e / 96, n = e - row * 96;
            const float sacc = xs[row * LC_QP + n] + xs[(64 + row) * LC_QP + n];
            const float v = rowstat[2 * row] * sacc - rowstat[2 * row + 1] * cfold[n] + cfold[96 + n];
            const int sel = n >> 5, d = n & 31;
            if (sel == 0) qs[row * 33 + d] = v;
            else if (sel == 1) ks[row * 32 + d] = v;
            else vs[row * 32 + d] = v;
        }
        __syncthreads();
        if (tid < 256) {     // column maxima of k over the 64 pixels: 8 row groups, then 8 -> 1
            const int d = tid & 31, ng = tid >> 5;
            float mx = ks[(ng * 8) * 32 + d];
#pragma unroll
            for (int n = 1; n < 8; ++n) mx = fmaxf(mx, ks[(ng * 8 + n) * 32 + d]);
            smax[ng * 32 + d] = mx;
        }
        __syncthreads();
        if (tid < 32) {
            float mx = smax[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, smax[j * 32 + tid]);
            smax[256 + tid] = mx;
        }
        __syncthreads();
        ks[tid] = __expf(ks[tid] - smax[256 + (tid & 31)]);
        ks[tid + 1024] = __expf(ks[tid + 1024] - smax[256 + (tid & 31)]);
        __syncthreads();
        {   // ctx[d][e] = sum_n p[n][d] v[n][e] / sum_n p[n][d]: one output per thread
            const int d = tid >> 5, e = tid & 31;
            float a = 0.f, den = 0.f;
#pragma unroll 4
            for (int n = 0; n < 64; ++n) {
                const float kd = ks[n * 32 + d];
                a += kd * vs[n * 32 + e];
                den += kd;
            }
            cs[d * 36 + e] = a * (1.0f / den);
        }
        __syncthreads();
        if (matrix) {
            const int tt = (tid >> 5) & 15, e = tid & 31;
            const int tty = tt >> 2, ttx = tt & 3;
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            const int r00 = (2 * tty) * 8 + 2 * ttx;
#pragma unroll 4
            for (int d = 0; d < 32; ++d) {
                const float cv = cs[d * 36 + e];
                y[0] += qs[r00 * 33 + d] * cv;
                y[1] += qs[(r00 + 1) * 33 + d] * cv;
                y[2] += qs[(r00 + 8) * 33 + d] * cv;
                y[3] += qs[(r00 + 9) * 33 + d] * cv;
            }
            c8_finish(p, o, c, kp, y, h * 32);
        }
    } else {
        if (o.flags & CHF_WAIT) chain_wait(p, c);        // see chain_attn: no arrival for this op before every arrival of the one before
        if ((o.flags & CHF_SIGNAL) && tid == 0) __hip_atomic_fetch_add(p.cnt + c.b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (o.flags & CHF_SIGNAL) ++c.signals;
    LC_STAMP(c, c.k, 4);
}

__global__ __launch_bounds__(1024) void level8_chain_kernel(const ChainParams p) {
    extern __shared__ __align__(16) float lds[];
    ChainCtx c;
    c.lds = lds;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
    c.nt = blockIdx.x & 7;
    c.misc = lds + L8_MISC;
    c.pf_dst = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + L8_PF);
    c.nwaves = 16;
    c.units3 = 16;
    c.watch = blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : blockIdx.x == 255 ? 2 : -1;
    if (c.watch >= 0) c.watch += p.n_ops == 6 ? 3 : 6;
    c.k = 0;
    c.keep = c.keep2 = 0.f;
    if (c.tid == 0) { *reinterpret_cast<int*>(c.misc) = 0; reinterpret_cast<unsigned*>(c.misc)[1] = 0u; }
    __syncthreads();
    const int slots = gridDim.x >> 3;
    c.slot = blockIdx.x >> 3;
    c.slots = slots;
    for (int b = blockIdx.x >> 3; b < p.B; b += slots) {
        c.b = b;
        c.signals = 0;
        C8Keep kp;
#pragma unroll
        for (int k = 0; k < 4; ++k) kp.keep[k] = kp.keep2[k] = 0.f;
        chain_prefetch(p.op[0], c);
        for (int k = 0; k < p.n_ops; ++k) {
            const ChainOp& o = p.op[k];
            c.k = k;
            c.tid = c8_fresh(c.tid);
            c.lane = c.tid & 63;
            c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
            if (o.kind == CH_ATTN) c8_attn(p, o, c, kp);
            else if (o.kind == CH_CONV1) c8_conv1(p, o, c, kp);
            else c8_conv3(p, o, c, kp);
        }
        __syncthreads();
        if (c.tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.done + b * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == 7u) {
                __hip_atomic_store(p.cnt + b * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.done + b * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

bool level_chain_device_ok() { return conv_wino_cluster_device_ok(); }

int level_chain_init_device() {
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(level_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(level8_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return DDK_OK;
}

int level_chain_launch(const ChainParams& p, hipStream_t st) {
    DDK_REQUIRE(p.n_ops > 0 && p.n_ops <= CH_MAX_OPS && p.B > 0 && p.cnt && p.done && p.fail && (p.hw == 16 || p.hw == 64), "level_chain: arguments");
    for (int k = 0; k < p.n_ops; ++k) {
        const ChainOp& o = p.op[k];
        const int cin = o.c0 + o.c1;
        DDK_REQUIRE(o.src0 && o.w && aligned16(o.src0) && aligned16(o.src1) && aligned16(o.w) && aligned16(o.out), "level_chain: op pointers");
        DDK_REQUIRE((cin == 128 || cin == 256 || cin == 512) && o.c0 % 4 == 0 && o.c1 % 4 == 0 && (o.c1 == 0 || o.src1), "level_chain: op channels");
        if (p.hw == 64) {
            DDK_REQUIRE(o.kind != CH_CONV3 || cin == 256 || (cin == 512 && o.c0 == 256), "level_chain: 8x8 conv3x3 takes 256 channels, or 256 + 256");
            DDK_REQUIRE(o.kind != CH_CONV1 || cin == 128 || cin == 256 || (cin == 512 && o.c0 == 256), "level_chain: 8x8 conv1x1 channels");
            DDK_REQUIRE(o.kind != CH_ATTN || cin == 256, "level_chain: 8x8 attention takes 256 channels");
        }
        DDK_REQUIRE((o.flags & CHF_NO_OUT) || o.out, "level_chain: op output");
        if (o.kind == CH_ATTN) DDK_REQUIRE(o.c1 == 0 && o.n_out == 128 && o.gamma && o.beta, "level_chain: attention op");
        else DDK_REQUIRE(o.n_out == 256 && (o.kind != CH_CONV3 || (o.flags & CHF_NO_GN) || (o.gamma && o.beta)), "level_chain: conv op (8 slices of 32 channels)");
        if (o.flags & CHF_DOWN) DDK_REQUIRE(p.hw == 16 && o.kind == CH_CONV3 && cin == 256 && o.c1 == 0, "level_chain: stride-2 conv (4x4 chain, 256 channels)");
        if (o.kind == CH_UPT) DDK_REQUIRE(p.hw == 16 && cin == 256 && o.c1 == 0, "level_chain: transpose conv (4x4 chain, 256 channels)");
    }
    DDK_TRY(ensure_device_init());
    const int slots = p.B < 32 ? p.B : 32;
    if (p.hw == 64) {
        hipLaunchKernelGGL(level8_chain_kernel, dim3(8 * slots), dim3(1024), (size_t)L8_FLOATS * 4, st, p);
        return check_launch("level8_chain_kernel");
    }
    hipLaunchKernelGGL(level_chain_kernel, dim3(8 * slots), dim3(512), level_chain_lds_bytes(), st, p);
    return check_launch("level_chain_kernel");
}

}  // namespace ddk

// A level chain from a caller-made op list (tests/test_level_chain_gpu.py drives single ops and short chains of both kernels against
// the stand-alone kernels they replace).  ops: n records laid out like ddk::ChainOp (7 pointers, 6 ints); counters: 64 * B + 16 zeroed
// words of device memory ([B] arrivals, [B] departures, 32 words apart, then the give-up word).  Not part of the product surface.
extern "C" int ddk_debug_level_chain(const void* ops, int n, int hw, int B, const float* temb, int temb_stride, void* counters, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(ops && n > 0 && n <= CH_MAX_OPS && counters, "debug_level_chain: arguments");
    if (!level_chain_device_ok()) return fail_arg("debug_level_chain: needs a whole MI355X");
    ChainParams p{};
    const ChainOp* src = static_cast<const ChainOp*>(ops);
    for (int i = 0; i < n; ++i) p.op[i] = src[i];
    p.n_ops = n;
    p.B = B;
    p.hw = hw;
    p.temb = temb;
    p.temb_rows = nullptr;
    p.temb_stride = temb_stride;
    p.cnt = static_cast<unsigned*>(counters);
    p.done = p.cnt + (size_t)B * 32;
    p.fail = p.cnt + (size_t)B * 64;
    p.gn_eps = 1e-5f;
    p.ln_eps = 1e-5f;
    return level_chain_launch(p, as_stream(s));
}

#ifdef DDK_TUNING
extern "C" int ddk_debug_read_lc_stamps(unsigned long long* host_out) {   // tuning build only (not in include/ddk.h)
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ddk::g_lc_stamps), sizeof(unsigned long long) * 9 * ddk::CH_MAX_OPS * 8) == hipSuccess ? 0 : -2;
}
#endif
