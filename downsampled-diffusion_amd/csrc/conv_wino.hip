// conv_wino.hip -- 3x3 stride-1 conv as Winograd F(2x2, 3x3) on the fp32 matrix pipe (gfx950).
//
// Replaces the Conv2d(k3, p1) dispatch of reference models/unet/blocks.py:78 (and unet.py:97, the concat feeding it) where
// the channel counts make it a dense contraction.  Why: sections 3.1 / 3.1b of DESIGN.md -- the direct implicit GEMM is
// bound by the matrix pipe (MFMA busy 78 %, fp32 MFMA = the chip's fp32 rate), not by operand traffic (a CU takes in
// 60-130 GB/s from L2 when enough is in flight).  F(2x2, 3x3) needs 16 multiplies per 2x2 output tile instead of 36:
// 2.25x less MFMA work for 16/9 more weight bytes, which the DMA path has room for.
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A     per 4x4 input patch d (tiles stride 2), 3x3 filter g
//   U[p][n][c] = (G g[n][c] G^T)[p]         packed once per weight update (pack_conv_weight_wino_kernel)
//   V[p][t][c] = (B^T d[t][c] B)[p]         computed by the loader waves while they stage the input
//   M[p][t][n] = sum_c V[p][t][c] U[p][n][c]   16 independent GEMMs (positions p = 4i + j), v_mfma_f32_32x32x2_f32
//   Y[t][n]    = A^T M[:][t][n] A           register adds across the 4 accumulators of a wave + one LDS exchange
// Exact fp32 products; the transforms only add and halve, so the result differs from the direct sum by reordering only
// (tests: <= 2e-5 of the tensor's max against F.conv2d, same bar as the direct kernels).
//
// Workgroup = 32 tiles (128 output pixels, any 32 consecutive tiles of the (b, ty, tx) list) x 64 output channels,
// 4 matrix waves + 4 loader waves (the halo kernel's split).  k runs over (32-channel chunk, position row i): one STAGE =
// the 4 positions p = 4i + w of one chunk, matrix wave w multiplies position 4i + w: V image 32 tiles x 32 ch against U image
// 64 n x 32 ch into acc[i][2] -- 32 MFMAs (2048 cycles) per wave and stage, 8 accumulators (128 registers) per wave.
//   LDS: 3 stages x (V 16 KB + U 32 KB) = 144 KB; rows of 128 bytes, k-chunk position XOR-swizzled by (row >> 1) & 7
//   loader thread (tile t = id >> 3, channel quad cq = id & 7): 16 float4 global loads per chunk (its 4x4 patch), ~100 adds,
//     then one ds_write_b128 per position; row i of V goes out during stage i-1 (row 0 of the next chunk during stage 3)
//   U image of position 4i + w (8 LDS-DMA pieces): pieces 0..3 by loader wave w, pieces 4..7 by matrix wave w (one per
//     quarter, in MFMA shadows), two stages ahead; SGPR base + 32-bit lane offset, no VALU address arithmetic
//   one s_barrier per stage: B(S) = V of stages <= S+1 written, U of stages <= S landed, everyone done reading stage S-1;
//     before it a wave waits only for the DMA it issued one stage earlier (s_waitcnt vmcnt(4)); the matrix waves prefetch
//     stage S+1's first A fragments in the last quarter of stage S.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "conv_common.h"
#include "pack_elems.h"

namespace ddk {

__device__ __attribute__((aligned(128))) float g_wino_zero[32];

struct WinoParams {
    const float* src0;
    const float* src1;
    const float* wu;       // [cin/32][16][N][32]
    const float* bias;
    const float* resid;
    float* out;
    int c0, c1, cin;
    int B, H, W, TH, TW;
    int N, tiles;          // tiles = B * TH * TW
    int splits, chunks, chunks_per_split;
    long long slab_stride;
    int post_mish;
    FastDivU dTW, dTH;
    float2* gn_part;       // optional: per (m tile, GroupNorm group) {mean, M2} of the output (splits == 1 only)
    int cpg, groups;       // channels per group, groups in N
    // CL variant (GroupNorm + Mish + shift + residual finished in THIS launch): the np workgroups that hold one image's
    // m tiles of the same n tile exchange their {mean, M2} records through memory and normalise their own tile in registers
    const float* gn_gamma;
    const float* gn_beta;
    const float* gn_temb;          // [rows][gn_temb_stride] or null
    const long long* gn_temb_rows; // row of image b (sampler: its timestep), null = b
    int gn_temb_stride;
    float gn_eps;
    float* cl_rec;                 // [m tile][n tile][32 floats]: 64-byte record, {mean, M2} of up to 8 groups, alone on its 128-byte line
    unsigned* cl_cnt;              // [image][n tile][16]: [0] arrivals, [1] departures (self-resetting)
    unsigned* cl_fail;             // workgroups of launches on THIS workspace that gave up waiting (sticky; ddk_unet_cluster_check)
    int cl_np;                     // m tiles per image = workgroups per cluster
    float* cl_slab;                // CL with splits > 1: [splits - 1][B H W N] partial tiles of the workgroups split >= 1
    unsigned* cl_pair;             // ... and [m tile][n tile][16] arrival counters of those workgroups (zero between launches)
    const float* r1_x;             // optional addend of the in-launch GroupNorm: a 1x1 conv of this narrow tensor [pixels][r1_cin] ...
    const float* r1_w;             // ... with weight rows [N][r1_ld] and bias r1_b [N] (WinoGnFuse::res_*)
    const float* r1_b;
    int r1_cin, r1_ld;
};

constexpr int WBT = 32, WBN = 64;                        // tiles and output channels per workgroup
constexpr int W_V = 4 * WBT * 32, W_U = 4 * WBN * 32;    // floats per stage: 4 V images, 4 U images
constexpr int W_STAGE = W_V + W_U, W_NS = 3;
constexpr int W_LDS_FLOATS = W_NS * W_STAGE;             // 36864 floats = 144 KB
constexpr int W_TP = WBN + 4;                            // epilogue staging pitch
#ifndef DDK_WINO_LP
#define DDK_WINO_LP 4
#endif
constexpr int W_LP = DDK_WINO_LP;                        // of a U image's 8 DMA pieces per stage, the loader wave issues W_LP, the matrix wave the rest
static_assert(4 * 2 * WBT * W_TP <= W_LDS_FLOATS, "epilogue staging fits");

// G g G^T for one (n, c), forward or input-gradient (DGRAD) filter: pack_elems.h
template <bool DGRAD>
__global__ __launch_bounds__(256) void pack_conv_weight_wino_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I,
                                                                     int i_pad, long long total, int lo, int wi) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        pack_wino_elem<DGRAD>(w, dst, idx, O, I, i_pad, lo, wi);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // native vector: usable as an inline-asm register operand

// Raw buffer descriptor (wave-uniform) over `bytes` bytes at p: a lane whose offset is >= bytes reads zeros -- the zero padding
// of the convolution costs no address arithmetic, no select and no branch in the loader.
__device__ __forceinline__ i32x4 make_srd(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
constexpr unsigned WINO_OOB = 0x80000000u;      // > any tensor this kernel takes (host: bytes < 2^31)

// 16 bytes from descriptor + lane offset + wave-uniform offset.  Issued from inline asm on purpose: hipcc cannot count the
// LDS-DMA pieces this wave issues from asm, so every load it counts itself makes it wait vmcnt(0) -- for the pieces too -- in
// front of the next use of ANY loaded register; the loader's waits are placed by hand (see the schedule below).
__device__ __forceinline__ void buf_load16(f32x4& d, unsigned voff, const i32x4& srd, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// 16-byte sc1 store, and four 16-byte sc1 loads issued and waited for in ONE asm statement (no compiler-made copy of a register a
// load is still writing): the hand-off of a partial tile between the workgroups that split a conv's channel chunks (CL variant)
typedef float wino_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1_f4(float* p, float4 v) {
    const wino_f32x4 t = {v.x, v.y, v.z, v.w};
    // s_nop 1: a store of more than 8 bytes reads its upper data registers a cycle or two after issue, and the compiler's hazard
    // recogniser does not know this asm is a store -- without the wait states it may (and did) place the next address computation in them
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void ld_sc1_f4x4(float4& a, float4& b, float4& c, float4& d, const float* pa, const float* pb, const float* pc,
                                            const float* pd) {
    wino_f32x4 x, y, z, w;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w)
                 : "v"(pa), "v"(pb), "v"(pc), "v"(pd)
                 : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w);
    c = make_float4(z.x, z.y, z.z, z.w); d = make_float4(w.x, w.y, w.z, w.w);
}
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Diagnostic stamps (DBG = 1, tuning build only): per workgroup {entry, loop start, loop end, end} in 100 MHz ticks and the
// matrix wave's cycles spent waiting at the stage barriers vs multiplying; written to a buffer nothing else reads.
__device__ unsigned long long g_wino_stamps[8 * 1024];

__device__ unsigned g_wino_cl_timeouts;       // workgroups that gave up waiting for their cluster (must stay 0; see ddk_debug_cluster_timeouts)
constexpr unsigned long long CL_TIMEOUT_TICKS = 2000000ull;   // 20 ms of the 100 MHz s_memrealtime clock: a peer is normally < 0.1 ms away

template <int DBG, bool CL = false>
__global__ __launch_bounds__(512) void conv3x3_wino_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned long long r_entry = 0;
    if (DBG) r_entry = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n, split;
    {   // XCD-aware order: every XCD gets a contiguous run of (n fastest, then m, then split) tiles
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = logical % gridDim.y;
        const int rest = logical / gridDim.y;
        tile_m = rest % gridDim.x;
        split = rest / gridDim.x;
    }
    const int t0 = tile_m * WBT, n0 = tile_n * WBN;
    const int c_begin = split * p.chunks_per_split;
    const int n_chunks = min(p.chunks, c_begin + p.chunks_per_split) - c_begin;
    const int n_stages = n_chunks * 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;

    if (wid >= 4) {
        // ================================================================ loader wave
        const int lw = wid - 4;
        const int lt = lw * 64 + lane;
        const int t = lt >> 3, cq = lt & 7;
        if (DBG != 3) __builtin_amdgcn_s_setprio(3);
        // this thread's tile and the validity of its 4x4 patch pixels
        const int g = t0 + t;
        const bool tile_ok = g < p.tiles;
        const unsigned gg = tile_ok ? (unsigned)g : 0u;
        const unsigned tmp = fdiv_u(gg, p.dTW);
        const int tx = (int)(gg - tmp * (unsigned)p.TW);
        const unsigned bq = fdiv_u(tmp, p.dTH);
        const int ty = (int)(tmp - bq * (unsigned)p.TH), b = (int)bq;
        const int py0 = 2 * ty - 1, px0 = 2 * tx - 1;
        const int pix0 = (b * p.H + py0) * p.W + px0;          // may be "negative": only used where the mask says valid
        unsigned mask = 0;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx)
                if (tile_ok && (unsigned)(py0 + dy) < (unsigned)p.H && (unsigned)(px0 + dx) < (unsigned)p.W) mask |= 1u << (dy * 4 + dx);
        // buffer descriptors of the two sources and this thread's 16 patch-pixel offsets into the current one (padding pixels:
        // an out-of-range offset, which the buffer load turns into zeros)
        const i32x4 srd0 = make_srd(p.src0, (unsigned)((long long)p.B * p.H * p.W * p.c0 * 4));
        const i32x4 srd1 = make_srd(p.src1 ? p.src1 : p.src0, (unsigned)((long long)p.B * p.H * p.W * (p.src1 ? p.c1 : p.c0) * 4));
        unsigned voff[16];
        int cur_src = -1;
        auto set_source = [&](int which) {
            const int cs = which == 0 ? p.c0 : p.c1;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx)
                    voff[dy * 4 + dx] = ((mask >> (dy * 4 + dx)) & 1u) ? (unsigned)(((pix0 + dy * p.W + dx) * cs + cq * 4) * 4) : WINO_OOB;
            cur_src = which;
        };
        const int v_off = t * 32 + ((cq ^ ((t >> 1) & 7)) << 2);     // float offset of this thread's float4 inside a V image
        // U image of position 4i + lw (64 rows x 128 B = 8 pieces of 8 rows): in the loop this loader issues pieces 0..3 and
        // matrix wave lw issues pieces 4..7 (one per quarter, in MFMA shadows) -- an LDS-DMA piece costs 100-200 issue cycles
        // next to a busy matrix pipe, and 8 per loader and stage made the loaders the critical path (tools/wino_clock.py).
        // Wave-uniform base (SGPR pair) + per-lane 32-bit offset: no VALU address arithmetic per piece.
        const int prow = lane >> 3, ppos = lane & 7;
        unsigned u_voff[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = j * 8 + prow;
            u_voff[j] = (unsigned)(((n0 + r) * 32 + ((ppos ^ ((r >> 1) & 7)) << 2)) * 4);
        }
        // patch rows (4 pixels x this thread's 4 channels each) of one channel chunk; rows are loaded two at a time so the loads
        // spread over the stages of the previous chunk.  One buffer_load per pixel, nothing else.
        auto load_rows = [&](int chunk, f32x4 (&d)[16], int ra, int rb) {
            const int cc = chunk << 5;
            const bool first = cc < p.c0;                              // wave-uniform
            if ((first ? 0 : 1) != cur_src) set_source(first ? 0 : 1);
            const unsigned soff = (unsigned)((first ? cc : cc - p.c0) * 4);
            i32x4 srd;                                                 // four scalar selects, not a branch per load
            srd.x = first ? srd0.x : srd1.x; srd.y = first ? srd0.y : srd1.y; srd.z = first ? srd0.z : srd1.z; srd.w = srd0.w;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
                if (dy == ra || dy == rb) {
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) buf_load16(d[dy * 4 + dx], voff[dy * 4 + dx], srd, soff);
                }
        };
        // the compiler must not move a use of the patch registers above this point (their loads are invisible to it)
        auto pin = [&](f32x4 (&d)[16]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(d[i]));
        };
        // Row i of V = B^T d B (the 4 positions 4i .. 4i+3) of this thread's tile / channels, straight into the V images of
        // stage buffer `buf`.  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: row i needs two patch rows only (0: d0-d2,
        // 1: d1+d2, 2: d2-d1, 3: d1-d3), so the transform is 8 float4 operations per stage instead of 32 once per chunk.
        auto write_v_row = [&](const f32x4 (&d)[16], int i, int buf) {
            f32x4 r[4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
                r[x] = i == 0 ? d[0 * 4 + x] - d[2 * 4 + x] : i == 1 ? d[1 * 4 + x] + d[2 * 4 + x]
                     : i == 2 ? d[2 * 4 + x] - d[1 * 4 + x] : d[1 * 4 + x] - d[3 * 4 + x];
            float* base = smem + buf * W_STAGE + v_off;
            // (plain adds on purpose: forcing v_pk_add_f32 -- half the instructions -- made the stage slower, 2699 vs 2654 cycles, the
            //  loader parked the matrix waves 335 instead of 100 cycles: next to the fp32 MFMA a packed add costs more than two plain ones)
            *reinterpret_cast<f32x4*>(base + 0 * (WBT * 32)) = r[0] - r[2];
            *reinterpret_cast<f32x4*>(base + 1 * (WBT * 32)) = r[1] + r[2];
            *reinterpret_cast<f32x4*>(base + 2 * (WBT * 32)) = r[2] - r[1];
            *reinterpret_cast<f32x4*>(base + 3 * (WBT * 32)) = r[1] - r[3];
        };
        auto issue_u = [&](int stage_idx, int buf, int j_begin, int j_end) {   // stage index relative to this workgroup's first
            const int chunk = c_begin + (stage_idx >> 2), pos = 4 * (stage_idx & 3) + lw;
            const float* ub = p.wu + (((long long)chunk * 16 + pos) * p.N) * 32;
            const unsigned dst = lds_base + (unsigned)((buf * W_STAGE + W_V + lw * (WBN * 32)) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j >= j_begin && j < j_end) lds_dma16_s(ub, u_voff[j], dst + (unsigned)(j * 1024));
        };

        // Schedule.  Barrier B(S) guarantees: V of stages <= S+1 written (so a matrix wave prefetches its next A fragments across
        // the barrier) and U of stages <= S landed.  In iteration S (after B(S)) a loader writes V row (S+2) & 3 into the buffer
        // stage S-1 just released and issues its half of U(S+2) into it.  Its work is spread evenly over the stages: 8 patch loads
        // of the NEXT chunk in iterations s = 0 and 1, one V row (8 float4 adds + 4 ds_write_b128) and 4 DMA pieces per iteration.
        // Nothing waits for what it has just issued: before B(S+1) a loader waits for the pieces of iteration S-1 only
        // (s_waitcnt vmcnt(4 + the loads of this iteration)); loads are issued BEFORE the pieces of their iteration, so they are
        // forced complete one iteration later, long before their first use.
        int b0 = 0;                                                         // buffer of stage S
        // one chunk = iterations s = 0..3; `cur` holds the patch of chunk (S + 2) / 4 - for s < 2 that is chunk qi, from s = 2 on
        // chunk qi + 1, loaded into `nxt` during s = 0, 1 (the two arrays swap roles every chunk: static register allocation)
        auto chunk_body = [&](f32x4 (&cur)[16], f32x4 (&nxt)[16], int qi) {
            const bool has_next = qi + 1 < n_chunks;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int S = qi * 4 + s;
                const int buf2 = b0 == 0 ? 2 : b0 - 1;                      // (S + 2) % 3 == (S - 1) % 3
                const bool more = S + 2 < n_stages;
                const bool loads = has_next && s < 2;
                if (loads) load_rows(c_begin + qi + 1, nxt, s == 0 ? 0 : 1, s == 0 ? 2 : 3);
                if (more) {
                    // rows 0 / 2 of `nxt` were requested in iteration s = 0, rows 1 / 3 in s = 1; the wait in front of the barrier
                    // that ended iteration s = 1 (resp. 2) left only younger operations outstanding, so they have landed
                    if (s == 0) write_v_row(cur, 2, buf2);
                    else if (s == 1) write_v_row(cur, 3, buf2);
                    else if (s == 2) { pin(nxt); write_v_row(nxt, 0, buf2); }
                    else write_v_row(nxt, 1, buf2);
                    if (DBG != 4) issue_u(S + 2, buf2, 0, W_LP);
                }
                if (more && DBG != 4) { if (loads) wait_vmcnt<W_LP + 8>(); else wait_vmcnt<W_LP>(); }
                else { if (loads) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                               // B(S+1)
                b0 = b0 == 2 ? 0 : b0 + 1;
            }
        };
        f32x4 da[16], db[16];
        load_rows(c_begin, da, 0, 2);
        load_rows(c_begin, da, 1, 3);
        issue_u(0, 0, 0, 8);
        issue_u(1, 1, 0, 8);
        wait_vmcnt<16>();                                                   // the 16 patch loads (older than the 16 pieces)
        pin(da);
        write_v_row(da, 0, 0);
        write_v_row(da, 1, 1);
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // B0: stages 0 and 1 complete
        if (DBG == 2) return;                                               // ablation: matrix waves alone (terminated waves leave the barrier count)
        for (int qi = 0; qi < n_chunks; qi += 2) {
            chunk_body(da, db, qi);
            if (qi + 1 < n_chunks) chunk_body(db, da, qi + 1);
        }
    } else {
        // ================================================================ matrix wave
        const int w = wid;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
        int foff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) foff[q] = (lane & 31) * 32 + (((2 * q + fh) ^ fsw) << 2);
        const int a_img = w * (WBT * 32), b_img = W_V + w * (WBN * 32);

        // this wave's half of the U DMA: rows 32..63 of the image of position 4i + w (pieces 4..7), one piece per quarter
        const int prow = lane >> 3, ppos = lane & 7;
        unsigned u_voff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = ((W_LP + j) & 7) * 8 + prow;
            u_voff[j] = (unsigned)(((n0 + r) * 32 + ((ppos ^ ((r >> 1) & 7)) << 2)) * 4);
        }
        const unsigned u_dst = lds_base + (unsigned)((W_V + w * (WBN * 32) + (W_LP & 7) * 256) * 4);

        __builtin_amdgcn_s_barrier();                                       // B0: stages 0, 1 complete
        unsigned long long r_loop0 = 0, c_wait = 0, c_mma = 0;
        if (DBG) r_loop0 = __builtin_amdgcn_s_memrealtime();
        int buf = 0, S = 0;
        float4 a[2], b[2][2];
        a[0] = *reinterpret_cast<const float4*>(smem + a_img + foff[0]);
        for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i, ++S) {
                const float* As = smem + buf * W_STAGE + a_img;
                const float* Bs = smem + buf * W_STAGE + b_img;
                const int nbuf = buf == 2 ? 0 : buf + 1, pbuf = buf == 0 ? 2 : buf - 1;
                const float* An = smem + nbuf * W_STAGE + a_img;             // V of stage S+1: complete since B(S)
                const bool more = S + 2 < n_stages;                          // wave-uniform
                // U(S+2), position 4 * ((S+2) & 3) + w, of chunk (S+2) / 4, into the buffer stage S-1 released
                const int pos2 = 4 * ((i + 2) & 3) + w;
                const float* ub = p.wu + (((long long)(c_begin + ((S + 2) >> 2)) * 16 + pos2) * p.N) * 32;
                const unsigned dst2 = u_dst + (unsigned)(pbuf * W_STAGE * 4);
                b[0][0] = *reinterpret_cast<const float4*>(Bs + foff[0]);
                b[0][1] = *reinterpret_cast<const float4*>(Bs + 1024 + foff[0]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cur = q & 1, nxt = cur ^ 1;
                    if (q < 3) {
                        a[nxt] = *reinterpret_cast<const float4*>(As + foff[q + 1]);
                        b[nxt][0] = *reinterpret_cast<const float4*>(Bs + foff[q + 1]);
                        b[nxt][1] = *reinterpret_cast<const float4*>(Bs + 1024 + foff[q + 1]);
                    } else {
                        a[nxt] = *reinterpret_cast<const float4*>(An + foff[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float av = e == 0 ? a[cur].x : e == 1 ? a[cur].y : e == 2 ? a[cur].z : a[cur].w;
                            const float bv = e == 0 ? b[cur][j].x : e == 1 ? b[cur][j].y : e == 2 ? b[cur][j].z : b[cur][j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
                        if (e == 0 && more && DBG != 4 && q < 8 - W_LP) {      // its DMA pieces: one per quarter, behind the quarter's first MFMAs
                            __builtin_amdgcn_sched_barrier(0);
                            lds_dma16_s(ub, q == 0 ? u_voff[0] : q == 1 ? u_voff[1] : q == 2 ? u_voff[2] : u_voff[3], dst2 + (unsigned)(q * 1024));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (more) wait_vmcnt<8 - W_LP>(); else wait_vmcnt<0>();     // my pieces of U(S+1), issued during stage S-1, have landed
                unsigned long long tA = 0, tB = 0;
                if (DBG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tA)::"memory"); }
                __builtin_amdgcn_s_barrier();                               // B(S+1): stage S released; V <= S+2, U <= S+1 complete
                if (DBG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB)::"memory"); c_wait += tB - tA; if (c_mma == 0) c_mma = tA; }
                buf = nbuf;
            }
        }
        if (DBG && lane == 0 && w == 0) {
            const int wg = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 1023;
            unsigned long long tE;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tE)::"memory");
            g_wino_stamps[wg * 8 + 0] = r_entry;
            g_wino_stamps[wg * 8 + 1] = r_loop0;
            g_wino_stamps[wg * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            g_wino_stamps[wg * 8 + 3] = c_wait;                 // shader cycles parked at the stage barriers
            g_wino_stamps[wg * 8 + 4] = tE - c_mma;             // shader cycles from the end of stage 0's multiply to the end of the loop
            g_wino_stamps[wg * 8 + 5] = (unsigned long long)n_stages;
        }
        // ---- output transform, rows first (in registers): T[a] = sum_i A^T[a][i] M[i][w],  A^T = [[1,1,1,0],[0,1,-1,-1]]
        float* Ts = smem + w * (2 * WBT * W_TP);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float m0 = acc[0][j][r], m1 = acc[1][j][r], m2 = acc[2][j][r], m3 = acc[3][j][r];
                Ts[row * W_TP + j * 32 + (lane & 31)] = (m0 + m1) + m2;
                Ts[(WBT + row) * W_TP + j * 32 + (lane & 31)] = (m1 - m2) - m3;
            }
    }
    __syncthreads();   // all 8 waves: the T blocks of the 4 matrix waves are in LDS

    // ---- columns: Y[a][0] = T[a]_0 + T[a]_1 + T[a]_2,  Y[a][1] = T[a]_1 - T[a]_2 - T[a]_3  (T[a]_w = matrix wave w's block)
    const bool direct = p.splits == 1;
    float* outp = direct ? p.out : p.out + (long long)split * p.slab_stride;
    float4 keep[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * 512;
        const int t = item >> 5, a = (item >> 4) & 1, c4 = (item & 15) * 4;
        const int g = t0 + t, gn = n0 + c4;
        keep[it][0] = keep[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g >= p.tiles || gn >= p.N) continue;
        const float* tp = smem + (a * WBT + t) * W_TP + c4;
        const float4 q0 = *reinterpret_cast<const float4*>(tp);
        const float4 q1 = *reinterpret_cast<const float4*>(tp + 1 * (2 * WBT * W_TP));
        const float4 q2 = *reinterpret_cast<const float4*>(tp + 2 * (2 * WBT * W_TP));
        const float4 q3 = *reinterpret_cast<const float4*>(tp + 3 * (2 * WBT * W_TP));
        float4 y0 = f4add(f4add(q0, q1), q2);
        float4 y1 = f4sub(f4sub(q1, q2), q3);
        const unsigned tmp = fdiv_u((unsigned)g, p.dTW);
        const int tx = g - (int)tmp * p.TW;
        const unsigned bq = fdiv_u(tmp, p.dTH);
        const int ty = (int)tmp - (int)bq * p.TH;
        const long long o0 = ((((long long)bq * p.H + 2 * ty + a) * p.W) + 2 * tx) * p.N + gn;
        const long long o1 = o0 + p.N;
        if (direct) {
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + gn);
                y0 = f4add(y0, bb);
                y1 = f4add(y1, bb);
            }
            if (!CL && p.resid) {
                y0 = f4add(y0, *reinterpret_cast<const float4*>(p.resid + o0));
                y1 = f4add(y1, *reinterpret_cast<const float4*>(p.resid + o1));
            }
            if (p.post_mish) {
                y0 = make_float4(mish_f(y0.x), mish_f(y0.y), mish_f(y0.z), mish_f(y0.w));
                y1 = make_float4(mish_f(y1.x), mish_f(y1.y), mish_f(y1.z), mish_f(y1.w));
            }
        }
        if (!CL) {
            *reinterpret_cast<float4*>(outp + o0) = y0;
            *reinterpret_cast<float4*>(outp + o1) = y1;
        }
        keep[it][0] = y0;
        keep[it][1] = y1;
    }
    // ---- optional GroupNorm statistics of this tile (the host only asks for them when every m tile is full and lies inside one
    //      image): per group {mean, M2 about that mean} over the tile's 128 pixels x cpg channels, two passes over the registers;
    //      the consumer (gn_apply_parts_kernel) merges an image's tiles in fixed order -- GroupNorm then is one read + one write.
    if (CL || p.gn_part) {
        const int cq = tid & 15, qpg = p.cpg >> 2;     // this thread's channel quad; quads per group
        const int gl = cq / qpg;                       // group within the 64-channel tile
        float* red = smem + 4 * 2 * WBT * W_TP;        // behind the staging area; one 128-float region per pass
        // barrier on LDS traffic only: __syncthreads() would also wait for the output stores issued above (vmcnt), ~1 us
        auto group_sum = [&](float v, float* rd) {
            for (int o = 1; o < qpg; o <<= 1) v += __shfl_xor(v, o, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if ((lane & 48) == 0 && (cq & (qpg - 1)) == 0) rd[wid * 16 + gl] = v;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            float t = rd[gl];
#pragma unroll
            for (int w = 1; w < 8; ++w) t += rd[w * 16 + gl];
            return t;
        };
        auto sum4 = [](float4 v) { return (v.x + v.y) + (v.z + v.w); };
        const float inv = 1.0f / (float)(4 * WBT * p.cpg);
        const float mean = group_sum((sum4(keep[0][0]) + sum4(keep[0][1])) + (sum4(keep[1][0]) + sum4(keep[1][1])), red) * inv;
        auto sq4 = [&](float4 v) {
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            return (a * a + b * b) + (c * c + d * d);
        };
        const float m2 = group_sum((sq4(keep[0][0]) + sq4(keep[0][1])) + (sq4(keep[1][0]) + sq4(keep[1][1])), red + 128);
        if (!CL) {
            if (tid < 16 && (cq & (qpg - 1)) == 0) p.gn_part[(long long)tile_m * p.groups + n0 / p.cpg + gl] = make_float2(mean, m2);
        } else {
            // ---- cluster exchange (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table): every record byte is stored
            // sc1 by wave 0, which drains its stores (vmcnt(0)) before its lane 0 adds to the cluster's arrival counter (agent
            // scope); lane 0 polls the counter with sc1 loads, the other waves pass a workgroup barrier behind it, and every record
            // is read with sc1 loads: row 1 of that table (one lane signals for the workgroup's 8-byte sc1 stores, sc1 poll, the
            // other waves behind a workgroup barrier, 8-byte sc1 loads, hipMalloc memory, ONE workgroup per CU -- 144 KB of LDS).
            // Measured form, not an architectural guarantee; a record has its 128-byte line to itself.  The workgroups of a
            // cluster have consecutive launch indices inside one XCD's run, so they are co-resident whenever the dispatcher works
            // in order on a whole, otherwise idle device (conv_wino_cluster_device_ok gates on that); the wait is bounded by wall
            // time anyway, and a give-up is never silent: NaN output + sticky counters (ddk_unet_cluster_check).
            const int image = tile_m / p.cl_np;
            unsigned long long* rec = reinterpret_cast<unsigned long long*>(p.cl_rec) + ((long long)tile_m * gridDim.y + tile_n) * 16;
            unsigned* cnt = p.cl_cnt + ((long long)image * gridDim.y + tile_n) * 16;
            int* gave_up = reinterpret_cast<int*>(red + 256);       // LDS word: this workgroup's wait timed out
            if (tid == 0) *gave_up = 0;                             // ordered before its readers by the barrier below
            if (tid < 16 && (cq & (qpg - 1)) == 0) {
                const unsigned long long bits = (unsigned long long)__float_as_uint(mean) | ((unsigned long long)__float_as_uint(m2) << 32);
                __hip_atomic_store(rec + gl, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // what the finish needs from memory besides the records is requested BEFORE the wait (waves 1..7; wave 0 right behind
            // its counter add): affine and time shift of this thread's channel quad
            const int gn = n0 + (tid & 15) * 4;
            float4 ga, be, ts = make_float4(0.f, 0.f, 0.f, 0.f);
            auto load_affine = [&] {
                ga = *reinterpret_cast<const float4*>(p.gn_gamma + gn);
                be = *reinterpret_cast<const float4*>(p.gn_beta + gn);
                if (p.gn_temb) {
                    const int b_img = (int)((long long)t0 / ((long long)p.TH * p.TW));
                    const long long tr = p.gn_temb_rows ? p.gn_temb_rows[b_img] : b_img;
                    ts = *reinterpret_cast<const float4*>(p.gn_temb + tr * p.gn_temb_stride + gn);
                }
            };
            // ... and the residual of this thread's four output pixels
            long long oo[2];
            float4 rs[2][2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = tid + it * 512;
                const int t = item >> 5, a = (item >> 4) & 1;
                const int g = t0 + t;
                const unsigned tmp = fdiv_u((unsigned)g, p.dTW);
                const int tx = g - (int)tmp * p.TW;
                const unsigned bq = fdiv_u(tmp, p.dTH);
                const int ty = (int)tmp - (int)bq * p.TH;
                oo[it] = ((((long long)bq * p.H + 2 * ty + a) * p.W) + 2 * tx) * p.N + gn;
            }
            auto load_resid = [&] {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    rs[it][0] = rs[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.resid) {
                        rs[it][0] = *reinterpret_cast<const float4*>(p.resid + oo[it]);
                        rs[it][1] = *reinterpret_cast<const float4*>(p.resid + oo[it] + p.N);
                    }
                }
            };
            if (wid != 0) { load_affine(); load_resid(); }
            if (wid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                load_affine();
                load_resid();
                if (lane == 0 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.cl_np) {
                    // peers not all here yet: poll, bounded by WALL time (co-residency of a cluster is an assumption about the
                    // dispatcher, not a guarantee: a foreign kernel, a CU mask or a partitioned device breaks it).  A give-up is
                    // sticky and loud: the counters below make ddk_unet_cluster_check() fail and this tile's output is NaN.
                    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        __builtin_amdgcn_s_sleep(4);
                        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)p.cl_np) break;
                        if (__builtin_amdgcn_s_memrealtime() - t_begin > CL_TIMEOUT_TICKS) {
                            atomicAdd(&g_wino_cl_timeouts, 1u);
                            if (p.cl_fail) __hip_atomic_fetch_add(p.cl_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *gave_up = 1;
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            // the image's statistics: the np records of this thread's group merged in tile order (the arithmetic of
            // gn_apply_parts_kernel, so both paths give the same bits)
            const unsigned long long* r0 = reinterpret_cast<const unsigned long long*>(p.cl_rec) +
                                           ((long long)image * p.cl_np * gridDim.y + tile_n) * 16 + gl;
            const bool poisoned = *gave_up != 0;
            float rm[8], rq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                rm[i] = 0.f; rq[i] = 0.f;
                if (i < p.cl_np) {
                    const unsigned long long bits = __hip_atomic_load(r0 + (long long)i * gridDim.y * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rm[i] = __uint_as_float((unsigned)bits);
                    rq[i] = __uint_as_float((unsigned)(bits >> 32));
                }
            }
            if (tid == 0) {          // departure: the last one out re-arms the counters for the next launch
                const unsigned old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)p.cl_np - 1u) {
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            float ms = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < p.cl_np) ms += rm[i];
            const float gmean = poisoned ? __builtin_nanf("") : ms / (float)p.cl_np;
            float gm2 = 0.f, gd2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < p.cl_np) { gm2 += rq[i]; gd2 += (rm[i] - gmean) * (rm[i] - gmean); }
            const float n_i = 128.0f * (float)p.cpg;
            const float rstd = 1.0f / sqrtf((gm2 + n_i * gd2) / ((float)p.cl_np * n_i) + p.gn_eps);
            auto fin = [&](float4 v) {
                return make_float4(mish_f((v.x - gmean) * rstd * ga.x + be.x) + ts.x, mish_f((v.y - gmean) * rstd * ga.y + be.y) + ts.y,
                                   mish_f((v.z - gmean) * rstd * ga.z + be.z) + ts.z, mish_f((v.w - gmean) * rstd * ga.w + be.w) + ts.w);
            };
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                *reinterpret_cast<float4*>(p.out + oo[it]) = f4add(fin(keep[it][0]), rs[it][0]);
                *reinterpret_cast<float4*>(p.out + oo[it] + p.N) = f4add(fin(keep[it][1]), rs[it][1]);
            }
        }
    }
    if (DBG && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int wg = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 1023;
        g_wino_stamps[wg * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        g_wino_stamps[wg * 8 + 7] = 1;
    }
}

#include "conv_wino2_kernel.inc"
#include "conv_winoT_kernel.inc"

// ------------------------------------------------------------------------------------------------ host side
// Which kernel takes a shape.  F (128 output channels per workgroup: the input transform shared by two n tiles) wherever that still
// gives the chip a full round of workgroups; D (64 per workgroup) for the small maps; both are the 8-matrix-wave kernel.  The 4 + 4
// kernel above stays for comparison in the tuning build (DDK_WINO_VARIANT=0).
enum { WINO_V_OLD = 0, WINO_V_D = 1, WINO_V_F = 2 };
static int wino_variant(int B, int H, int W, int N) {
#ifdef DDK_TUNING
    if (const char* e = getenv("DDK_WINO_VARIANT")) {
        const int v = atoi(e);
        if (v == WINO_V_OLD || v == WINO_V_D || (v == WINO_V_F && N % 128 == 0)) return v;
    }
#endif
    const long long mt = ceil_div((long long)B * (H / 2) * (W / 2), WBT);
    return (N % 128 == 0 && mt * (N / 128) >= 224) ? WINO_V_F : WINO_V_D;
}
static int wino_nt(int variant) { return variant == WINO_V_F ? 128 : WBN; }
bool conv_wino_ok(int kind, int H, int W, int cin, int N) {
    return kind == DDK_CONV3X3_S1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && cin > 0 && cin % 32 == 0 && N % 64 == 0;
}

// Channel-chunk splits: one workgroup per CU needs >= 256 workgroups; every split keeps at least one chunk.
int conv_wino_splits(int B, int H, int W, int cin, int N) {
    const long long tiles = (long long)B * (H / 2) * (W / 2);
    const long long wgs = ceil_div(tiles, WBT) * (N / wino_nt(wino_variant(B, H, W, N)));
    const int chunks = cin / 32;
    int s = 1;
    while (wgs * s < 256 && s * 2 <= chunks) s *= 2;
    const int cps = (int)ceil_div(chunks, s);
    return (int)ceil_div(chunks, cps);
}

// Can the kernel emit GroupNorm partials for this shape?  One pass (no channel-chunk split), every m tile full and inside one
// image, whole groups inside an n tile, and few enough tiles per image that the consumer's merge stays trivial.
static int wino_stats_shape_parts(int B, int H, int W, int cin, int N, int groups) {     // ... whatever the channel-chunk split
    if (!conv_wino_ok(DDK_CONV3X3_S1, H, W, cin, N) || groups <= 0 || N % groups) return 0;
    const int cpg = N / groups, tpi = (H / 2) * (W / 2), nt_w = wino_nt(wino_variant(B, H, W, N));
    if (cpg % 4 || nt_w % cpg || nt_w / cpg > 8 || tpi % WBT || tpi / WBT > 32) return 0;
    return tpi / WBT;
}
int conv_wino_stats_parts(int B, int H, int W, int cin, int N, int groups) {
    if (conv_wino_splits(B, H, W, cin, N) != 1) return 0;
    return wino_stats_shape_parts(B, H, W, cin, N, groups);
}

// ---- transpose conv (conv_winoT_kernel.inc)
bool convT_wino_ok(int H, int W, int cin, int N) {
    return H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && cin > 0 && cin % 32 == 0 && N > 0 && N % WT_NT == 0;
}
// channel-chunk splits: (m tiles) x (n tiles) x 4 phases workgroups, one per CU
int convT_wino_splits(int B, int H, int W, int cin, int N) {
    const long long wgs = ceil_div((long long)B * (H / 2) * (W / 2), WBT) * (N / WT_NT) * 4;
    const int chunks = cin / 32;
    int s = 1;
    while (wgs * s < 256 && s * 2 <= chunks) s *= 2;
    const int cps = (int)ceil_div(chunks, s);
    return (int)ceil_div(chunks, cps);
}
// a: validated by conv_forward.  Writes the result, or (splits > 1) the slabs in a.workspace
int convT_wino_forward(const ddk_conv_args& a, int splits, hipStream_t st) {
    WinoParams p{};
    p.src0 = a.src0; p.wu = a.weight_wino; p.bias = a.bias; p.out = a.out;
    p.c0 = a.c0; p.c1 = 0; p.cin = a.c0;
    p.B = a.B; p.H = a.H; p.W = a.W; p.TH = a.H / 2; p.TW = a.W / 2;
    p.N = a.N; p.tiles = a.B * p.TH * p.TW;
    p.chunks = p.cin / 32;
    p.chunks_per_split = (int)ceil_div(p.chunks, splits);
    p.splits = (int)ceil_div(p.chunks, p.chunks_per_split);
    p.slab_stride = (long long)a.B * (2 * a.H) * (2 * a.W) * a.N;
    p.dTW = make_fastdiv_u((unsigned)p.TW);
    p.dTH = make_fastdiv_u((unsigned)p.TH);
    if (p.splits > 1) p.out = static_cast<float*>(a.workspace);
    dim3 grid((unsigned)ceil_div(p.tiles, WBT), (unsigned)(a.N / WT_NT), (unsigned)(4 * p.splits));
#ifdef DDK_TUNING
    if (getenv("DDK_WINO_STAMPS")) {
        hipLaunchKernelGGL(convT_wino_kernel<1>, grid, dim3(1024), WT_LDS_FL * sizeof(float), st, p);
        return check_launch("convT_wino_kernel<dbg>");
    }
#endif
    hipLaunchKernelGGL(convT_wino_kernel<0>, grid, dim3(1024), WT_LDS_FL * sizeof(float), st, p);
    return check_launch("convT_wino_kernel");
}

int conv_wino_init_device() {
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_wino_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(WT_LDS_FL * sizeof(float))));
#ifdef DDK_TUNING
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_wino_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(WT_LDS_FL * sizeof(float))));
#endif
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(W_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(W_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino2_kernel<2, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(W2<2>::LDS_FL * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino2_kernel<2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(W2<2>::LDS_FL * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino2_kernel<4, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(W2<4>::LDS_FL * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino2_kernel<4, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(W2<4>::LDS_FL * sizeof(float))));
#ifdef DDK_TUNING   // 1: stamps; 2: + loaders exit after the prologue; 3: + loaders at priority 0; 4: + no U DMA in the loop (2-4: wrong results)
#define W2_DBG_ATTR(P_, D_) DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino2_kernel<P_, D_, false>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(W2<P_>::LDS_FL * sizeof(float))))
    W2_DBG_ATTR(2, 1); W2_DBG_ATTR(2, 7); W2_DBG_ATTR(2, 9); W2_DBG_ATTR(2, 15); W2_DBG_ATTR(2, 17);
    W2_DBG_ATTR(4, 1); W2_DBG_ATTR(4, 7); W2_DBG_ATTR(4, 9); W2_DBG_ATTR(4, 15); W2_DBG_ATTR(4, 17);
#undef W2_DBG_ATTR
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(W_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(W_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(W_LDS_FLOATS * sizeof(float))));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(W_LDS_FLOATS * sizeof(float))));
#endif
    return DDK_OK;
}

// Cluster variant: can this shape finish GroupNorm inside the conv launch?  One-pass shape with partials, at most 8 m tiles per
// image, and every cluster resident at once under in-order dispatch: at most 256 workgroups per round and whole clusters per round.
int conv_wino_cluster_np(int B, int H, int W, int cin, int N, int groups) {
    const int np = conv_wino_stats_parts(B, H, W, cin, N, groups);
    if (np <= 0 || np > 8) return 0;
    const int nt_w = wino_nt(wino_variant(B, H, W, N));
    if (N / groups > 32 || nt_w % (N / groups)) return 0;       // a 64-byte record holds the n tile's <= 8 groups ... of >= 8 channels
    if (nt_w / (N / groups) > 8) return 0;
    const long long mt = (long long)B * np, nt = N / nt_w;
    if ((mt * nt) % 8) return 0;                                 // the XCD-aware order below assumes equal runs per XCD
    // a round = the first 256 launch indices = 32 logical tiles (n fastest, then m) of each XCD's run: whole clusters only
    const long long run = mt * nt / 8;                           // logical tiles per XCD
    const long long per_round = run < 32 ? run : 32;
    if (run % (np * nt) || per_round % (np * nt)) return 0;      // runs and rounds begin on cluster boundaries
    return np;
}
// ... and the shapes whose channel chunks are split over several workgroups (too few tiles for 256 workgroups otherwise): workgroup
// `split 0` of a tile collects its partners' partial tiles in the launch, then takes part in the exchange above.  Everything must be
// resident at once: at most 256 workgroups in all (one dispatch round), on the 8-matrix-wave kernels.  Returns np, *splits_out = splits.
int conv_wino_cluster_split_np(int B, int H, int W, int cin, int N, int groups, int* splits_out) {
    if (!conv_wino_ok(DDK_CONV3X3_S1, H, W, cin, N)) return 0;
    const int splits = conv_wino_splits(B, H, W, cin, N);
    if (splits_out) *splits_out = splits;
    if (splits <= 1 || splits > 4) return 0;
    const int np = wino_stats_shape_parts(B, H, W, cin, N, groups);
    if (np <= 0 || np > 8) return 0;
    const int variant = wino_variant(B, H, W, N);
    if (variant != WINO_V_D) return 0;                           // the 64-channel tile: the only one with such shapes
    const int nt_w = wino_nt(variant);
    if (N / groups > 32 || nt_w % (N / groups) || nt_w / (N / groups) > 8) return 0;
    const long long mt = (long long)B * np, nt = N / nt_w;
    if (mt * nt * splits > 256) return 0;
    return np;
}
size_t conv_wino_cluster_pair_words(int B, int H, int W, int N) {      // pair counters of the split form: 16 words per (m tile, n tile)
    return (size_t)ceil_div((long long)B * (H / 2) * (W / 2), WBT) * (size_t)(N / WBN) * 16;
}
size_t conv_wino_cluster_ws_floats(int B, int H, int W, int N) {
    const long long mt = (long long)B * (H / 2) * (W / 2) / WBT, nt = N / WBN;      // sized for the narrow tile: enough for either
    return (size_t)(mt * nt * 32 + (long long)B * nt * 16);
}

// The co-residency the exchange relies on is only plausible on a whole MI355X in SPX mode with no CU mask: 256 CUs in 8 XCDs.
// Anything else (CPX / DPX / QPX partitions, HSA_CU_MASK, ROC_GLOBAL_CU_MASK) takes the conv + GroupNorm-apply pair.
bool conv_wino_cluster_device_ok() {
    static std::atomic<int> cached[64];                      // 0 unknown, 1 ok, 2 not ok
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0, xccs = 0;
        bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus == 256;
        ok = ok && hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && xccs == 8;
        ok = ok && !getenv("HSA_CU_MASK") && !getenv("ROC_GLOBAL_CU_MASK");
        (void)hipGetLastError();
        v = ok ? 1 : 2;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v == 1;
}

// a: validated by conv_forward (shapes, alignment).  Writes the result (or, with splits > 1, the slabs in a.workspace).
bool conv_wino_variant_new(int B, int H, int W, int N) { return wino_variant(B, H, W, N) != WINO_V_OLD; }

int conv_wino_forward(const ddk_conv_args& a, int splits, hipStream_t st, const WinoGnFuse* fuse) {
    WinoParams p{};
    if (fuse) {
        const int np = splits == 1 ? conv_wino_cluster_np(a.B, a.H, a.W, a.c0 + a.c1, a.N, fuse->groups)
                                   : conv_wino_cluster_split_np(a.B, a.H, a.W, a.c0 + a.c1, a.N, fuse->groups, nullptr);
        DDK_REQUIRE(np > 0 && !a.post_mish && !a.gn_partials, "conv(wino): shape not eligible for the in-launch GroupNorm");
        if (splits > 1) {
            DDK_REQUIRE(fuse->pairs && a.workspace && aligned16(a.workspace) && !fuse->res_x,
                        "conv(wino, cluster): a split shape needs pair counters and the slab workspace");
            p.cl_slab = static_cast<float*>(a.workspace);
            p.cl_pair = fuse->pairs;
        }
        DDK_REQUIRE(fuse->gamma && fuse->beta && fuse->records && fuse->counters && aligned16(fuse->gamma) && aligned16(fuse->beta) &&
                        aligned16(fuse->temb) && fuse->temb_stride % 4 == 0 && aligned16(fuse->records) && aligned16(fuse->counters),
                    "conv(wino): in-launch GroupNorm arguments");
        p.gn_gamma = fuse->gamma; p.gn_beta = fuse->beta; p.gn_temb = fuse->temb; p.gn_temb_rows = fuse->temb_rows;
        p.gn_temb_stride = fuse->temb_stride; p.gn_eps = fuse->eps;
        p.cl_rec = fuse->records; p.cl_cnt = fuse->counters; p.cl_fail = fuse->fail; p.cl_np = np;
        if (fuse->res_x) {
            DDK_REQUIRE(fuse->res_w && fuse->res_cin > 0 && fuse->res_cin <= 8 && fuse->res_ld >= fuse->res_cin && !a.resid,
                        "conv(wino, cluster): the 1x1 addend takes <= 8 input channels and excludes a tensor addend");
            DDK_REQUIRE(wino_variant(a.B, a.H, a.W, a.N) != WINO_V_OLD, "conv(wino, cluster): the 1x1 addend exists in the 8-matrix-wave kernels only");
            p.r1_x = fuse->res_x; p.r1_w = fuse->res_w; p.r1_b = fuse->res_b; p.r1_cin = fuse->res_cin; p.r1_ld = fuse->res_ld;
        }
        p.groups = fuse->groups;
        p.cpg = a.N / fuse->groups;
    }
    if (a.gn_partials) {
        DDK_REQUIRE(conv_wino_stats_parts(a.B, a.H, a.W, a.c0 + a.c1, a.N, a.gn_groups) > 0 && splits == 1 && !a.resid && !a.post_mish,
                    "conv: gn_partials needs a shape with ddk_conv_gn_partials() > 0 and no resid / post_mish");
        DDK_REQUIRE((reinterpret_cast<uintptr_t>(a.gn_partials) & 7u) == 0, "conv: gn_partials alignment");
        p.gn_part = reinterpret_cast<float2*>(a.gn_partials);
        p.groups = a.gn_groups;
        p.cpg = a.N / a.gn_groups;
    }
    p.src0 = a.src0; p.src1 = a.src1; p.wu = a.weight_wino; p.bias = a.bias; p.resid = a.resid; p.out = a.out;
    p.c0 = a.c0; p.c1 = a.c1; p.cin = a.c0 + a.c1;
    p.B = a.B; p.H = a.H; p.W = a.W; p.TH = a.H / 2; p.TW = a.W / 2;
    p.N = a.N; p.tiles = a.B * p.TH * p.TW;
    p.chunks = p.cin / 32;
    p.chunks_per_split = (int)ceil_div(p.chunks, splits);
    p.splits = (int)ceil_div(p.chunks, p.chunks_per_split);
    p.slab_stride = (long long)a.B * a.H * a.W * a.N;
    p.post_mish = a.post_mish;
    p.dTW = make_fastdiv_u((unsigned)p.TW);
    p.dTH = make_fastdiv_u((unsigned)p.TH);
    if (p.splits > 1 && !fuse) p.out = static_cast<float*>(a.workspace);
    const int variant = wino_variant(a.B, a.H, a.W, a.N);
    if (variant != WINO_V_OLD) {
        dim3 grid2((unsigned)ceil_div(p.tiles, WBT), (unsigned)(a.N / wino_nt(variant)), (unsigned)p.splits);
        const bool f = variant == WINO_V_F;
        const size_t lds = (size_t)(f ? W2<2>::LDS_FL : W2<4>::LDS_FL) * sizeof(float);
#ifdef DDK_TUNING
        if (getenv("DDK_WINO_STAMPS")) {
            const int v = getenv("DDK_WINO_DEBUG") ? atoi(getenv("DDK_WINO_DEBUG")) : 1;
#define W2_DBG_LAUNCH(D_)                                                                                              \
    if (v == D_) {                                                                                                     \
        if (f) hipLaunchKernelGGL((conv3x3_wino2_kernel<2, D_, false>), grid2, dim3(768), lds, st, p);                 \
        else hipLaunchKernelGGL((conv3x3_wino2_kernel<4, D_, false>), grid2, dim3(768), lds, st, p);                   \
        return check_launch("conv3x3_wino2_kernel<dbg>");                                                              \
    }
            W2_DBG_LAUNCH(7) W2_DBG_LAUNCH(9) W2_DBG_LAUNCH(15) W2_DBG_LAUNCH(17)
#undef W2_DBG_LAUNCH
            if (f) hipLaunchKernelGGL((conv3x3_wino2_kernel<2, 1, false>), grid2, dim3(768), lds, st, p);
            else hipLaunchKernelGGL((conv3x3_wino2_kernel<4, 1, false>), grid2, dim3(768), lds, st, p);
            return check_launch("conv3x3_wino2_kernel<dbg>");
        }
#endif
        if (fuse) {
            if (f) hipLaunchKernelGGL((conv3x3_wino2_kernel<2, 0, true>), grid2, dim3(768), lds, st, p);
            else hipLaunchKernelGGL((conv3x3_wino2_kernel<4, 0, true>), grid2, dim3(768), lds, st, p);
            return check_launch("conv3x3_wino2_kernel<cluster>");
        }
        if (f) hipLaunchKernelGGL((conv3x3_wino2_kernel<2, 0, false>), grid2, dim3(768), lds, st, p);
        else hipLaunchKernelGGL((conv3x3_wino2_kernel<4, 0, false>), grid2, dim3(768), lds, st, p);
        return check_launch("conv3x3_wino2_kernel");
    }
    dim3 grid((unsigned)ceil_div(p.tiles, WBT), (unsigned)(a.N / WBN), (unsigned)p.splits);
#ifdef DDK_TUNING
    if (getenv("DDK_WINO_STAMPS")) {
        const int v = getenv("DDK_WINO_DEBUG") ? atoi(getenv("DDK_WINO_DEBUG")) : 1;
        if (v == 2) hipLaunchKernelGGL(conv3x3_wino_kernel<2>, grid, dim3(512), W_LDS_FLOATS * sizeof(float), st, p);
        else if (v == 3) hipLaunchKernelGGL(conv3x3_wino_kernel<3>, grid, dim3(512), W_LDS_FLOATS * sizeof(float), st, p);
        else if (v == 4) hipLaunchKernelGGL(conv3x3_wino_kernel<4>, grid, dim3(512), W_LDS_FLOATS * sizeof(float), st, p);
        else hipLaunchKernelGGL(conv3x3_wino_kernel<1>, grid, dim3(512), W_LDS_FLOATS * sizeof(float), st, p);
        return check_launch("conv3x3_wino_kernel<dbg>");
    }
#endif
    if (fuse) {
        hipLaunchKernelGGL((conv3x3_wino_kernel<0, true>), grid, dim3(512), W_LDS_FLOATS * sizeof(float), st, p);
        return check_launch("conv3x3_wino_kernel<cluster>");
    }
    hipLaunchKernelGGL(conv3x3_wino_kernel<0>, grid, dim3(512), W_LDS_FLOATS * sizeof(float), st, p);
    return check_launch("conv3x3_wino_kernel");
}

// test support: a "foreign" kernel that holds CUs (through its LDS footprint) for a bounded wall time
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ float occupy_lds[];
    occupy_lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t_begin < ticks) __builtin_amdgcn_s_sleep(32);
    if (occupy_lds[(threadIdx.x + 1) & 255] < 0.f && sink) *sink = 1;
}

unsigned conv_wino_cluster_timeouts() {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wino_cl_timeouts), sizeof(v)) != hipSuccess) return ~0u;
    return v;
}

}  // namespace ddk

extern "C" int ddk_pack_conv_weight_wino(const float* w_oihw, float* dst, int O, int I, int i_pad, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_oihw && dst && O > 0 && I > 0 && i_pad >= I && i_pad % 32 == 0, "pack_conv_weight_wino: arguments (i_pad % 32 == 0)");
    const long long total = (long long)O * i_pad;
    DDK_REQUIRE(total < (1LL << 31), "pack_conv_weight_wino: 2^31 (n, c) pairs or more");
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_conv_weight_wino_kernel<false>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_oihw,
                       dst, O, I, i_pad, total, 0, I);
    return check_launch("pack_conv_weight_wino_kernel");
}

extern "C" int ddk_pack_convT_weight_wino(const float* w_iohw, float* dst, int I, int O, int i_pad, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_iohw && dst && O > 0 && I > 0 && i_pad >= I && i_pad % 32 == 0, "pack_convT_weight_wino: arguments (i_pad % 32 == 0)");
    const long long total = (long long)O * i_pad;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_convT_weight_wino_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_iohw, dst,
                       I, O, i_pad, total);
    return check_launch("pack_convT_weight_wino_kernel");
}
extern "C" int ddk_convT_wino_splits(int B, int H, int W, int cin, int N) {
    if (!ddk::convT_wino_ok(H, W, cin, N)) return 0;
    return ddk::convT_wino_splits(B, H, W, cin, N);
}

extern "C" int ddk_pack_conv_weight_wino_dgrad(const float* w_oihw, float* dst, int O, int I, int c_lo, int c_hi, int o_pad, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(w_oihw && dst && O > 0 && I > 0 && c_lo >= 0 && c_hi > c_lo && c_hi <= I && o_pad >= O && o_pad % 32 == 0,
                "pack_conv_weight_wino_dgrad: arguments (0 <= c_lo < c_hi <= I, o_pad % 32 == 0)");
    const long long total = (long long)(c_hi - c_lo) * o_pad;
    DDK_REQUIRE(total < (1LL << 31), "pack_conv_weight_wino_dgrad: 2^31 (n, c) pairs or more");
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_conv_weight_wino_kernel<true>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), w_oihw,
                       dst, c_hi - c_lo, O, o_pad, total, c_lo, I);
    return check_launch("pack_conv_weight_wino_kernel<dgrad>");
}

#ifdef DDK_TUNING
extern "C" int ddk_debug_read_wino_timeline(unsigned long long* host_out) {   // tuning build only: [12 waves][8 stages][arrive, leave]
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ddk::g_wino_timeline), sizeof(unsigned long long) * 12 * 8 * 2) == hipSuccess ? 0 : -2;
}
extern "C" int ddk_debug_read_wino_stamps(unsigned long long* host_out) {   // tuning build only (not in include/ddk.h)
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ddk::g_wino_stamps), sizeof(unsigned long long) * 8 * 1024) != hipSuccess) return -2;
    static unsigned long long zeros[8 * 1024];
    return hipMemcpyToSymbol(HIP_SYMBOL(ddk::g_wino_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int ddk_conv_gn_partials(int B, int H, int W, int cin, int N, int groups) {
    return ddk::conv_wino_stats_parts(B, H, W, cin, N, groups);
}

extern "C" int ddk_conv_wino_splits(int B, int H, int W, int cin, int N) {
    if (!ddk::conv_wino_ok(DDK_CONV3X3_S1, H, W, cin, N)) return 0;
    return ddk::conv_wino_splits(B, H, W, cin, N);
}

/* Block of models/unet/blocks.py:75-84,110-115 in ONE Winograd launch on maps whose images span several workgroups (32x32,
 * 16x16): see WinoGnFuse / the CL variant above. */
extern "C" int ddk_conv3x3_gn_mish_cluster_ok(int B, int H, int W, int cin, int N, int groups) {
    if (!ddk::conv_wino_cluster_device_ok()) return 0;
    const int np = ddk::conv_wino_cluster_np(B, H, W, cin, N, groups);
    return np > 0 ? np : ddk::conv_wino_cluster_split_np(B, H, W, cin, N, groups, nullptr);
}
extern "C" size_t ddk_conv3x3_gn_mish_cluster_workspace_bytes(int B, int H, int W, int N) {
    if (B <= 0 || H <= 0 || W <= 0 || N <= 0 || N % ddk::WBN) return 0;
    return ((size_t)B * 8 * 16 + 16 + ddk::conv_wino_cluster_ws_floats(B, H, W, N)) * sizeof(float);
}
/* a shape whose channel chunks are split over workgroups (ddk_conv_wino_splits() > 1) needs the pair counters and (splits - 1) slabs
 * of partial tiles behind that: the whole workspace in bytes, or 0 when the shape is not of that kind */
extern "C" size_t ddk_conv3x3_gn_mish_cluster_split_workspace_bytes(int B, int H, int W, int cin, int N, int groups) {
    int splits = 1;
    if (ddk::conv_wino_cluster_split_np(B, H, W, cin, N, groups, &splits) <= 0) return 0;
    return ddk_conv3x3_gn_mish_cluster_workspace_bytes(B, H, W, N) +
           (ddk::conv_wino_cluster_pair_words(B, H, W, N) + (size_t)(splits - 1) * B * H * W * N) * sizeof(float);
}
extern "C" int ddk_conv3x3_gn_mish_cluster_check(void* workspace, int B, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(workspace && B > 0, "conv3x3_gn_mish_cluster_check: arguments");
    unsigned* word = reinterpret_cast<unsigned*>(static_cast<float*>(workspace) + (size_t)B * 8 * 16);
    unsigned v = 0;
    DDK_HIP(hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, as_stream(s)));
    DDK_HIP(hipStreamSynchronize(as_stream(s)));
    if (v == 0) return DDK_OK;
    DDK_HIP(hipMemsetAsync(word, 0, sizeof(v), as_stream(s)));
    set_error("conv3x3_gn_mish_cluster: %u workgroup(s) gave up waiting for their cluster; the output is invalid (NaN tiles)", v);
    return DDK_ERR_CLUSTER;
}

extern "C" int ddk_debug_occupy(int workgroups, int lds_bytes, int microseconds, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(workgroups > 0 && workgroups <= 4096 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && microseconds > 0 &&
                    microseconds <= 200000, "debug_occupy: 1..4096 workgroups, 1 KiB..160 KiB of LDS, at most 0.2 s");
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)workgroups), dim3(256), (size_t)lds_bytes, as_stream(s),
                       (unsigned long long)microseconds * 100ull, static_cast<unsigned*>(nullptr));
    return check_launch("occupy_kernel");
}

extern "C" int ddk_conv3x3_gn_mish_cluster(const float* src0, int c0, const float* src1, int c1, const float* weight_wino, const float* bias,
                                           const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                                           float* out, int B, int H, int W, int N, int groups, float eps, void* workspace,
                                           size_t workspace_bytes, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(src0 && weight_wino && gamma && beta && out && workspace, "conv3x3_gn_mish_cluster: null pointer");
    DDK_REQUIRE(conv_wino_cluster_device_ok() && ddk_conv3x3_gn_mish_cluster_ok(B, H, W, c0 + c1, N, groups) > 0,
                "conv3x3_gn_mish_cluster: shape or device not eligible (ddk_conv3x3_gn_mish_cluster_ok)");
    const size_t plain_bytes = ddk_conv3x3_gn_mish_cluster_workspace_bytes(B, H, W, N);
    const size_t split_bytes = ddk_conv3x3_gn_mish_cluster_split_workspace_bytes(B, H, W, c0 + c1, N, groups);
    DDK_REQUIRE(workspace_bytes >= (split_bytes ? split_bytes : plain_bytes) && aligned16(workspace), "conv3x3_gn_mish_cluster: workspace");
    float* ws = static_cast<float*>(workspace);
    // counters + the sticky give-up word behind them: zeroed per call here (a plan zeroes them once per forward / chain)
    DDK_HIP(hipMemsetAsync(ws, 0, ((size_t)B * 8 * 16 + 16) * sizeof(float), as_stream(s)));
    float* pairs = ws + plain_bytes / sizeof(float);
    if (split_bytes) DDK_HIP(hipMemsetAsync(pairs, 0, conv_wino_cluster_pair_words(B, H, W, N) * sizeof(float), as_stream(s)));
    ddk_conv_args a{};
    a.kind = DDK_CONV3X3_S1;
    a.src0 = src0; a.src1 = src1; a.c0 = c0; a.c1 = c1;
    a.weight = weight_wino;          // only validated as non-null: the Winograd path reads weight_wino
    a.weight_wino = weight_wino;
    a.bias = bias;
    a.resid = addend;
    a.out = out;
    a.B = B; a.H = H; a.W = W; a.N = N;
    WinoGnFuse f{gamma, beta, temb, nullptr, temb_stride, eps, groups, ws + (size_t)B * 8 * 16 + 16, reinterpret_cast<unsigned*>(ws),
                 reinterpret_cast<unsigned*>(ws + (size_t)B * 8 * 16)};
    if (split_bytes) {
        f.pairs = reinterpret_cast<unsigned*>(pairs);
        a.workspace = pairs + conv_wino_cluster_pair_words(B, H, W, N);
        a.workspace_bytes = split_bytes - plain_bytes - conv_wino_cluster_pair_words(B, H, W, N) * sizeof(float);
    }
    return conv_forward(a, as_stream(s), nullptr, &f);
}

