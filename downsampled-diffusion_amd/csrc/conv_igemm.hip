// conv_igemm.hip -- the conv family of the UNet as one fp32-MFMA implicit GEMM (gfx950).
//
// Replaces the ATen conv dispatches of reference models/unet/blocks.py:35 (ConvTranspose2d k4 s2 p1),
// :44 (Conv2d k3 s2 p1), :78 (Conv2d k3 p1), :103,123-124 (1x1) and unet.py:97 (torch.cat feeding a conv).
//
// GEMM view: rows m = output pixels (b, y, x) of one output phase, columns n = output channels,
// contraction k = (tap, input channel).  Activations are NHWC, so for one tap the 32 channels of a
// k-chunk are one contiguous 128-byte segment of an input pixel; weights are packed [n][tap][cin] so a
// weight row's k-chunk is contiguous too.  Both operand tiles are staged global -> registers -> LDS as
// [row][32 k] with a 36-float row pitch (conflict-free ds_read_b128 fragments and ds_write_b128 stores),
// double buffered, one barrier per k-chunk.  The inner product is v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 FLOP/clk/SIMD = the chip's fp32 peak); each lane feeds it from one ds_read_b128 per 4 MFMA steps by
// contracting k in the order {8q+e, 8q+4+e} (same permutation on both operands).
//
// A second source pointer lets the up-path read (x, skip) without materialising the concat.  Small-M
// layers (8x8, 4x4 latents) are split along k across workgroups into fp32 slabs and reduced by a second
// tiny kernel in a fixed order, so results are run-to-run deterministic (no float atomics).
#include <cstdlib>

#include "ddk_internal.h"

namespace ddk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmParams {
    const float* src0;
    const float* src1;
    const float* w;
    const float* bias;
    const float* resid;
    float* out;
    int c0, c1, cin;
    int B, H, W;
    int Hm, Wm;
    int Ho, Wo;
    int N, M;
    int in_stride, out_scale;
    int ntaps, nphase;
    int splits, kiters, kiters_per_split;
    long long slab_stride;
    int pre_mish, post_mish;
    int debug;    // tuning only (DDK_DEBUG): 1 skip in-loop DMA, 2 skip barrier, 4 skip output stores
    int tapmode;  // 0: single tap (1x1); 1: 3x3, (dy,dx) = (tap/3-1, tap%3-1); 2: transpose-conv phase taps (py-a, px-b);
                  // 3: 4x4, (dy,dx) = (tap/4-1, tap%4-1)
};

// Input offset of a tap: pure scalar arithmetic on wave-uniform values (a lookup table in the kernel
// arguments costs a dependent memory load per k-chunk in front of the staging loads).
__device__ __forceinline__ void tap_offset(int tapmode, int phase, int tap, int& dy, int& dx) {
    if (tapmode == 1) {
        const int ty = (tap * 11) >> 5;  // tap / 3 for tap in 0..8
        dy = ty - 1;
        dx = tap - 3 * ty - 1;
    } else if (tapmode == 2) {
        dy = (phase >> 1) - (tap >> 1);
        dx = (phase & 1) - (tap & 1);
    } else if (tapmode == 3) {
        dy = (tap >> 2) - 1;
        dx = (tap & 3) - 1;
    } else {
        dy = 0;
        dx = 0;
    }
}

constexpr int LDK = 36;  // LDS row pitch in floats (32 k + one 16-byte pad)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const IgemmParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int RPP = NT / 8;  // tile rows staged per pass (8 lanes x 16 B per row)
    constexpr int A_PASS = BM / RPP, B_PASS = BN / RPP;
    constexpr int BUF = (BM + BN) * LDK;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of rows per pass");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int phase = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
    const int lrow = tid >> 3, lchunk = tid & 7;

    // ---- per-thread row bookkeeping (constant over the k loop)
    int a_iy[A_PASS], a_ix[A_PASS], a_rb[A_PASS];
    bool a_ok[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int gm = m0 + lrow + i * RPP;
        a_ok[i] = gm < p.M;
        const int xm = gm % p.Wm, tmp = gm / p.Wm;
        const int ym = tmp % p.Hm, b = tmp / p.Hm;
        a_iy[i] = ym * p.in_stride;
        a_ix[i] = xm * p.in_stride;
        a_rb[i] = b * p.H;
    }
    const float* b_row[B_PASS];
    bool b_ok[B_PASS];
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int n = n0 + lrow + i * RPP;
        b_ok[i] = n < p.N;
        b_row[i] = p.w + ((long long)(phase * p.N + (b_ok[i] ? n : 0)) * p.ntaps) * p.cin + lchunk * 4;
    }

    const int it_begin = split * p.kiters_per_split;
    const int it_end = min(p.kiters, it_begin + p.kiters_per_split);
    const int cpt = p.cin >> 5;  // k-chunks per tap
    int tap = it_begin / cpt, cc = (it_begin % cpt) << 5;

    float4 ra[A_PASS], rb[B_PASS];
    auto gload = [&]() {
        int dy, dx;
        tap_offset(p.tapmode, phase, tap, dy, dx);
        const float* src;
        int cs, coff;
        if (cc < p.c0) {
            src = p.src0; cs = p.c0; coff = cc;
        } else {
            src = p.src1; cs = p.c1; coff = cc - p.c0;
        }
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const long long off = ((long long)(a_rb[i] + iy) * p.W + ix) * cs + coff + lchunk * 4;
            ra[i] = ok ? *reinterpret_cast<const float4*>(src + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(b_row[i] + (long long)tap * p.cin + cc)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        cc += 32;
        if (cc == p.cin) { cc = 0; ++tap; }
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * BUF + lrow * LDK + lchunk * 4;
        float* Bs = smem + buf * BUF + BM * LDK + lrow * LDK + lchunk * 4;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            float4 v = ra[i];
            if (p.pre_mish) { v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w); }
            *reinterpret_cast<float4*>(As + i * RPP * LDK) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<float4*>(Bs + i * RPP * LDK) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDK + (lane >> 5) * 4;
    const int a_frag = wm * TM * 32 * LDK + frag_off;
    const int b_frag = BM * LDK + wn * TN * 32 * LDK + frag_off;

    if (it_begin < it_end) {
        gload();
        lstore(0);
    }
    __syncthreads();

    for (int it = it_begin; it < it_end; ++it) {
        const int buf = (it - it_begin) & 1;
        const bool more = it + 1 < it_end;
        if (more) gload();  // global loads for the next k-chunk fly under this chunk's MFMAs
        const float* As = smem + buf * BUF + a_frag;
        const float* Bs = smem + buf * BUF + b_frag;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * LDK + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDK + q * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool direct = p.splits == 1;
    float* outp = direct ? p.out : p.out + (long long)split * p.slab_stride;
    const int py = phase >> 1, px = phase & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int gm = m0 + row;
            if (gm >= p.M) continue;
            long long opix = gm;
            if (p.out_scale != 1) {
                const int xm = gm % p.Wm, tmp = gm / p.Wm;
                const int ym = tmp % p.Hm, b = tmp / p.Hm;
                opix = ((long long)b * p.Ho + ym * p.out_scale + py) * p.Wo + xm * p.out_scale + px;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
                if (gn >= p.N) continue;
                float v = acc[i][j][r];
                const long long o = opix * p.N + gn;
                if (direct) {
                    if (p.bias) v += p.bias[gn];
                    if (p.resid) v += p.resid[o];
                    if (p.post_mish) v = mish_f(v);
                }
                outp[o] = v;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (used whenever no Mish has to be applied while staging): operand tiles go global -> LDS
// directly with global_load_lds_dwordx4 (no VGPR staging, no ds_write), two stages, the DMA of chunk i+1 in
// flight under the MFMAs of chunk i.  An LDS-DMA writes wave-base + lane*16, so a stage is the plain
// [row][32 k] image (128-byte rows, 8 rows per wave-instruction) and bank conflicts are avoided by an XOR
// swizzle applied to the per-lane SOURCE address: position p of row r holds k-chunk p ^ ((r >> 1) & 7), and the
// fragment read applies the same involution.  Out-of-image taps (zero padding) and rows past M / N read a
// 128-byte zero page instead, so every lane always loads and there is no branch in the staging code.
//
// The DMA is issued from inline asm on purpose: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the first
// ds_read after a __builtin_amdgcn_global_load_lds (it cannot prove the DMA's target stage differs from the
// stage being read), which would drain the prefetch before the MFMAs instead of under them.  With the asm form
// the compiler does not see the load; the wait is placed by hand: vmcnt(0) + barrier at the top of each k-chunk,
// which orders every wave's DMA into stage s before any wave's reads of stage s, and every wave's reads of
// stage s^1 (finished: their MFMAs consumed them) before the next DMA into it.
__device__ __attribute__((aligned(128))) float g_zero_page[32];

// one 1-KiB piece: LDS[m0 + lane*16 .. +16) <- 16 bytes at this lane's global address
__device__ __forceinline__ void lds_dma16(const float* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(g) : "memory", "m0");
}

// STAGES = 2: the pieces of chunk k+1 are all issued right after the barrier of chunk k (small LDS footprint, so
// 2-5 workgroups share a CU and cover each other's barrier / issue phases -- the faster choice measured);
// STAGES = 3: chunk k+2 is issued piece by piece between the MFMAs of chunk k (one workgroup per CU).
template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM* WN * 64) void igemm_dma_kernel(const IgemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_PW = BM / 8 / NW, B_PW = BN / 8 / NW;  // 1-KiB DMA pieces per wave per stage
    constexpr int PW = A_PW + B_PW;
    constexpr int STAGE = (BM + BN) * 32;                  // floats
    constexpr int NMFMA = 16 * TM * TN;                    // MFMAs per wave per k-chunk
    constexpr int PIECE_EVERY = (NMFMA / 2 / PW) > 0 ? (NMFMA / 2 / PW) : 1;  // pieces go out over the first half
    static_assert(PW * PIECE_EVERY <= NMFMA, "not enough MFMA slots to issue the DMA pieces");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row pieces per wave");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
    // Give every XCD a CONTIGUOUS run of logical tiles, ordered (n-tile fastest, then m-tile, then split / phase), so
    // the workgroups sharing an L2 are the ones that re-read the same input rows (9 taps, halo rows, all n-tiles).
    int tile_m, tile_n, tile_z;
    {
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = logical % gridDim.y;
        const int rest = logical / gridDim.y;
        tile_m = rest % gridDim.x;
        tile_z = rest / gridDim.x;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int phase = tile_z / p.splits, split = tile_z % p.splits;
    const int prow = lane >> 3, ppos = lane & 7;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);  // wave-uniform, lives in an SGPR (feeds m0)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;

    // ---- DMA bookkeeping: piece j of this wave covers tile rows (wid*PW + j)*8 + prow.
    // Per lane and piece, computed once: the linear index of the pixel under the CENTRE tap, the swizzled
    // k-chunk offset, and a bit mask of the taps that fall inside the image.  Per k-chunk the tap / channel
    // offset is wave-uniform scalar arithmetic, so a piece costs a bit test, one 64-bit mad and a select.
    // (Plain int arrays on purpose: a `cond ? ptrA[j] : ptrB[j]` select of two pointer arrays defeats SROA and
    // sends the arrays to scratch, whose reloads wait vmcnt(0) and serialise the DMA.)
    int a_pix[A_PW], a_sw[A_PW];
    unsigned a_mask[A_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int r = (wid * A_PW + j) * 8 + prow;
        const int gm = m0 + r;
        const int xm = gm % p.Wm, tmp = gm / p.Wm;
        const int ym = tmp % p.Hm, b = tmp / p.Hm;
        const int iy0 = ym * p.in_stride, ix0 = xm * p.in_stride;
        a_sw[j] = (ppos ^ ((r >> 1) & 7)) * 4;  // float offset of the k-chunk this lane fetches
        a_pix[j] = (b * p.H + iy0) * p.W + ix0;
        unsigned m = 0;
        if (p.debug & 16) m = 0xffffu;
        else
        for (int t = 0; t < p.ntaps; ++t) {
            int dy, dx;
            tap_offset(p.tapmode, phase, t, dy, dx);
            if ((unsigned)(iy0 + dy) < (unsigned)p.H && (unsigned)(ix0 + dx) < (unsigned)p.W) m |= 1u << t;
        }
        a_mask[j] = gm < p.M ? m : 0u;
    }
    long long b_off[B_PW];
    bool b_ok[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int r = (wid * B_PW + j) * 8 + prow;
        const int n = n0 + r;
        b_ok[j] = n < p.N;
        b_off[j] = ((long long)(phase * p.N + (b_ok[j] ? n : 0)) * p.ntaps) * p.cin + (ppos ^ ((r >> 1) & 7)) * 4;
    }
    const float* zero = g_zero_page + ppos * 4;

    const int it_begin = split * p.kiters_per_split;
    const int it_end = min(p.kiters, it_begin + p.kiters_per_split);
    const int cpt = p.cin >> 5;
    int tap = it_begin / cpt, cc = (it_begin % cpt) << 5;

    // wave-uniform state of the k-chunk whose pieces are being issued
    const float* i_src = nullptr;
    int i_cs = 0, i_coff = 0, i_dpix = 0, i_tap = 0;
    long long i_woff = 0;
    unsigned i_st = 0;
    auto issue_begin = [&](int stage) {
        int dy, dx;
        tap_offset(p.tapmode, phase, tap, dy, dx);
        const bool first = cc < p.c0;
        i_src = first ? p.src0 : p.src1;
        i_cs = first ? p.c0 : p.c1;
        i_coff = first ? cc : cc - p.c0;
        i_dpix = dy * p.W + dx;
        i_woff = (long long)tap * p.cin + cc;
        i_tap = tap;
        i_st = lds_base + (unsigned)(stage * STAGE * 4);
        cc += 32;
        if (cc == p.cin) { cc = 0; ++tap; }
    };
    auto issue_piece = [&](int j) {  // j is a compile-time constant after unrolling
        if (j < A_PW) {
            const long long off = (long long)(a_pix[j] + i_dpix) * i_cs + (i_coff + a_sw[j]);
            const float* g = ((a_mask[j] >> i_tap) & 1u) ? i_src + off : zero;
            lds_dma16(g, i_st + (unsigned)((wid_u * A_PW + j) * 1024));
        } else {
            const int jb = j - A_PW;
            const float* g = b_ok[jb] ? p.w + (b_off[jb] + i_woff) : zero;
            lds_dma16(g, i_st + (unsigned)(BM * 128 + (wid_u * B_PW + jb) * 1024));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: row = tile base + (lane & 31); k-chunk 2q + (lane >> 5) sits at position chunk ^ f(row)
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = (lane & 31) * 32 + (((2 * q + fh) ^ fsw) << 2);
    const int a_base = wm * TM * 32 * 32;
    const int b_base = BM * 32 + wn * TN * 32 * 32;

    // ---- ring of STAGES buffers: chunk k+STAGES-1 is DMA'd while chunk k is multiplied.
    constexpr int AHEAD = STAGES - 1;
    const int n_it = (p.debug & 8) ? 0 : it_end - it_begin;
#pragma unroll
    for (int pre = 0; pre < AHEAD; ++pre) {
        if (pre < n_it) {
            issue_begin(pre);
#pragma unroll
            for (int j = 0; j < PW; ++j) issue_piece(j);
        }
    }
    int stage = 0;
    for (int k = 0; k < n_it; ++k) {
        // pieces of chunk k must have landed; with 3 stages those of chunk k+1 (the PW youngest) may stay in flight
        if (STAGES == 3 && k + 1 < n_it) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.debug & 2)) __syncthreads();  // everyone's pieces of chunk k landed; everyone finished reading chunk k-1's stage
        const bool more = k + AHEAD < n_it && !(p.debug & 1);
        int wr = stage + AHEAD;
        if (wr >= STAGES) wr -= STAGES;
        if (more) {
            issue_begin(wr);
            if (STAGES == 2) {
#pragma unroll
                for (int j = 0; j < PW; ++j) issue_piece(j);
            }
        }
        const float* As = smem + stage * STAGE + a_base;
        const float* Bs = smem + stage * STAGE + b_base;
        float4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const float4*>(As + i * 1024 + foff[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const float4*>(Bs + j * 1024 + foff[0]);
        int mf = 0;  // MFMA counter: folds to constants once the loops below are unrolled
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            if (q < 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = *reinterpret_cast<const float4*>(As + i * 1024 + foff[q + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = *reinterpret_cast<const float4*>(Bs + j * 1024 + foff[q + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs (hipcc otherwise sinks it to first use)
            // e outermost: consecutive MFMAs go to different accumulators (round-robin over the TM*TN tiles)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = e == 0 ? a[cur][i].x : e == 1 ? a[cur][i].y : e == 2 ? a[cur][i].z : a[cur][i].w;
                        const float bv = e == 0 ? b[cur][j].x : e == 1 ? b[cur][j].y : e == 2 ? b[cur][j].z : b[cur][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        ++mf;
                        if (STAGES == 3 && mf % PIECE_EVERY == 0 && mf / PIECE_EVERY <= PW) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue_piece(mf / PIECE_EVERY - 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    }

    const bool direct = p.splits == 1;
    float* outp = direct ? p.out : p.out + (long long)split * p.slab_stride;
    const int py = phase >> 1, px = phase & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int gm = m0 + row;
            if (gm >= p.M) continue;
            long long opix = gm;
            if (p.out_scale != 1) {
                const int xm = gm % p.Wm, tmp = gm / p.Wm;
                const int ym = tmp % p.Hm, b = tmp / p.Hm;
                opix = ((long long)b * p.Ho + ym * p.out_scale + py) * p.Wo + xm * p.out_scale + px;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
                if (gn >= p.N) continue;
                float v = acc[i][j][r];
                const long long o = opix * p.N + gn;
                if (direct) {
                    if (p.bias) v += p.bias[gn];
                    if (p.resid) v += p.resid[o];
                    if (p.post_mish) v = mish_f(v);
                }
                if (!(p.debug & 4)) outp[o] = v;
            }
        }
    }
}

// out[i] = sum_s slab[s][i] (fixed order) + bias[i % N] + resid[i]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits,
                                                            long long slab_stride, const float* __restrict__ bias,
                                                            const float* __restrict__ resid, float* __restrict__ out,
                                                            long long n4, int N, int post_mish) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(slabs)[i];
        for (int s = 1; s < splits; ++s) {
            const float4 u = reinterpret_cast<const float4*>(slabs + s * slab_stride)[i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + (i * 4) % N);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (resid) {
            const float4 r = reinterpret_cast<const float4*>(resid)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (post_mish) { v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w); }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
enum TileId { T128x128 = 0, T128x64 = 1, T64x64 = 2, T128x32 = 3, T64x32 = 4 };

struct Choice {
    TileId tile;
    int splits;
    int kps;  // k-chunks per split
};

struct Geometry {
    int Hm, Wm, Ho, Wo, in_stride, out_scale, ntaps, nphase;
};

static bool conv_geometry(int kind, int H, int W, Geometry& g) {
    switch (kind) {
        case DDK_CONV3X3_S1: g = {H, W, H, W, 1, 1, 9, 1}; return true;
        case DDK_CONV3X3_S2: g = {(H - 1) / 2 + 1, (W - 1) / 2 + 1, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 2, 1, 9, 1}; return true;
        case DDK_CONV1X1: g = {H, W, H, W, 1, 1, 1, 1}; return true;
        case DDK_CONVT4X4_S2: g = {H, W, 2 * H, 2 * W, 1, 2, 4, 4}; return true;
        case DDK_CONV4X4_S2: g = {H / 2, W / 2, H / 2, W / 2, 2, 1, 16, 1}; return H % 2 == 0 && W % 2 == 0;
        default: return false;
    }
}

static void tile_dims(TileId t, int& bm, int& bn) {
    switch (t) {
        case T128x128: bm = 128; bn = 128; break;
        case T128x64: bm = 128; bn = 64; break;
        case T64x64: bm = 64; bn = 64; break;
        case T128x32: bm = 128; bn = 32; break;
        default: bm = 64; bn = 32; break;
    }
}

// Pick the largest tile that still gives the 256 CUs about a full wave of workgroups; when even the
// smallest does not, split k across workgroups (at least 4 k-chunks per split).
static Choice choose_tile(long long M, int N, int nphase, int kiters) {
    // tuning knob for tools/conv_bench.py only: DDK_FORCE_TILE="<tile id>[,<splits>]"
    if (const char* f = getenv("DDK_FORCE_TILE")) {
        int t = 0, s = 1;
        if (sscanf(f, "%d,%d", &t, &s) >= 1 && t >= 0 && t <= 4) {
            int bm, bn;
            tile_dims((TileId)t, bm, bn);
            if (N % 32 == 0 && (bn <= N || bn == 32)) {
                if (s < 1) s = 1;
                if (s > kiters) s = kiters;
                Choice c{(TileId)t, 1, kiters};
                c.kps = (int)ceil_div(kiters, s);
                c.splits = (int)ceil_div(kiters, c.kps);
                return c;
            }
        }
    }
    TileId order_wide[] = {T128x128, T128x64, T64x64};
    TileId order_n64[] = {T128x64, T64x64};
    TileId order_n32[] = {T128x32, T64x32};
    TileId* order;
    int n_order;
    if (N % 64 != 0) { order = order_n32; n_order = 2; }
    else if (N % 128 != 0 && N < 128) { order = order_n64; n_order = 2; }
    else { order = order_wide; n_order = 3; }
    auto n_tiles = [&](TileId t) {
        int bm, bn;
        tile_dims(t, bm, bn);
        return ceil_div(M, bm) * ceil_div(N, bn) * nphase;
    };
    auto with_splits = [&](TileId t, long long s) {
        Choice c{t, 1, kiters};
        c.kps = (int)ceil_div(kiters, s);
        c.splits = (int)ceil_div(kiters, c.kps);
        return c;
    };
    // Rules fitted to tools/conv_bench.py sweeps on MI355X (profiles/r01_conv_sweep.txt).  The kernel runs best
    // with 2+ workgroups per CU (~512 workgroups) and at least ~18 k-chunks per workgroup:
    // 1. short contractions (1x1 convs, <= 8 k-chunks): smallest tile, no split -- prologue/epilogue bound;
    if (kiters <= 8) return with_splits(order[n_order - 1], 1);
    // 2. the largest tile that gives >= 512 workgroups without splitting k;
    for (int i = 0; i < n_order; ++i)
        if (n_tiles(order[i]) >= 512) return with_splits(order[i], 1);
    // 3. else the largest tile that reaches ~512 workgroups by splitting k with >= 18 chunks left per workgroup;
    for (int i = 0; i < n_order; ++i) {
        long long s = 1;
        while (n_tiles(order[i]) * s < 512 && s < 16) s *= 2;
        if (kiters / s >= 18) return with_splits(order[i], s);
    }
    // 4. else the smallest tile with the deepest split that keeps >= 9 chunks per workgroup.
    long long s = 16;
    while (s > 1 && (kiters / s < 9 || n_tiles(order[n_order - 1]) * s > 1024)) s /= 2;
    return with_splits(order[n_order - 1], s);
}

template <int BM, int BN, int WM, int WN>
static int launch_tile(const IgemmParams& p, hipStream_t st) {
    static const bool no_dma = getenv("DDK_NO_DMA") != nullptr;  // A/B knob for tools/conv_bench.py
    dim3 grid((unsigned)ceil_div(p.M, BM), (unsigned)ceil_div(p.N, BN), (unsigned)(p.nphase * p.splits));
    if (!p.pre_mish && !no_dma) {
        constexpr int STAGES = 2;
        constexpr size_t lds = STAGES * (size_t)(BM + BN) * 32 * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_dma_kernel<BM, BN, WM, WN, STAGES>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, STAGES>), grid, dim3(WM * WN * 64), lds, st, p);
        return check_launch("igemm_dma_kernel");
    }
    constexpr size_t lds = 2 * (size_t)(BM + BN) * LDK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), lds, st, p);
    return check_launch("igemm_kernel");
}

size_t conv_workspace_bytes(int kind, int B, int H, int W, int cin, int N) {
    Geometry g;
    if (!conv_geometry(kind, H, W, g) || cin <= 0 || cin % 32) return 0;
    const Choice c = choose_tile((long long)B * g.Hm * g.Wm, N, g.nphase, g.ntaps * (cin / 32));
    if (c.splits == 1) return 0;
    return (size_t)c.splits * B * g.Ho * g.Wo * N * sizeof(float);
}

int conv_forward(const ddk_conv_args& a, hipStream_t st) {
    Geometry g;
    DDK_REQUIRE(conv_geometry(a.kind, a.H, a.W, g), "conv kind");
    DDK_REQUIRE(a.src0 && a.weight && a.out, "conv: null src0/weight/out");
    DDK_REQUIRE(a.c0 > 0 && a.c0 % 32 == 0, "conv: c0 must be a positive multiple of 32");
    DDK_REQUIRE((a.src1 == nullptr) == (a.c1 == 0), "conv: src1/c1 mismatch");
    DDK_REQUIRE(a.c1 % 32 == 0, "conv: c1 must be a multiple of 32");
    DDK_REQUIRE(a.N > 0 && a.N % 32 == 0, "conv: N must be a positive multiple of 32");
    DDK_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "conv: B/H/W");
    DDK_REQUIRE(aligned16(a.src0) && aligned16(a.src1) && aligned16(a.weight) && aligned16(a.out) &&
                    aligned16(a.bias) && aligned16(a.resid),
                "conv: pointers must be 16-byte aligned");
    DDK_REQUIRE((long long)a.B * g.Ho * g.Wo * a.N < (1LL << 31) && (long long)a.B * a.H * a.W * (a.c0 + a.c1) < (1LL << 31),
                "conv: tensor too large for 32-bit pixel indexing");

    IgemmParams p{};
    p.src0 = a.src0; p.src1 = a.src1; p.w = a.weight; p.bias = a.bias; p.resid = a.resid; p.out = a.out;
    p.c0 = a.c0; p.c1 = a.c1; p.cin = a.c0 + a.c1;
    p.B = a.B; p.H = a.H; p.W = a.W;
    p.Hm = g.Hm; p.Wm = g.Wm; p.Ho = g.Ho; p.Wo = g.Wo;
    p.N = a.N; p.M = a.B * g.Hm * g.Wm;
    p.in_stride = g.in_stride; p.out_scale = g.out_scale;
    p.ntaps = g.ntaps; p.nphase = g.nphase;
    p.kiters = g.ntaps * (p.cin / 32);
    p.pre_mish = a.pre_mish;
    {
        static const int dbg = getenv("DDK_DEBUG") ? atoi(getenv("DDK_DEBUG")) : 0;
        p.debug = dbg;
    }
    p.post_mish = a.post_mish;
    p.slab_stride = (long long)a.B * g.Ho * g.Wo * a.N;
    p.tapmode = (a.kind == DDK_CONV1X1) ? 0 : (a.kind == DDK_CONVT4X4_S2 ? 2 : (a.kind == DDK_CONV4X4_S2 ? 3 : 1));
    const Choice c = choose_tile(p.M, p.N, p.nphase, p.kiters);
    {
        static const bool trace = getenv("DDK_TRACE") != nullptr;  // tuning aid: one line per conv launch
        if (trace) {
            int bm, bn;
            tile_dims(c.tile, bm, bn);
            fprintf(stderr, "[ddk] conv kind=%d B=%d %dx%d cin=%d N=%d M=%d kiters=%d -> tile %dx%d splits=%d (kps %d) wgs=%lld\n", a.kind,
                    a.B, a.H, a.W, p.cin, p.N, p.M, p.kiters, bm, bn, c.splits, c.kps,
                    ceil_div(p.M, bm) * ceil_div(p.N, bn) * p.nphase * c.splits);
        }
    }
    p.splits = c.splits;
    p.kiters_per_split = c.kps;
    float* final_out = a.out;
    if (c.splits > 1) {
        const size_t need = (size_t)c.splits * p.slab_stride * sizeof(float);
        if (!a.workspace || a.workspace_bytes < need) {
            set_error("conv: split-K workspace too small (%zu < %zu)", a.workspace_bytes, need);
            return DDK_ERR_WORKSPACE;
        }
        DDK_REQUIRE(aligned16(a.workspace), "conv: workspace alignment");
        p.out = static_cast<float*>(a.workspace);
    }
    int rc;
    switch (c.tile) {
        case T128x128: rc = launch_tile<128, 128, 2, 2>(p, st); break;
        case T128x64: rc = launch_tile<128, 64, 2, 2>(p, st); break;
        case T64x64: rc = launch_tile<64, 64, 2, 2>(p, st); break;
        case T128x32: rc = launch_tile<128, 32, 4, 1>(p, st); break;
        default: rc = launch_tile<64, 32, 2, 1>(p, st); break;
    }
    DDK_TRY(rc);
    if (c.splits > 1 && !a.defer_reduce) {
        const long long n4 = p.slab_stride / 4;
        const int blocks = (int)(ceil_div(n4, 256) < 2048 ? ceil_div(n4, 256) : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a.workspace),
                           c.splits, p.slab_stride, a.bias, a.resid, final_out, n4, p.N, p.post_mish);
        DDK_TRY(check_launch("splitk_reduce_kernel"));
    }
    return DDK_OK;
}

int conv_splits(int kind, int B, int H, int W, int cin, int N) {
    Geometry g;
    if (!conv_geometry(kind, H, W, g) || cin <= 0 || cin % 32) return 1;
    return choose_tile((long long)B * g.Hm * g.Wm, N, g.nphase, g.ntaps * (cin / 32)).splits;
}

double conv_flops(int kind, int B, int H, int W, int cin, int N) {
    Geometry g;
    if (!conv_geometry(kind, H, W, g)) return 0;
    return 2.0 * B * g.Hm * g.Wm * g.nphase * (double)g.ntaps * cin * N;
}

}  // namespace ddk

extern "C" size_t ddk_conv_workspace_bytes(int kind, int B, int H, int W, int cin, int N) {
    return ddk::conv_workspace_bytes(kind, B, H, W, cin, N);
}

extern "C" int ddk_conv_splits(int kind, int B, int H, int W, int cin, int N) { return ddk::conv_splits(kind, B, H, W, cin, N); }

extern "C" int ddk_conv_forward(const ddk_conv_args* a, ddk_stream_t s) {
    if (!a) return ddk::fail_arg("conv: null args");
    return ddk::conv_forward(*a, ddk::as_stream(s));
}
