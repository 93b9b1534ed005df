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

#include "conv_common.h"

namespace ddk {

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
    float* mish_out;          // optional second output: Mish(out)
    const float* dmish_src;   // optional: out = (acc + bias) * Mish'(dmish_src) (+ resid)
    int debug;    // tuning only (DDK_DEBUG): 1 skip in-loop DMA, 2 skip barrier, 4 skip output stores
    int tapmode;  // 0: single tap (1x1); 1: 3x3, (dy,dx) = (tap/3-1, tap%3-1); 2: transpose-conv phase taps (py-a, px-b);
                  // 3: 4x4, (dy,dx) = (tap/4-1, tap%4-1)
    // Channel LayerNorm folded into a 1x1 conv (blocks.py:57-60 feeding to_qkv): with r = 1/(std + eps), W.LN(x) =
    // r (W o g) x - r mean (W g) + W b.  The weight passed in is already W o g; ln_c1 = W g, ln_c2 = W b (per output
    // channel); mean and r of every pixel row are accumulated from the staged A tile while it is multiplied.
    const float* ln_c1;  // nullptr: no LayerNorm folding
    const float* ln_c2;
    float ln_eps;
    FastDivU dWm, dHm;   // division by Wm / Hm (pixel index -> (b, y, x))
    FastDivU dImg, dWp, dRows;   // halo kernel: halo pixels per image, halo row pitch W+2, output rows per image in a tile
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
                    if (p.dmish_src) v *= mish_grad_f(p.dmish_src[o]);
                    if (p.resid) v += p.resid[o];
                    if (p.post_mish) v = mish_f(v);
                    if (p.mish_out) p.mish_out[o] = mish_f(v);
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

// Diagnostic only (DDK_DEBUG & 32): shader-clock and 100 MHz real-time stamps around the k-loop of each workgroup,
// written to a buffer of their own, never read by any kernel (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ unsigned long long g_stamps[6 * 4096];

// Shared epilogue of the DMA kernels: a wave's (TM*32) x (TN*32) accumulator block goes through LDS (pitch TN*32+8
// floats: the two half-waves of an MFMA register, 4 rows apart, land 32 banks apart) and leaves as float4 rows --
// 4x fewer store instructions than one dword store per accumulator register, and full 128/256-byte row segments.
// `Es` is this wave's private staging area; the caller has made sure no wave still reads the LDS it overlays.
template <int TM, int TN>
__device__ __forceinline__ void store_block_via_lds(const IgemmParams& p, f32x16 (&acc)[TM][TN], float* Es, int lane, int m_base,
                                                    int n_base, int split, int phase, const float* rowstat = nullptr) {
    constexpr int PITCH = TN * 32 + 8;
    constexpr int LPR = TN * 8;          // lanes per row (one float4 each)
    constexpr int RPP = 64 / LPR;        // rows per pass
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Es[row * PITCH + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool direct = p.splits == 1;
    float* outp = direct ? p.out : p.out + (long long)split * p.slab_stride;
    const int col4 = (lane % LPR) * 4;
    const int gn = n_base + col4;
    if (gn >= p.N) return;               // N % 32 == 0: a float4 is all-or-nothing
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (direct && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + gn);
    float4 lc1 = make_float4(0.f, 0.f, 0.f, 0.f), lc2 = lc1;
    if (rowstat) { lc1 = *reinterpret_cast<const float4*>(p.ln_c1 + gn); lc2 = *reinterpret_cast<const float4*>(p.ln_c2 + gn); }
    const int py = phase >> 1, px = phase & 1;
#pragma unroll 4
    for (int pass = 0; pass < TM * 32 / RPP; ++pass) {
        const int row = pass * RPP + lane / LPR;
        const int gm = m_base + row;
        if (gm >= p.M) continue;
        long long opix = gm;
        if (p.out_scale != 1) {          // transpose conv: this phase's pixels interleave into the 2x larger output
            const unsigned tmp = fdiv_u((unsigned)gm, p.dWm);
            const int xm = gm - (int)tmp * p.Wm;
            const unsigned bq = fdiv_u(tmp, p.dHm);
            const int ym = (int)tmp - (int)bq * p.Hm;
            const long long b = bq;
            opix = ((long long)b * p.Ho + ym * p.out_scale + py) * p.Wo + xm * p.out_scale + px;
        }
        float4 v = *reinterpret_cast<const float4*>(Es + row * PITCH + col4);
        if (rowstat) {                   // folded LayerNorm: rowstat[row] = (r, r * mean) of this pixel
            const float r = rowstat[2 * row], rm = rowstat[2 * row + 1];
            v.x = r * v.x - rm * lc1.x + lc2.x; v.y = r * v.y - rm * lc1.y + lc2.y;
            v.z = r * v.z - rm * lc1.z + lc2.z; v.w = r * v.w - rm * lc1.w + lc2.w;
        }
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        const long long o = opix * p.N + gn;
        if (direct && p.dmish_src) {
            const float4 ss = *reinterpret_cast<const float4*>(p.dmish_src + o);
            v.x *= mish_grad_f(ss.x); v.y *= mish_grad_f(ss.y); v.z *= mish_grad_f(ss.z); v.w *= mish_grad_f(ss.w);
        }
        if (direct && p.resid) {
            const float4 rr = *reinterpret_cast<const float4*>(p.resid + o);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (direct && p.post_mish) { v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w); }
        *reinterpret_cast<float4*>(outp + o) = v;
        if (direct && p.mish_out) {
            *reinterpret_cast<float4*>(p.mish_out + o) = make_float4(mish_f(v.x), mish_f(v.y), mish_f(v.z), mish_f(v.w));
        }
    }
}

// Keep SMEM out of the k-loops: a scalar load that is still pending at the loop header (kernel arguments only used by
// the epilogue, or an s_memtime stamp behind a never-taken branch) makes hipcc wait lgkmcnt(0) -- i.e. for the LDS
// fragments it has just requested -- in every iteration, because SMEM returns out of order.  Consuming the epilogue's
// arguments here forces their wait in front of the loop.
__device__ __forceinline__ void consume_epilogue_args(const IgemmParams& p) {
    const long long ss = p.slab_stride;
    const int sp = p.splits, pm = p.post_mish, MM = p.M, NN = p.N, os = p.out_scale, wm = p.Wm, hm = p.Hm, ho = p.Ho, wo = p.Wo;
    const int le = __float_as_int(p.ln_eps);
    asm volatile("" ::"s"(ss), "s"(sp), "s"(pm), "s"(MM), "s"(NN), "s"(os), "s"(wm), "s"(hm), "s"(ho), "s"(wo), "s"(p.out), "s"(p.bias),
                 "s"(p.resid), "s"(p.ln_c1), "s"(p.ln_c2), "s"(le), "s"(p.mish_out), "s"(p.dmish_src));
}

// im2col implicit GEMM, all waves load and multiply (every conv kind; the 3x3 stride-1 layers with enough pixels use the
// halo kernel further down).
//
// NS-stage LDS ring with counted waits.  tools/dma_rate.hip (profiles/r02_dma_rate.txt) shows what a CU's LDS-DMA intake
// is bounded by: the bytes it keeps in flight over the latency of the level that serves them (L2 hit ~300 cycles:
// 30 GB/s per CU with 4 KiB in flight, 63-79 with 16 KiB, > 100 with 32 KiB; Infinity Cache ~700 cycles) -- not a fixed
// per-CU rate.  A 2-stage ring has ONE k-chunk in flight per workgroup and only while that workgroup multiplies, so layers
// with one or two workgroups per CU ran at the latency, not the bandwidth.  Here chunks k+1 .. k+NS-2 are already in flight
// while chunk k is multiplied, and the pieces of chunk k+NS-1 are issued one at a time BETWEEN the MFMAs of chunk k (their
// address arithmetic and issue slots hide in MFMA shadows instead of stalling the matrix pipe after each barrier).
//   iteration k:  s_waitcnt vmcnt((NS-2)*PW)   this wave's pieces of chunk k have landed (PW pieces per wave and chunk)
//                 s_barrier                    everyone's have; everyone is done reading stage (k-1) % NS
//                 4 quarters of { prefetch next fragments | MFMA | issue PW/4 pieces of chunk k+NS-1 into stage (k-1) % NS | MFMAs }
template <int BM, int BN, int WM, int WN, int NS, int DBG = 0>   // DBG = 1: timeline stamps (DDK_TUNING build, DDK_DEBUG & 32)
__global__ __launch_bounds__(WM* WN * 64) void igemm_dma_kernel(const IgemmParams p) {
    unsigned long long r_entry = 0;
    if (DBG) r_entry = __builtin_amdgcn_s_memrealtime();
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_PW = BM / 8 / NW, B_PW = BN / 8 / NW;  // 1-KiB DMA pieces per wave per stage
    constexpr int PW = A_PW + B_PW;
    constexpr int STAGE = (BM + BN) * 32;                  // floats
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row pieces per wave");
    static_assert(NS >= 2 && (NS - 2) * PW <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
    // Give every XCD a CONTIGUOUS run of logical tiles, ordered (n-tile fastest, then m-tile, then split / phase), so
    // the workgroups sharing an L2 are the ones that re-read the same input rows (9 taps, halo rows, all n-tiles) and,
    // for split-k layers, the same k-slice of the weights (each XCD then pulls 1/8 of the weights from the Infinity Cache).
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
        const unsigned tmp = fdiv_u((unsigned)gm, p.dWm);
        const int xm = gm - (int)tmp * p.Wm;
        const unsigned bq = fdiv_u(tmp, p.dHm);
        const int ym = (int)tmp - (int)bq * p.Hm, b = (int)bq;
        const int iy0 = ym * p.in_stride, ix0 = xm * p.in_stride;
        a_sw[j] = (ppos ^ ((r >> 1) & 7)) * 4;  // float offset of the k-chunk this lane fetches
        a_pix[j] = (b * p.H + iy0) * p.W + ix0;
        unsigned m = 0;
        if (p.tapmode == 1) {            // 3x3: row / column validity separately, 9 ANDs (the common case, kept branch-light)
            const unsigned vy = ((unsigned)(iy0 - 1) < (unsigned)p.H ? 1u : 0u) | ((unsigned)iy0 < (unsigned)p.H ? 2u : 0u) |
                                ((unsigned)(iy0 + 1) < (unsigned)p.H ? 4u : 0u);
            const unsigned vx = ((unsigned)(ix0 - 1) < (unsigned)p.W ? 1u : 0u) | ((unsigned)ix0 < (unsigned)p.W ? 2u : 0u) |
                                ((unsigned)(ix0 + 1) < (unsigned)p.W ? 4u : 0u);
            m = ((vy & 1u) ? vx : 0u) | ((vy & 2u) ? vx << 3 : 0u) | ((vy & 4u) ? vx << 6 : 0u);
        } else {
            for (int t = 0; t < p.ntaps; ++t) {
                int dy, dx;
                tap_offset(p.tapmode, phase, t, dy, dx);
                if ((unsigned)(iy0 + dy) < (unsigned)p.H && (unsigned)(ix0 + dx) < (unsigned)p.W) m |= 1u << t;
            }
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
    const int n_it = it_end - it_begin;

    // ---- the issue cursor runs NS-1 chunks ahead of the multiply; everything in it is wave-uniform
    int tap = it_begin / cpt, cc = (it_begin % cpt) << 5;
    const float* i_src = p.src0;
    int i_cs = 0, i_coff = 0, i_dpix = 0, i_tap = 0;
    long long i_woff = 0;
    unsigned i_st = 0;
    auto begin_chunk = [&](int stage) {   // latch the position of the next chunk to issue, advance the cursor
        int dy, dx;
        tap_offset(p.tapmode, phase, tap, dy, dx);
        const bool first = cc < p.c0;
        i_src = first ? p.src0 : p.src1;
        i_cs = first ? p.c0 : p.c1;
        i_coff = first ? cc : cc - p.c0;
        i_dpix = dy * p.W + dx;
        i_woff = (long long)tap * p.cin + cc;
        i_st = lds_base + (unsigned)(stage * STAGE * 4);
        i_tap = tap;
        cc += 32;
        if (cc == p.cin) { cc = 0; ++tap; }
    };
    auto issue_piece = [&](int j) {       // j is a compile-time constant after unrolling
        if (j < A_PW) {
            const long long off = (long long)(a_pix[j < A_PW ? j : 0] + i_dpix) * i_cs + (i_coff + a_sw[j < A_PW ? j : 0]);
            const float* g = ((a_mask[j < A_PW ? j : 0] >> i_tap) & 1u) ? i_src + off : zero;
            lds_dma16(g, i_st + (unsigned)((wid_u * A_PW + j) * 1024));
        } else {
            const int jb = j - A_PW;
            const float* g = b_ok[jb >= 0 ? jb : 0] ? p.w + (b_off[jb >= 0 ? jb : 0] + i_woff) : zero;
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

    const bool ln_fold = p.ln_c1 != nullptr;
    float ln_s = 0.f, ln_q = 0.f;
    // prologue: chunks 0 .. NS-2 go out back to back
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < n_it) {
            begin_chunk(s);
#pragma unroll
            for (int j = 0; j < PW; ++j) issue_piece(j);
        }
    consume_epilogue_args(p);
    unsigned long long r0 = 0, t0 = 0;
    if (DBG) { r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime(); asm volatile("" ::"s"(r0), "s"(t0), "s"(r_entry)); }
    int stage = 0, istage = NS - 1;      // stage being multiplied, stage the next issued chunk goes to
    for (int k = 0; k < n_it; ++k) {
        // chunks k+1 .. min(k+NS-2, n_it-1) may still fly; in the last NS-2 iterations (nothing left to issue) wait for all
        if (k + NS - 2 < n_it) wait_vmcnt<(NS - 2) * PW>();
        else wait_vmcnt<0>();
        __syncthreads();
        const bool more = k + NS - 1 < n_it;     // wave-uniform
        // 2-stage ring: the next chunk has only this chunk's multiply to land in, so its pieces go out at once (2+
        // workgroups per CU cover the issue slots with each other's MFMAs).  Deeper rings have slack: their pieces are
        // spread over the four quarters below, in MFMA shadows.
        constexpr bool EARLY = NS == 2;
        if (EARLY && more) {
            begin_chunk(istage);
#pragma unroll
            for (int j = 0; j < PW; ++j) issue_piece(j);
        }
        const float* As = smem + stage * STAGE + a_base;
        const float* Bs = smem + stage * STAGE + b_base;
        float4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const float4*>(As + i * 1024 + foff[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const float4*>(Bs + j * 1024 + foff[0]);
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
            // e outermost: consecutive MFMAs go to different accumulators (round-robin over the TM*TN tiles).
            // The first MFMA of the quarter goes out alone; everything that is not an MFMA or a fragment read -- the
            // wave-uniform bookkeeping of the next chunk to issue (quarter 0), this quarter's share of its DMA pieces, the
            // LayerNorm row statistics -- is placed in its shadow, then the rest of the MFMAs follow.  (With that work in
            // front of the first MFMA, right after the barrier, a 64x64 tile spent 1600-1750 cycles per chunk for
            // 1024 cycles of MFMA: tools/conv_clock.py.)
            if (!EARLY) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0].x, b[cur][0].x, acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!EARLY && more) {
                if (q == 0) begin_chunk(istage);
#pragma unroll
                for (int j = 0; j < PW; ++j)
                    if ((j * 4) / PW == q) issue_piece(j);
            }
            if (q == 0 && ln_fold) {   // per-row sum / sum of squares of the staged A chunk (the swizzle only permutes a row's floats)
                constexpr int TPR = NW * 64 / BM, F4 = 8 / TPR;      // threads per row, float4 per thread
                const float4* rowp = reinterpret_cast<const float4*>(smem + stage * STAGE + (tid / TPR) * 32) + (tid % TPR) * F4;
#pragma unroll
                for (int i = 0; i < F4; ++i) {
                    const float4 v = rowp[i];
                    ln_s += (v.x + v.y) + (v.z + v.w);
                    ln_q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
            if (!EARLY) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (!EARLY && e == 0 && i == 0 && j == 0) continue;   // issued above
                        const float av = e == 0 ? a[cur][i].x : e == 1 ? a[cur][i].y : e == 2 ? a[cur][i].z : a[cur][i].w;
                        const float bv = e == 0 ? b[cur][j].x : e == 1 ? b[cur][j].y : e == 2 ? b[cur][j].z : b[cur][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
        istage = istage + 1 == NS ? 0 : istage + 1;
    }
    unsigned long long r1 = 0, t1 = 0;
    if (DBG) { r1 = __builtin_amdgcn_s_memrealtime(); t1 = __builtin_amdgcn_s_memtime(); }

    __syncthreads();   // every wave is done reading the ring: its LDS now stages the output block
    constexpr int EPI_FLOATS = TM * 32 * (TN * 32 + 8);
    float* rowstat = nullptr;
    if (ln_fold) {     // mean and 1/(std + eps) per pixel row (blocks.py:57-60: biased variance, eps added to the std)
        constexpr int TPR = NW * 64 / BM;
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) { ln_s += __shfl_xor(ln_s, o, 64); ln_q += __shfl_xor(ln_q, o, 64); }
        rowstat = smem + NW * EPI_FLOATS;            // behind the staging areas (the launch sizes the LDS for it)
        if (tid % TPR == 0) {
            const float inv_c = 1.0f / (float)p.cin;
            const float mean = ln_s * inv_c;
            const float var = fmaxf(ln_q * inv_c - mean * mean, 0.f);
            const float r = 1.0f / (sqrtf(var) + p.ln_eps);
            rowstat[2 * (tid / TPR)] = r;
            rowstat[2 * (tid / TPR) + 1] = r * mean;
        }
        __syncthreads();
    }
    store_block_via_lds<TM, TN>(p, acc, smem + wid * EPI_FLOATS, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, split, phase,
                                rowstat ? rowstat + 2 * wm * TM * 32 : nullptr);
    if (DBG && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int wg = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 2047;
        g_stamps[wg * 8 + 0] = t1 - t0;          // shader cycles in the k-loop
        g_stamps[wg * 8 + 1] = r1 - r0;          // 100 MHz ticks in the k-loop
        g_stamps[wg * 8 + 2] = (unsigned long long)n_it;
        g_stamps[wg * 8 + 3] = 1;
        g_stamps[wg * 8 + 4] = r_entry;          // absolute: kernel entry
        g_stamps[wg * 8 + 5] = r0;               // absolute: loop start
        g_stamps[wg * 8 + 6] = r1;               // absolute: loop end
        g_stamps[wg * 8 + 7] = __builtin_amdgcn_s_memrealtime();   // absolute: this wave's stores drained
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 conv with the INPUT HALO TILE staged once per channel chunk (the default for the UNet Block convs).
//
// Why: a CU sustains only ~6.8 B/clk of LDS-DMA traffic (1 KiB piece per ~150 cycles CU-wide, measured with
// tools/conv_clock.py loader stamps; consistent with a bounded number of outstanding L2 requests per CU).  The
// im2col kernels above re-fetch the activation tile for every tap: 32 KB per 4096 MFMA cycles for a 128x128 tile
// (7.8 B/clk), so the loaders, not the matrix pipe, set their pace.  Here a workgroup's 128 output pixels are
// whole image rows (128/W rows, or several whole images when H*W < 128); their (rows+2) x (W+2) halo of one
// 32-channel chunk (<= 224 pixels x 128 B) is DMA'd into LDS ONCE and all 9 taps read their A fragments from it at
// a wave-uniform row offset dy*(W+2)+dx.  Only the weights (16 KB per k-step) still stream: ~19 KB instead of 32 KB
// per 4096 MFMA cycles.  k order is (channel chunk, tap) so one halo serves 9 consecutive k-steps.
//
// Waves: 4 matrix waves (ds_read_b128 + MFMA only) + 4 loader waves (LDS-DMA only), one s_barrier per k-step.
//   weights: 4-stage ring.  Barrier s guarantees steps <= s+1 have landed, so a matrix wave prefetches the first
//            fragments of step s+1 while it multiplies step s -- no LDS latency is exposed after a barrier.
//   halo:    2 buffers; loader l issues piece j (8 halo rows) of chunk c+1 right after barrier 9c+j, j = 0..6;
//            they have landed by barrier 9c+8, i.e. before the first fragments of chunk c+1 are prefetched.
//   A loader tracks the size of its youngest issue group (4 weight pieces + 0/1 halo piece) and waits
//   s_waitcnt vmcnt(that size): everything older has landed.
// Epilogue: each matrix wave transposes its 64x64 accumulator block through LDS and writes rows as 256-byte
// dwordx4 segments (4x fewer store instructions than the per-register dword stores of the im2col kernels).
constexpr int HALO_PIECES_PER_LOADER = 7;
constexpr int HALO_MAX_PX = HALO_PIECES_PER_LOADER * 4 * 8;  // 224: 6x34 = 204 at W=32, 10x18 = 180 at W=16, 2x10x10 = 200 at W=8
constexpr int HALO_BN = 128;
constexpr int HALO_LDS_FLOATS = 2 * HALO_MAX_PX * 32 + 4 * HALO_BN * 32;   // 30720 floats = 120 KB

// STAMPS: s_memtime is an SMEM op (out-of-order lgkmcnt), so even a never-taken stamp branch forces lgkmcnt(0)
// waits into the loop -- the diagnostic build is a separate instantiation.
template <int STAMPS>   // 0: production, 1: timeline stamps outside the loop (DDK_DEBUG & 32), 2: per-step segments (64 / 128)
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const IgemmParams p) {
    constexpr int BM = 128, BN = HALO_BN, WN = 2;
    constexpr int NMW = 4, NLW = 4;
    constexpr int TM = 2, TN = 2;
    constexpr int B_PW = BN / 8 / NLW;                       // 4 weight pieces per loader per k-step
    constexpr int H_PW = HALO_PIECES_PER_LOADER;
    constexpr int HALO_FLOATS = HALO_MAX_PX * 32;
    constexpr int BST = BN * 32;                             // floats per weight stage
    constexpr int EPI_PITCH = 72;                            // 64 + 8: rows +4 apart land 32 banks apart
    static_assert(4 * 64 * EPI_PITCH <= HALO_LDS_FLOATS, "epilogue staging fits");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n, tile_z;
    {   // XCD-aware tile order (see igemm_dma_kernel)
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = logical % gridDim.y;
        const int rest = logical / gridDim.y;
        tile_m = rest % gridDim.x;
        tile_z = rest / gridDim.x;
    }
    unsigned long long r_entry = 0;
    if (STAMPS == 1) r_entry = __builtin_amdgcn_s_memrealtime();
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = tile_z;                                // nphase == 1
    // geometry of the pixel tile: TR = 128 / W whole rows of the row-major (b, y) list; TB images when TR > H
    const int W = p.W, H = p.H, Wp = W + 2;
    const int TR = BM / W;
    const int TB = TR > H ? TR / H : 1;                      // images per tile (H*W < 128)
    const int rows_img = TB > 1 ? H : TR;                    // output rows per image inside the tile
    const int img_px = (rows_img + 2) * Wp;                  // halo pixels per image
    const int halo_px = TB * img_px;
    const int R0 = m0 / W;                                   // first global row (b*H + y) of the tile
    const int cpt = p.cin >> 5;                              // channel chunks
    const int c_begin = split * p.kiters_per_split;          // for this kernel kiters_per_split counts CHUNKS
    const int n_chunks = min(cpt, c_begin + p.kiters_per_split) - c_begin;
    const int n_steps = n_chunks * 9;

    if (wid >= NMW) {
        // ================================================================ loader wave
        const int lw = wid - NMW;
        const int prow = lane >> 3, ppos = lane & 7;
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        if (!(p.debug & 256)) __builtin_amdgcn_s_setprio(3);   // loader instructions win issue arbitration against the MFMA stream
        // halo piece j of this loader = halo rows (lw*7 + j)*8 + prow; its source pixel never changes with the chunk
        int h_pix[H_PW], h_sw[H_PW];
        unsigned h_ok = 0, h_live = 0;
#pragma unroll
        for (int j = 0; j < H_PW; ++j) {
            const int piece = lw * H_PW + j;
            const int hp = piece * 8 + prow;
            h_sw[j] = (ppos ^ ((hp >> 1) & 7)) * 4;
            h_pix[j] = 0;
            if (piece * 8 < halo_px) h_live |= 1u << j;           // wave-uniform
            if (hp < halo_px) {
                const int img = (int)fdiv_u((unsigned)hp, p.dImg), rem = hp - img * img_px;
                const int hy = (int)fdiv_u((unsigned)rem, p.dWp), hx = rem - hy * Wp;
                const int row0 = R0 + img * rows_img;           // first output row of this image's part of the tile
                const int b = (int)fdiv_u((unsigned)row0, p.dHm);   // Hm == H for this kind
                const int y = row0 - b * H + hy - 1, x = hx - 1;
                if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && b < p.B) {
                    h_pix[j] = (b * H + y) * W + x;
                    h_ok |= 1u << j;
                }
            }
        }
        h_live = __builtin_amdgcn_readfirstlane(h_live);
        long long b_off[B_PW];
        bool b_ok[B_PW];
#pragma unroll
        for (int j = 0; j < B_PW; ++j) {
            const int r = (lw * B_PW + j) * 8 + prow;
            const int n = n0 + r;
            b_ok[j] = n < p.N;
            b_off[j] = ((long long)(b_ok[j] ? n : 0) * 9) * p.cin + (ppos ^ ((r >> 1) & 7)) * 4;
        }
        const float* zero = g_zero_page + ppos * 4;

        auto issue_halo_piece = [&](int chunk, int j, int buf) {   // piece j of channel chunk `chunk` into halo buffer buf
            const int cc = chunk << 5;
            const bool first = cc < p.c0;                          // wave-uniform
            const float* src = first ? p.src0 : p.src1;
            const int cs = first ? p.c0 : p.c1, coff = first ? cc : cc - p.c0;
            const float* g = ((h_ok >> j) & 1u) ? src + ((long long)h_pix[j] * cs + (coff + h_sw[j])) : zero;
            lds_dma16(g, lds_base + (unsigned)((buf * HALO_FLOATS + (lw * H_PW + j) * 256) * 4));
        };
        auto issue_b = [&](int chunk, int tap, int stage) {        // weights of k-step (chunk, tap)
            const long long woff = (long long)tap * p.cin + (chunk << 5);
#pragma unroll
            for (int j = 0; j < B_PW; ++j) {
                const float* g = b_ok[j] ? p.w + (b_off[j] + woff) : zero;
                lds_dma16(g, lds_base + (unsigned)((2 * HALO_FLOATS + stage * BST + (lw * B_PW + j) * 256) * 4));
            }
        };

        // prologue: the whole halo of the first chunk + weights of step 0, then the weights of steps 1 and 2
#pragma unroll
        for (int j = 0; j < H_PW; ++j)
            if ((h_live >> j) & 1u) issue_halo_piece(c_begin, j, 0);
        issue_b(c_begin, 0, 0);
        issue_b(c_begin, 1, 1);
        issue_b(c_begin, 2, 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * B_PW) : "memory");   // step 0 landed (1 and 2 may fly)
        __builtin_amdgcn_s_barrier();                                      // "pre" barrier: matrix waves prefetch step 0
        int wr = 3;
        int tap = 0, ci = 0;             // position of step s
        int tap3 = 3, ci3 = 0;           // position of step s + 3
        int young = B_PW;                // pieces in the youngest issue group (step s+2's group, issued during step s-1)
        unsigned long long l_vm = 0, l_bar = 0, l_iss = 0;  // DDK_DEBUG & 128: loader-side segment cycles
        for (int s = 0; s < n_steps; ++s) {
            unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (STAMPS == 2 && (p.debug & 128)) t0 = __builtin_amdgcn_s_memtime();
            // everything but the youngest group has landed => steps <= s+1 (and halo pieces issued before step s-1)
            if (s + 2 < n_steps) {
                if (young == B_PW + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_PW + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_PW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (STAMPS == 2 && (p.debug & 128)) t1 = __builtin_amdgcn_s_memtime();
            if (!(STAMPS == 1 && (p.debug & 2))) __builtin_amdgcn_s_barrier();
            if (STAMPS == 2 && (p.debug & 128)) t2 = __builtin_amdgcn_s_memtime();
            if (s + 3 < n_steps && !(STAMPS == 1 && (p.debug & 1))) {
                issue_b(c_begin + ci3, tap3, wr);
                wr = (wr + 1) & 3;
                young = B_PW;
                if (tap < H_PW && ci + 1 < n_chunks && ((h_live >> tap) & 1u)) {
#pragma unroll
                    for (int j = 0; j < H_PW; ++j)              // static piece index keeps h_pix/h_sw in registers
                        if (j == tap) issue_halo_piece(c_begin + ci + 1, j, (ci + 1) & 1);
                    young = B_PW + 1;
                }
            }
            if (++tap == 9) { tap = 0; ++ci; }
            if (++tap3 == 9) { tap3 = 0; ++ci3; }
            if (STAMPS == 2 && (p.debug & 128)) { t3 = __builtin_amdgcn_s_memtime(); l_vm += t1 - t0; l_bar += t2 - t1; l_iss += t3 - t2; }
        }
        __builtin_amdgcn_s_barrier();    // matrix waves reuse the LDS for their epilogue after this one
        if (STAMPS == 2 && (p.debug & 128) && lane == 0) {
            const int wg = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 511;
            g_stamps[2048 * 8 + (wg * 4 + lw) * 4 + 0] = l_vm;
            g_stamps[2048 * 8 + (wg * 4 + lw) * 4 + 1] = l_bar;
            g_stamps[2048 * 8 + (wg * 4 + lw) * 4 + 2] = l_iss;
            g_stamps[2048 * 8 + (wg * 4 + lw) * 4 + 3] = (unsigned long long)n_steps;
        }
        return;
    }

    // ==================================================================== matrix wave
    const int wm = wid / WN, wn = wid % WN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fh = lane >> 5;
    // A fragments: this lane's output pixel -> its halo row (centre tap), per MFMA tile i
    int a_row[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pm = (wm * TM + i) * 32 + (lane & 31);         // pixel inside the tile
        const int rt = (int)fdiv_u((unsigned)pm, p.dWm), x = pm - rt * W;   // Wm == W for this kind
        const int img = (int)fdiv_u((unsigned)rt, p.dRows), y = rt - img * rows_img;
        a_row[i] = img * img_px + (y + 1) * Wp + x + 1;
    }
    const int bsw = ((lane & 31) >> 1) & 7;
    int b_foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b_foff[q] = (lane & 31) * 32 + (((2 * q + fh) ^ bsw) << 2);
    const int b_base = 2 * HALO_FLOATS + wn * TN * 32 * 32;

    // LDS addressing in float4 units.  For one k-step (tap t, halo buffer hb, ring stage st) this lane reads
    //   A quarter q of MFMA tile i at  hb*HALO4 + hr_i*8 + ((2q+fh) ^ sw_i),  hr_i = a_row[i] + tap offset, sw_i = (hr_i>>1)&7
    //   B quarter q                at  b_base4 + st*BST4 + b_f4[q]  (+ 256 per n-tile)
    // The offsets of quarter q are recomputed for step s+1 right after quarter q of step s has been requested, a few VALU
    // ops at a time in the shadow of an MFMA (explicit sched_barrier fences keep them there).
    constexpr int HALO4 = HALO_FLOATS / 4, BST4 = BST / 4;
    const float4* smem4 = reinterpret_cast<const float4*>(smem);
    const int b_base4 = b_base / 4;
    int b_f4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b_f4[q] = (lane & 31) * 8 + ((2 * q + fh) ^ bsw);
    int cA[4][TM], cB[4];            // offsets of the quarter that will be requested next time around
    int n_base[TM], n_sw[TM], n_soff = 0;   // step s+1: per-tile base / swizzle key, stage offset
    auto next_bases = [&](int t, int hb, int st) {
        const int ty = (t * 11) >> 5;
        const int toff = (ty - 1) * Wp + (t - 3 * ty - 1);        // wave-uniform halo-row offset of this tap
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int hr = a_row[i] + toff;
            n_base[i] = hb * HALO4 + hr * 8;
            n_sw[i] = (hr >> 1) & 7;
        }
        n_soff = b_base4 + st * BST4;
    };
    auto next_quarter = [&](int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) cA[q][i] = n_base[i] + ((2 * q + fh) ^ n_sw[i]);
        cB[q] = n_soff + b_f4[q];
    };
    float4 a[2][TM], b[2][TN];
    auto load_frags = [&](int slot, int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[slot][i] = smem4[cA[q][i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[slot][j] = smem4[cB[q] + j * 256];
    };
    auto mfma_e = [&](int cur, int e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float av = e == 0 ? a[cur][i].x : e == 1 ? a[cur][i].y : e == 2 ? a[cur][i].z : a[cur][i].w;
                const float bv = e == 0 ? b[cur][j].x : e == 1 ? b[cur][j].y : e == 2 ? b[cur][j].z : b[cur][j].w;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
    };

    int stage = 0, tap = 0, hbuf = 0;
    __builtin_amdgcn_s_barrier();                                  // "pre" barrier: step 0 landed
    next_bases(0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) next_quarter(q);
    load_frags(0, 0);
    unsigned long long r_loop0 = 0, t_loop0 = 0;
    if (STAMPS == 1) {
        r_loop0 = __builtin_amdgcn_s_memrealtime();
        t_loop0 = __builtin_amdgcn_s_memtime();
        asm volatile("" ::"s"(r_loop0), "s"(t_loop0), "s"(r_entry));
    }
    consume_epilogue_args(p);
    unsigned long long seg_wait = 0, seg_mma = 0;   // DDK_DEBUG & 64
    for (int s = 0; s < n_steps; ++s) {
        unsigned long long t0 = 0, t1 = 0;
        if (STAMPS == 2 && (p.debug & 64)) t0 = __builtin_amdgcn_s_memtime();
        if (!(STAMPS == 1 && (p.debug & 2))) __builtin_amdgcn_s_barrier();   // steps <= s+1 landed; stage (s+3)&3 is free for the loaders
        __builtin_amdgcn_sched_barrier(0);
        if (STAMPS == 2 && (p.debug & 64)) t1 = __builtin_amdgcn_s_memtime();
        int stage_n = 0, tap_n = 0, hbuf_n = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            // one ds_read_b128 behind each of the first four MFMAs (a clump of four reads holds the wave's issue longer than
            // one MFMA runs); q == 3 requests quarter 0 of step s+1 (cA[0]/cB[0] already hold its offsets)
            {
                const int qn = (q + 1) & 3;
                const float av = a[cur][0].x, av1 = a[cur][1].x, bv = b[cur][0].x, bv1 = b[cur][1].x;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(STAMPS == 1 && (p.debug & 8))) a[nxt][0] = smem4[cA[qn][0]];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv1, acc[0][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(STAMPS == 1 && (p.debug & 8))) a[nxt][1] = smem4[cA[qn][1]];
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv, acc[1][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(STAMPS == 1 && (p.debug & 8))) b[nxt][0] = smem4[cB[qn]];
                __builtin_amdgcn_sched_barrier(0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(STAMPS == 1 && (p.debug & 8))) b[nxt][1] = smem4[cB[qn] + 256];
                __builtin_amdgcn_sched_barrier(0);
            }
            static_assert(TM == 2 && TN == 2, "hand-interleaved for a 64x64 wave tile");
            if (q == 0) {
                // position of step s+1 (the last step prefetches its own fragments again: unused, but keeps the loop branch-free)
                const bool more = s + 1 < n_steps;
                stage_n = more ? (stage + 1) & 3 : stage;
                tap_n = more ? tap + 1 : tap;
                hbuf_n = hbuf;
                if (tap_n == 9) { tap_n = 0; hbuf_n ^= 1; }
                stage_n = __builtin_amdgcn_readfirstlane(stage_n);
                tap_n = __builtin_amdgcn_readfirstlane(tap_n);
                hbuf_n = __builtin_amdgcn_readfirstlane(hbuf_n);
                next_bases(tap_n, hbuf_n, stage_n);
                next_quarter(0);                   // quarter 0 of step s was requested in step s-1
            } else {
                next_quarter(q);                   // quarter q of step s was requested in block q-1
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_e(cur, 1);
            mfma_e(cur, 2);
            mfma_e(cur, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage_n; tap = tap_n; hbuf = hbuf_n;
        if (STAMPS == 2 && (p.debug & 64)) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); seg_wait += t1 - t0; seg_mma += t2 - t1; }
    }
    if (STAMPS == 2 && (p.debug & 64) && lane == 0) {
        const int wg = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 511;
        g_stamps[2048 * 8 + (wg * 4 + (wid & 3)) * 4 + 0] = seg_wait;
        g_stamps[2048 * 8 + (wg * 4 + (wid & 3)) * 4 + 1] = 0;
        g_stamps[2048 * 8 + (wg * 4 + (wid & 3)) * 4 + 2] = seg_mma;
        g_stamps[2048 * 8 + (wg * 4 + (wid & 3)) * 4 + 3] = (unsigned long long)n_steps;
    }
    unsigned long long r_loop1 = 0, t_loop1 = 0;
    if (STAMPS == 1) { r_loop1 = __builtin_amdgcn_s_memrealtime(); t_loop1 = __builtin_amdgcn_s_memtime(); }
    __builtin_amdgcn_s_barrier();        // every wave is done reading the ring: the LDS is free for the epilogue

    // ---- epilogue: 64x64 block of this wave -> LDS -> float4 rows -> dwordx4 stores
    store_block_via_lds<TM, TN>(p, acc, smem + wid * 64 * EPI_PITCH, lane, m0 + wm * 64, n0 + wn * 64, split, 0);
    if (STAMPS == 1 && wid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int wg = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 2047;
            g_stamps[wg * 8 + 0] = t_loop1 - t_loop0;   // shader cycles in the k-loop
            g_stamps[wg * 8 + 1] = r_loop1 - r_loop0;   // 100 MHz ticks in the k-loop
            g_stamps[wg * 8 + 2] = (unsigned long long)n_steps;
            g_stamps[wg * 8 + 3] = 1;
            g_stamps[wg * 8 + 4] = r_entry;
            g_stamps[wg * 8 + 5] = r_loop0;
            g_stamps[wg * 8 + 6] = r_loop1;
            g_stamps[wg * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

#include "conv_c32_kernel.inc"

// out[i] = sum_s slab[s][i] (fixed order) + bias[i % N] + resid[i]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits,
                                                            long long slab_stride, const float* __restrict__ bias,
                                                            const float* __restrict__ resid, float* __restrict__ out,
                                                            long long n4, int N, int post_mish, float* __restrict__ mish_out,
                                                            const float* __restrict__ dmish_src) {
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
        if (dmish_src) {
            const float4 ss = reinterpret_cast<const float4*>(dmish_src)[i];
            v.x *= mish_grad_f(ss.x); v.y *= mish_grad_f(ss.y); v.z *= mish_grad_f(ss.z); v.w *= mish_grad_f(ss.w);
        }
        if (resid) {
            const float4 r = reinterpret_cast<const float4*>(resid)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (post_mish) { v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w); }
        reinterpret_cast<float4*>(out)[i] = v;
        if (mish_out) reinterpret_cast<float4*>(mish_out)[i] = make_float4(mish_f(v.x), mish_f(v.y), mish_f(v.z), mish_f(v.w));
    }
}

// ------------------------------------------------------------------------------------------------
enum TileId { T128x128 = 0, T128x64 = 1, T64x64 = 2, T128x32 = 3, T64x32 = 4 };

struct Choice {
    TileId tile;
    int splits;
    int kps;     // k-chunks per split
    int stages;  // LDS ring depth of the im2col kernel (ignored by the halo kernel)
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

// Tuning knobs (environment variables) exist only in the -DDDK_TUNING build (libddk_tune.so, `make tune`) that
// tools/conv_bench.py and friends load; the production library has no getenv in its launch path.
#ifdef DDK_TUNING
static bool tuning_force(int& tile, int& splits, int& stages) {
    const char* f = getenv("DDK_FORCE_TILE");   // "<tile id>[,<splits>[,<stages>]]"
    if (!f) return false;
    tile = 0; splits = 1; stages = 0;
    return sscanf(f, "%d,%d,%d", &tile, &splits, &stages) >= 1 && tile >= 0 && tile <= 4;
}
static bool tuning_flag(const char* name) { return getenv(name) != nullptr; }
static int tuning_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static bool tuning_force(int&, int&, int&) { return false; }
static bool tuning_flag(const char*) { return false; }
static int tuning_int(const char*, int dflt) { return dflt; }
#endif

// Ring depth for a tile and grid.  Measured (profiles/r02_conv_sweep.txt, tools/conv_clock.py): a deeper ring only pays
// when a CU holds ONE workgroup (<= 256 workgroups: the 4x4 maps, -5 % per launch); with 2+ workgroups per CU they cover each
// other's DMA latency and the extra LDS only costs occupancy (the 1x1 projections at 32x32 lost 20 % at 64 KB per
// workgroup), so everything else keeps the 2-stage ring.  Instantiated: {2, 4, 6} (64x64), {2, 4} (128x64), {2} (others).
static int choose_stages(TileId t, long long workgroups, int kps) {
    const bool one_per_cu = workgroups <= 256;
    if (t == T64x64 && one_per_cu) return kps >= 5 ? 6 : (kps >= 3 ? 4 : 2);
    if (t == T128x64 && one_per_cu && kps >= 3) return 4;
    return 2;
}

// Pick the largest tile that still gives the 256 CUs about a full wave of workgroups; when even the
// smallest does not, split k across workgroups (at least 4 k-chunks per split).
static Choice choose_tile(long long M, int N, int nphase, int kiters) {
    auto n_tiles = [&](TileId t) {
        int bm, bn;
        tile_dims(t, bm, bn);
        return ceil_div(M, bm) * ceil_div(N, bn) * nphase;
    };
    auto with_splits = [&](TileId t, long long s, int stages = 0) {
        Choice c{t, 1, kiters, 0};
        if (s < 1) s = 1;
        if (s > kiters) s = kiters;
        c.kps = (int)ceil_div(kiters, s);
        c.splits = (int)ceil_div(kiters, c.kps);
        c.stages = stages > 0 ? stages : choose_stages(t, n_tiles(t) * c.splits, c.kps);
        return c;
    };
    {
        int t, s, st;
        if (tuning_force(t, s, st)) {
            int bm, bn;
            tile_dims((TileId)t, bm, bn);
            if (N % 32 == 0 && (bn <= N || bn == 32)) return with_splits((TileId)t, s, st);
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
    // Rules fitted to tools/conv_bench.py sweeps on MI355X (profiles/r02_conv_sweep.txt):
    // 1. short contractions (1x1 convs, <= 8 k-chunks): smallest tile, no split -- prologue/epilogue bound;
    //    (32-wide outputs on many pixels -- the dDDPM encoder / decoder 1x1 convs at up to 262144 pixels -- take the 4-wave 128x32
    //    tile instead of the 2-wave 64x32 one: 15.4 -> 10.6 us at 131072 pixels, profiles/r03_conv_clock_sweep.txt)
    if (kiters <= 8 && N % 64 != 0 && n_tiles(T128x32) >= 512) return with_splits(T128x32, 1);
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
    // 3b. a small grid with a long contraction (the 1x1 skip convs 512 -> 256 of the 4x4 / 8x8 maps: 32 / 128 tiles, 16 chunks):
    //     split k up to ~256 workgroups while every workgroup keeps >= 4 chunks (16.3 -> 10.3 us at 4x4, 16.6 -> 14.8 at 8x8
    //     including the reduce; profiles/r02_conv_sweep.txt);
    if (kiters >= 16 && kiters < 36 && n_tiles(order[n_order - 1]) <= 128) {
        long long s = 1;
        while (n_tiles(order[n_order - 1]) * s * 2 <= 256 && kiters / (s * 2) >= 4) s *= 2;
        if (s > 1) return with_splits(order[n_order - 1], s);
    }
    // 4. else the smallest tile with the deepest split that keeps >= 9 chunks per workgroup.
    long long s = 16;
    while (s > 1 && (kiters / s < 9 || n_tiles(order[n_order - 1]) * s > 1024)) s /= 2;
    return with_splits(order[n_order - 1], s);
}

// ---- halo kernel eligibility and split choice.  Tiles are whole image rows, so W must divide 128, a tile must not
// straddle images partially, and the (rows+2) x (W+2) halo must fit HALO_MAX_PX.  One workgroup per CU (120 KB of
// LDS): the channel chunks are split until >= 208 workgroups exist, and the kernel is only used when that leaves
// >= 4 chunks (36 k-steps) per workgroup -- with less, its 3 us prologue + 3 us epilogue lose to the im2col kernel's
// smaller tiles (tools/conv_bench.py sweeps).
static bool choose_halo(int kind, int B, int H, int W, int cin, int N, Choice& c) {
    const bool off = tuning_flag("DDK_NO_HALO");
    const int min_chunks = tuning_int("DDK_HALO_MIN_CHUNKS", 4);
    if (off || kind != DDK_CONV3X3_S1 || N < 128 || N % 32) return false;
    if (W < 8 || W > 128 || 128 % W) return false;
    const int TR = 128 / W;
    if (TR <= H ? (H % TR != 0) : (TR % H != 0)) return false;
    const int TB = TR > H ? TR / H : 1, rows_img = TB > 1 ? H : TR;
    if (TB * (rows_img + 2) * (W + 2) > HALO_MAX_PX) return false;
    const int chunks = cin / 32;
    const long long tiles = ceil_div((long long)B * H * W, 128) * ceil_div(N, 128);
    long long s = 1;
    int ft, fs, fst;
    if (tuning_force(ft, fs, fst)) {
        s = fs < 1 ? 1 : fs;
        if (s > chunks) s = chunks;
    } else {
        while (tiles * s < 208 && s < chunks) s *= 2;
        if (tiles * s < 208 || chunks / s < min_chunks) return false;
    }
    c = Choice{T128x128, 1, chunks, 0};
    c.kps = (int)ceil_div(chunks, s);
    c.splits = (int)ceil_div(chunks, c.kps);
    return true;
}

#ifdef DDK_TUNING
#define DDK_HALO_VARIANTS(X) X(0) X(1) X(2)
#else
#define DDK_HALO_VARIANTS(X) X(0)
#endif

// ---- register-resident 32 -> 32 kernel (conv_c32_kernel.inc): the halo kernel's tile geometry, whole tiles only, and enough tiles
// to fill half the chip (below 128 tiles its once-per-workgroup filter load is not repaid: the im2col tile kernel stays)
static bool choose_c32(int kind, int B, int H, int W, int cin, int N, int c1, bool pre_mish) {
    if (tuning_flag("DDK_NO_C32") || kind != DDK_CONV3X3_S1 || cin != 32 || N != 32 || c1 != 0 || pre_mish) return false;
    if (W != 8 && W != 16 && W != 32 && W != 64) return false;
    const int TR = 128 / W;
    if (TR <= H ? (H % TR != 0) : (TR % H != 0)) return false;
    if (c32_halo_px(W, H) > C32_MAX_PX) return false;
    const long long M = (long long)B * H * W;
    if (M % 128 || M * 32 * 4 >= (1LL << 31)) return false;
    return M / 128 >= tuning_int("DDK_C32_MIN_TILES", 128);   // (16x16 x 64 images = 128 tiles: 10.4 / 12.0 / 12.2 -> 9.0 / 10.2 / 10.4 us)
}
#define DDK_C32_WIDTHS(X) X(8) X(16) X(32) X(64)
static int launch_c32(const IgemmParams& p, hipStream_t st) {
    const int tiles = p.M / 128;
    const int cap = tuning_int("DDK_C32_GRID", 512);
    const dim3 grid((unsigned)(tiles < cap ? tiles : cap));
    const size_t lds = C32_LDS_BYTES;
    switch (p.W) {
#define C32_CASE(WW)                                                                                         \
    case WW:                                                                                                 \
        if (p.dmish_src) hipLaunchKernelGGL((conv3x3_c32_kernel<WW, true>), grid, dim3(256), lds, st, p);    \
        else hipLaunchKernelGGL((conv3x3_c32_kernel<WW, false>), grid, dim3(256), lds, st, p);               \
        break;
        DDK_C32_WIDTHS(C32_CASE)
#undef C32_CASE
        default: DDK_REQUIRE(false, "conv(c32): width");
    }
    return check_launch("conv3x3_c32_kernel");
}

static int launch_halo(const IgemmParams& p, hipStream_t st) {
    constexpr size_t lds = (size_t)HALO_LDS_FLOATS * sizeof(float);
    dim3 grid((unsigned)ceil_div(p.M, 128), (unsigned)ceil_div(p.N, HALO_BN), (unsigned)p.splits);
#ifdef DDK_TUNING
    if (p.debug & (64 | 128)) hipLaunchKernelGGL(conv3x3_halo_kernel<2>, grid, dim3(512), lds, st, p);
    else if (p.debug & 32) hipLaunchKernelGGL(conv3x3_halo_kernel<1>, grid, dim3(512), lds, st, p);
    else
#endif
    hipLaunchKernelGGL(conv3x3_halo_kernel<0>, grid, dim3(512), lds, st, p);
    return check_launch("conv3x3_halo_kernel");
}

struct ConvPlan {
    bool halo;
    Choice c;
};
static ConvPlan plan_conv(int kind, int B, int H, int W, int cin, int N, const Geometry& g, bool pre_mish = false) {
    Choice c;
    if (!pre_mish && choose_halo(kind, B, H, W, cin, N, c)) return {true, c};
    return {false, choose_tile((long long)B * g.Hm * g.Wm, N, g.nphase, g.ntaps * (cin / 32))};
}

// Kernels that need more dynamic LDS than the default 64 KB limit must be told so once per device, outside any stream
// capture: ensure_device_init() (core.hip) runs this the first time a device is used (ddk_unet_create, every conv entry).
template <typename K>
static int allow_lds(K kernel, size_t bytes) {
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DDK_OK;
}
template <int BM, int BN, int WM, int WN, int NS>
constexpr size_t dma_lds_bytes() {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr size_t ring = (size_t)NS * (BM + BN) * 32 * sizeof(float);
    constexpr size_t epi = ((size_t)WM * WN * TM * 32 * (TN * 32 + 8) + 2 * BM) * sizeof(float);   // staging + folded-LN row stats
    return ring > epi ? ring : epi;
}
constexpr size_t reg_lds_bytes(int bm, int bn) { return 2 * (size_t)(bm + bn) * LDK * sizeof(float); }

template <int BM, int BN, int WM, int WN, int NS>
static int launch_dma(const IgemmParams& p, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(p.M, BM), (unsigned)ceil_div(p.N, BN), (unsigned)(p.nphase * p.splits));
#ifdef DDK_TUNING
    if (p.debug & 32) {
        static bool once = false;   // tuning build only
        if (!once) { DDK_TRY(allow_lds(&igemm_dma_kernel<BM, BN, WM, WN, NS, 1>, dma_lds_bytes<BM, BN, WM, WN, NS>())); once = true; }
        hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, NS, 1>), grid, dim3(WM * WN * 64), (dma_lds_bytes<BM, BN, WM, WN, NS>()), st, p);
        return check_launch("igemm_dma_kernel<dbg>");
    }
#endif
    hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, NS>), grid, dim3(WM * WN * 64), (dma_lds_bytes<BM, BN, WM, WN, NS>()), st, p);
    return check_launch("igemm_dma_kernel");
}
template <int BM, int BN, int WM, int WN>
static int launch_reg(const IgemmParams& p, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(p.M, BM), (unsigned)ceil_div(p.N, BN), (unsigned)(p.nphase * p.splits));
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), reg_lds_bytes(BM, BN), st, p);
    return check_launch("igemm_kernel");
}

static int launch_tile(const IgemmParams& p, const Choice& c, hipStream_t st) {
    const bool reg = p.pre_mish || tuning_flag("DDK_NO_DMA");   // Mish-on-load needs the register-staged kernel
    const int ns = c.stages;
    switch (c.tile) {
        case T128x128: return reg ? launch_reg<128, 128, 2, 2>(p, st) : launch_dma<128, 128, 2, 2, 2>(p, st);
        case T128x64:
            if (reg) return launch_reg<128, 64, 2, 2>(p, st);
            return ns >= 4 ? launch_dma<128, 64, 2, 2, 4>(p, st) : launch_dma<128, 64, 2, 2, 2>(p, st);
        case T64x64:
            if (reg) return launch_reg<64, 64, 2, 2>(p, st);
            return ns >= 6 ? launch_dma<64, 64, 2, 2, 6>(p, st) : ns >= 4 ? launch_dma<64, 64, 2, 2, 4>(p, st) : launch_dma<64, 64, 2, 2, 2>(p, st);
        case T128x32: return reg ? launch_reg<128, 32, 4, 1>(p, st) : launch_dma<128, 32, 4, 1, 2>(p, st);
        default: return reg ? launch_reg<64, 32, 2, 1>(p, st) : launch_dma<64, 32, 2, 1, 2>(p, st);
    }
}

int conv_init_device() {
#define HALO_ATTR(V) DDK_TRY(allow_lds(&conv3x3_halo_kernel<V>, (size_t)HALO_LDS_FLOATS * sizeof(float)));
    DDK_HALO_VARIANTS(HALO_ATTR)
#undef HALO_ATTR
#define C32_ATTR(WW)                                                                                      \
    DDK_TRY(allow_lds(&conv3x3_c32_kernel<WW, false>, (size_t)C32_LDS_BYTES));          \
    DDK_TRY(allow_lds(&conv3x3_c32_kernel<WW, true>, (size_t)C32_LDS_BYTES));
    DDK_C32_WIDTHS(C32_ATTR)
#undef C32_ATTR
#define DMA_ATTR(BM, BN, WM, WN, NS) DDK_TRY(allow_lds(&igemm_dma_kernel<BM, BN, WM, WN, NS>, dma_lds_bytes<BM, BN, WM, WN, NS>()));
    DMA_ATTR(128, 128, 2, 2, 2)
    DMA_ATTR(128, 64, 2, 2, 2) DMA_ATTR(128, 64, 2, 2, 4)
    DMA_ATTR(64, 64, 2, 2, 2) DMA_ATTR(64, 64, 2, 2, 4) DMA_ATTR(64, 64, 2, 2, 6)
    DMA_ATTR(128, 32, 4, 1, 2)
    DMA_ATTR(64, 32, 2, 1, 2)
#undef DMA_ATTR
    DDK_TRY(allow_lds(&igemm_kernel<128, 128, 2, 2>, reg_lds_bytes(128, 128)));
    DDK_TRY(allow_lds(&igemm_kernel<128, 64, 2, 2>, reg_lds_bytes(128, 64)));
    DDK_TRY(allow_lds(&igemm_kernel<64, 64, 2, 2>, reg_lds_bytes(64, 64)));
    DDK_TRY(allow_lds(&igemm_kernel<128, 32, 4, 1>, reg_lds_bytes(128, 32)));
    DDK_TRY(allow_lds(&igemm_kernel<64, 32, 2, 1>, reg_lds_bytes(64, 32)));
    return DDK_OK;
}

size_t conv_workspace_bytes(int kind, int B, int H, int W, int cin, int N) {
    Geometry g;
    if (!conv_geometry(kind, H, W, g) || cin <= 0 || cin % 32) return 0;
    const Choice c = plan_conv(kind, B, H, W, cin, N, g).c;
    size_t need = c.splits == 1 ? 0 : (size_t)c.splits * B * g.Ho * g.Wo * N * sizeof(float);
    if (kind == DDK_CONVT4X4_S2 && convT_wino_ok(H, W, cin, N)) {        // the Winograd form may split differently: room for either
        const int s = convT_wino_splits(B, H, W, cin, N);
        const size_t w = s == 1 ? 0 : (size_t)s * B * g.Ho * g.Wo * N * sizeof(float);
        if (w > need) need = w;
    }
    return need;
}

bool conv_ln_fold_ok(int B, int H, int W, int cin, int N) {
    const bool off = tuning_flag("DDK_NO_LN_FOLD");
    Geometry g;
    if (off || !conv_geometry(DDK_CONV1X1, H, W, g) || cin <= 0 || cin % 32 || N % 32) return false;
    const Choice c = choose_tile((long long)B * H * W, N, 1, cin / 32);
    return c.splits == 1 && (c.tile == T64x64 || c.tile == T128x64 || c.tile == T128x128);
}

int conv_forward(const ddk_conv_args& a, hipStream_t st, const ConvLnFold* ln, const WinoGnFuse* fuse) {
    Geometry g;
    DDK_REQUIRE(conv_geometry(a.kind, a.H, a.W, g), "conv kind");
    DDK_REQUIRE(a.src0 && a.out, "conv: null src0/out");
    DDK_REQUIRE(a.weight || a.weight_wino, "conv: null weight");     // the Winograd copy alone will do where that path is taken (checked below)
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

#ifdef DDK_HOST_SANITIZE
    {   // base + extent of every tensor the conv will touch (host_sanitize.h)
        const long long pix_in = (long long)a.B * a.H * a.W, pix_out = (long long)a.B * g.Ho * g.Wo;
        san::extent("conv src0", a.src0, pix_in * a.c0 * 4);
        san::extent("conv src1", a.src1, pix_in * a.c1 * 4);
        if (!a.defer_reduce) san::extent("conv out", a.out, pix_out * a.N * 4);
        san::extent("conv resid", a.resid, pix_out * a.N * 4);
        san::extent("conv bias", a.bias, (long long)a.N * 4);
        san::extent("conv workspace", a.workspace, (long long)a.workspace_bytes);
        san::extent("conv mish_out", a.mish_out, pix_out * a.N * 4);
        san::extent("conv dmish_src", a.dmish_src, pix_out * a.N * 4);
        if (a.gn_partials && a.gn_groups > 0) san::extent("conv gn_partials", a.gn_partials, (pix_out / 128) * a.gn_groups * 8);
    }
#endif
    DDK_REQUIRE(aligned16(a.mish_out) && aligned16(a.dmish_src), "conv: mish_out / dmish_src alignment");
    DDK_REQUIRE(!(a.mish_out || a.dmish_src) || !(a.defer_reduce || a.gn_partials || fuse || ln),
                "conv: mish_out / dmish_src go with a plain conv (no deferred reduce, GroupNorm partials or LayerNorm folding)");
    const bool act_epilogue = a.mish_out || a.dmish_src;     // only the im2col / halo epilogues and the slab reduce implement these
    if (a.weight_wino && !a.pre_mish && !act_epilogue && conv_wino_ok(a.kind, a.H, a.W, a.c0 + a.c1, a.N)) {
        // Winograd F(2x2, 3x3) path (conv_wino.hip): same slab / reduce conventions as the direct kernels
        DDK_REQUIRE(aligned16(a.weight_wino), "conv: weight_wino alignment");
        DDK_TRY(ensure_device_init());
        const int ws = conv_wino_splits(a.B, a.H, a.W, a.c0 + a.c1, a.N);
        const long long slab = (long long)a.B * a.H * a.W * a.N;
        if (ws > 1) {
            const size_t need = (size_t)(fuse ? ws - 1 : ws) * slab * sizeof(float);    // fuse: only the partners' partial tiles
            if (!a.workspace || a.workspace_bytes < need) {
                set_error("conv(wino): split-K workspace too small (%zu < %zu)", a.workspace_bytes, need);
                return DDK_ERR_WORKSPACE;
            }
            DDK_REQUIRE(aligned16(a.workspace), "conv: workspace alignment");
        }
        DDK_REQUIRE((long long)a.B * a.H * a.W * (a.c0 > a.c1 ? a.c0 : a.c1) * 4 < (1LL << 31), "conv(wino): a source of 2 GiB or more");
        DDK_TRY(conv_wino_forward(a, ws, st, fuse));
        if (ws > 1 && !a.defer_reduce && !fuse) {      // (fuse: the tiles' first workgroups have summed their partners in the launch)
            const long long n4 = slab / 4;
            const int blocks = (int)(ceil_div(n4, 256) < 2048 ? ceil_div(n4, 256) : 2048);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a.workspace), ws, slab, a.bias,
                               a.resid, a.out, n4, a.N, a.post_mish, static_cast<float*>(nullptr), static_cast<const float*>(nullptr));
            DDK_TRY(check_launch("splitk_reduce_kernel"));
        }
        return DDK_OK;
    }
    if (a.kind == DDK_CONVT4X4_S2 && a.weight_wino && a.c1 == 0 && !a.pre_mish && !a.post_mish && !a.resid && !act_epilogue && !a.gn_partials &&
        !fuse && !ln && convT_wino_ok(a.H, a.W, a.c0, a.N)) {
        // transpose conv as Winograd F(2x2, 2x2) per output phase (conv_winoT_kernel.inc): 9/16 of the direct multiplies
        DDK_REQUIRE(aligned16(a.weight_wino), "conv: weight_wino alignment");
        DDK_TRY(ensure_device_init());
        const int ws = convT_wino_splits(a.B, a.H, a.W, a.c0, a.N);
        const long long slab = (long long)a.B * g.Ho * g.Wo * a.N;
        if (ws > 1) {
            const size_t need = (size_t)(fuse ? ws - 1 : ws) * slab * sizeof(float);    // fuse: only the partners' partial tiles
            if (!a.workspace || a.workspace_bytes < need) {
                set_error("conv(winoT): split-K workspace too small (%zu < %zu)", a.workspace_bytes, need);
                return DDK_ERR_WORKSPACE;
            }
            DDK_REQUIRE(aligned16(a.workspace), "conv: workspace alignment");
        }
        DDK_REQUIRE((long long)a.B * a.H * a.W * a.c0 * 4 < (1LL << 31), "conv(winoT): a source of 2 GiB or more");
        DDK_TRY(convT_wino_forward(a, ws, st));
        if (ws > 1 && !a.defer_reduce && !fuse) {      // (fuse: the tiles' first workgroups have summed their partners in the launch)
            const long long n4 = slab / 4;
            const int blocks = (int)(ceil_div(n4, 256) < 2048 ? ceil_div(n4, 256) : 2048);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a.workspace), ws, slab, a.bias,
                               a.resid, a.out, n4, a.N, a.post_mish, static_cast<float*>(nullptr), static_cast<const float*>(nullptr));
            DDK_TRY(check_launch("splitk_reduce_kernel"));
        }
        return DDK_OK;
    }
    DDK_REQUIRE(a.weight, "conv: weight is null and the shape / arguments do not take the Winograd path (ddk_conv_wino_splits() == 0, "
                          "pre_mish, mish_out or dmish_src)");
    // (the LayerNorm-folded form only where every workgroup walks >= 16 tiles: its ragged last round and per-tile statistics pass
    //  lose to the tile kernel at cfg4's 6 tiles per workgroup -- 44.6 vs 42.2 us -- and win on the full-resolution maps: 128x128
    //  x 8 images 264 -> ~140 us; or where the tiles divide evenly over the workgroups, e.g. the 256-channel k, v projection of the
    //  folded attention block: 512 tiles over 2 x 128 workgroups, 30.2 -> ~26 us)
    if (a.kind == DDK_CONV1X1 && a.c1 == 0 && !a.pre_mish && !a.post_mish && !a.defer_reduce && !a.gn_partials && !fuse && !act_epilogue &&
        !tuning_flag("DDK_NO_CONV1X1_WS") && conv1x1_ws_ok((long long)a.B * a.H * a.W, a.c0, a.N) &&
        ((long long)a.B * a.H * a.W / 64) * (a.N / 128) >= 256 &&
        (!ln || ((long long)a.B * a.H * a.W / 64) * (a.N / 128) >= 16 * 256 ||
         (256 % (a.N / 128) == 0 && ((long long)a.B * a.H * a.W / 64) % (256 / (a.N / 128)) == 0))) {
        // 128 input channels on a large map: weights-stationary streaming kernel (conv1x1_ws.hip).  Measured in a cfg4 step
        // (profiles/r03_sampler_step_breakdown.txt): to_out 17.2 -> 15.4 us at 32x32, 11.0 -> 9.3 at 16x16 (256 out), res_conv
        // 9.7 -> 9.4; with fewer than 256 (tile, slice) pairs the chip is half empty and the tile kernel wins (9.0 vs 10.0 us).
        // The LayerNorm-folded to_qkv stays on the tile kernel below: the streaming kernel's per-tile statistics pass and its
        // ragged 512-tiles-over-85-workgroups split cost more than the tile kernel's weight re-reads (44.6 vs 42.2 us at 32x32).
        return conv1x1_ws(a.src0, a.weight, a.bias, a.resid, a.out, (long long)a.B * a.H * a.W, a.N, ln, st);
    }
    if (a.kind == DDK_CONV1X1 && !a.pre_mish && !a.post_mish && !a.defer_reduce && !a.gn_partials && !fuse && !act_epilogue &&
        !tuning_flag("DDK_NO_CONV1X1_SM") && conv1x1_sm_ok((long long)a.B * a.H * a.W, a.c0, a.c1, a.N) &&
        (!ln || (a.c1 == 0 && ((long long)a.B * a.H * a.W / 32) * (a.N / 32) <= 512)))   // (LayerNorm-folded at 768 tiles: 11.9 vs 13.5 / 11.7 vs 11.5 us: a draw)
        // 1x1 conv on a small map (to_out, res_conv, to_qkv with the LayerNorm folded, at 4x4 / 8x8): 32x32 tiles, the waves split K,
        // one barrier (conv1x1_sm.hip)
        return conv1x1_sm(a.src0, a.c0, a.src1, a.c1, a.weight, a.bias, a.resid, a.out, (long long)a.B * a.H * a.W, a.N, ln, st);
    if (a.kind == DDK_CONV1X1 && a.c1 == 0 && !a.defer_reduce && !a.gn_partials && !fuse && !ln && !tuning_flag("DDK_NO_CONV1X1_STREAM") &&
        conv1x1_stream_ok((long long)a.B * a.H * a.W, a.c0, a.N))
        // 32 / 64 channels on both sides of a large map (the encoder / decoder blocks of the dDDPM and their input-gradient convs): a
        // memory stream -- weights in registers, no LDS, float4 epilogue (conv1x1_stream.hip)
        return conv1x1_stream(a.src0, a.c0, a.weight, a.bias, a.dmish_src, a.resid, a.out, a.mish_out, (long long)a.B * a.H * a.W, a.N,
                              a.pre_mish, a.post_mish, st);
    DDK_REQUIRE(!a.gn_partials, "conv: gn_partials is only produced by the Winograd path (weight_wino given, ddk_conv_gn_partials() > 0)");
    DDK_REQUIRE(!fuse, "conv: the in-launch GroupNorm exists on the Winograd path only");
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
    p.debug = tuning_int("DDK_DEBUG", 0);   // diagnostic instantiations of the halo kernel (DDK_TUNING build only)
    p.post_mish = a.post_mish;
    p.mish_out = a.mish_out; p.dmish_src = a.dmish_src;
    p.slab_stride = (long long)a.B * g.Ho * g.Wo * a.N;
    p.dWm = make_fastdiv_u((unsigned)g.Wm);
    p.dHm = make_fastdiv_u((unsigned)g.Hm);
    p.tapmode = (a.kind == DDK_CONV1X1) ? 0 : (a.kind == DDK_CONVT4X4_S2 ? 2 : (a.kind == DDK_CONV4X4_S2 ? 3 : 1));
    if (!ln && !a.defer_reduce && choose_c32(a.kind, a.B, a.H, a.W, p.cin, p.N, a.c1, a.pre_mish != 0)) {
        const int TR = 128 / a.W, TB = TR > a.H ? TR / a.H : 1, rows_img = TB > 1 ? a.H : TR;
        p.dImg = make_fastdiv_u((unsigned)((rows_img + 2) * (a.W + 2)));
        p.dWp = make_fastdiv_u((unsigned)(a.W + 2));
        p.dRows = make_fastdiv_u((unsigned)rows_img);
        p.splits = 1;
        p.kiters_per_split = p.kiters;
        DDK_TRY(ensure_device_init());
        return launch_c32(p, st);
    }
    const ConvPlan plan = plan_conv(a.kind, a.B, a.H, a.W, p.cin, p.N, g, a.pre_mish != 0);
    const Choice c = plan.c;
    if (tuning_flag("DDK_TRACE")) {   // tuning aid: one line per conv launch
        int bm, bn;
        tile_dims(c.tile, bm, bn);
        fprintf(stderr, "[ddk] conv kind=%d B=%d %dx%d cin=%d N=%d M=%d kiters=%d -> %s tile %dx%d splits=%d (kps %d) stages=%d wgs=%lld\n",
                a.kind, a.B, a.H, a.W, p.cin, p.N, p.M, p.kiters, plan.halo ? "halo" : "igemm", bm, bn, c.splits, c.kps, c.stages,
                ceil_div(p.M, bm) * ceil_div(p.N, bn) * p.nphase * c.splits);
    }
    p.splits = c.splits;
    p.kiters_per_split = c.kps;
    if (ln) {
        DDK_REQUIRE(a.kind == DDK_CONV1X1 && !plan.halo && c.splits == 1 && a.c1 == 0 && !a.pre_mish &&
                        (c.tile == T64x64 || c.tile == T128x64 || c.tile == T128x128),
                    "conv: LayerNorm folding needs an unsplit single-source 1x1 conv on a 4-wave tile (conv_ln_fold_ok)");
        DDK_REQUIRE(ln->c1 && ln->c2 && aligned16(ln->c1) && aligned16(ln->c2), "conv: LayerNorm folding vectors");
        p.ln_c1 = ln->c1; p.ln_c2 = ln->c2; p.ln_eps = ln->eps;
    }
    if (plan.halo) {   // tile geometry of conv3x3_halo_kernel, for its exact-division helpers
        const int TR = 128 / a.W, TB = TR > a.H ? TR / a.H : 1, rows_img = TB > 1 ? a.H : TR;
        p.dImg = make_fastdiv_u((unsigned)((rows_img + 2) * (a.W + 2)));
        p.dWp = make_fastdiv_u((unsigned)(a.W + 2));
        p.dRows = make_fastdiv_u((unsigned)rows_img);
    }
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
    DDK_TRY(ensure_device_init());
    DDK_TRY(plan.halo ? launch_halo(p, st) : launch_tile(p, c, st));
    if (c.splits > 1 && !a.defer_reduce) {
        const long long n4 = p.slab_stride / 4;
        const int blocks = (int)(ceil_div(n4, 256) < 2048 ? ceil_div(n4, 256) : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a.workspace),
                           c.splits, p.slab_stride, a.bias, a.resid, final_out, n4, p.N, p.post_mish, a.mish_out, a.dmish_src);
        DDK_TRY(check_launch("splitk_reduce_kernel"));
    }
    return DDK_OK;
}

int conv_splits(int kind, int B, int H, int W, int cin, int N) {
    Geometry g;
    if (!conv_geometry(kind, H, W, g) || cin <= 0 || cin % 32) return 1;
    return plan_conv(kind, B, H, W, cin, N, g).c.splits;
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

// diagnostic: copy the stamp buffer (6 x 4096 u64) to the host and clear it (only the -DDDK_TUNING build writes stamps)
extern "C" int ddk_debug_read_stamps(unsigned long long* host_out) {
    if (!host_out) return ddk::fail_arg("debug_read_stamps: null");
#ifndef DDK_TUNING
    ddk::set_error("debug_read_stamps: this is the production build; use libddk_tune.so (make tune)");
    return DDK_ERR_ARG;
#endif
    DDK_HIP(hipDeviceSynchronize());
    DDK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ddk::g_stamps), sizeof(unsigned long long) * 6 * 4096));
    static unsigned long long zeros[6 * 4096];
    DDK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ddk::g_stamps), zeros, sizeof(zeros)));
    return DDK_OK;
}

extern "C" int ddk_conv_splits(int kind, int B, int H, int W, int cin, int N) { return ddk::conv_splits(kind, B, H, W, cin, N); }

extern "C" int ddk_conv_forward(const ddk_conv_args* a, ddk_stream_t s) {
    if (!a) return ddk::fail_arg("conv: null args");
    return ddk::conv_forward(*a, ddk::as_stream(s));
}
