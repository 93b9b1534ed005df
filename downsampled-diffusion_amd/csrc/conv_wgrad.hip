// conv_wgrad.hip -- backward of the conv family: weight gradient (fp32 MFMA GEMM contracting over pixels), bias
// gradient, input-gradient weight packing, zero-stuffing for the stride-2 conv's input gradient.
//
// These are the autograd counterparts of reference models/unet/blocks.py:35,44,78,103,123-124 (Conv2d /
// ConvTranspose2d), driven by the .backward() of trainers/trainer_ddpm.py:124-128,225-229.
//
// Input gradients reuse the forward implicit-GEMM (conv_igemm.hip) with re-packed weights:
//   conv3x3 s1 / 1x1 : dX = conv(dY, W')          W'[c][t][n] = W[n][c][T-1-t]            (ddk_pack_conv_weight_dgrad)
//   conv3x3 s2       : dX = conv3x3_s1(S, W')      S = dY zero-stuffed to the input size   (ddk_zero_stuff2)
//   convT 4x4 s2     : dX = conv4x4_s2(dY, W)      the (I,O,4,4) tensor read as OIHW        (DDK_CONV4X4_S2)
//
// Weight gradient: dW[n][c][tap] = sum_m dY[m][n] * X[pix(m) + tap][c]  -- a GEMM with rows n, columns c (one tap
// per workgroup), contracting over the M = B*H*W pixels.  Both operands are pixel-major ([m][channels]), so a
// 32-pixel k-chunk of either is [32][tile] with the tile's channels contiguous: staged by LDS-DMA as is, and the MFMA
// operand of lane l (row/col = l & 31, k = l >> 5) is a conflict-free ds_read_b32 (32 lanes read 32 consecutive
// floats).  M is split over workgroups into fp32 slabs [split][n][tap][c]; a second kernel sums the slabs in a fixed
// order and ACCUMULATES into the canonical (OIHW) gradient tensor -- deterministic, and gradient accumulation over
// micro-batches (trainer_ddpm.py:118-131) comes for free.
#include "ddk_internal.h"
#include "pack_elems.h"

namespace ddk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FastDiv {  // exact unsigned division by a runtime constant (Granlund-Montgomery)
    unsigned mul, sh1, sh2, d;
};
static FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    const unsigned t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct WgradParams {
    const float* x;   // [B][H][W][cx]
    const float* dy;  // [M][N], M = B*Hm*Wm
    float* slab;      // [splits][N][ntaps][cx]
    float* bias_slab; // [splits][N] column sums of dy (the bias gradient's slabs), or null
    int cx, N;
    int H, W, Hm, Wm, M;
    int in_stride, tapmode, ntaps;
    int splits, rows_per_split;
    FastDiv dW_, dH_;
};

__device__ __forceinline__ void wg_tap_offset(int tapmode, int tap, int& dy, int& dx) {
    if (tapmode == 1) {
        const int ty = (tap * 11) >> 5;
        dy = ty - 1;
        dx = tap - 3 * ty - 1;
    } else if (tapmode == 3) {
        dy = (tap >> 2) - 1;
        dx = (tap & 3) - 1;
    } else {
        dy = 0;
        dx = 0;
    }
}

__device__ __attribute__((aligned(128))) float g_zero_page_w[32];

__device__ __forceinline__ void lds_dma16_w(const float* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(g) : "memory", "m0");
}

// BN x BC output tile (rows n, cols c) of one tap; WN x WC waves, each owning (BN/WN) x (BC/WC).
template <int BN, int BC, int WN, int WC>
__global__ __launch_bounds__(WN* WC * 64) void wgrad_kernel(const WgradParams p) {
    constexpr int NW = WN * WC;
    constexpr int TN = BN / (WN * 32), TC = BC / (WC * 32);
    constexpr int A_PIECES = BN / 8, B_PIECES = BC / 8;     // 1-KiB pieces per stage (32 rows x tile floats / 256)
    constexpr int A_PW = (A_PIECES + NW - 1) / NW, B_PW = (B_PIECES + NW - 1) / NW;
    constexpr int STAGE = 32 * (BN + BC);                   // floats
    static_assert(TN >= 1 && TC >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int wn = wid / WC, wc = wid % WC;
    const int n0 = blockIdx.x * BN, c0 = blockIdx.y * BC;
    const int tap = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    int tdy, tdx;
    wg_tap_offset(p.tapmode, tap, tdy, tdx);

    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int n_it = m_begin < m_end ? (m_end - m_begin + 31) / 32 : 0;
    const float* zero = g_zero_page_w + (lane & 7) * 4;

    // lane's position inside a 256-float piece of a [32][T] tile: float index j*256 + lane*4
    auto issue = [&](int stage, int mbase) {
        const unsigned st = lds_base + (unsigned)(stage * STAGE * 4);
#pragma unroll
        for (int jj = 0; jj < A_PW; ++jj) {
            const int j = wid_u * A_PW + jj;
            if (j < A_PIECES) {
                const int idx = j * 256 + lane * 4;
                const int ml = idx / BN, nl = idx % BN;
                const int m = mbase + ml;
                const bool ok = m < m_end && n0 + nl < p.N;
                const float* g = ok ? p.dy + (long long)m * p.N + n0 + nl : zero;
                lds_dma16_w(g, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(j * 1024))));
            }
        }
#pragma unroll
        for (int jj = 0; jj < B_PW; ++jj) {
            const int j = wid_u * B_PW + jj;
            if (j < B_PIECES) {
                const int idx = j * 256 + lane * 4;
                const int ml = idx / BC, cl = idx % BC;
                const unsigned m = (unsigned)(mbase + ml);
                const unsigned q1 = fdiv(m, p.dW_);
                const int xm = (int)(m - q1 * p.dW_.d);
                const unsigned b = fdiv(q1, p.dH_);
                const int ym = (int)(q1 - b * p.dH_.d);
                const int iy = ym * p.in_stride + tdy, ix = xm * p.in_stride + tdx;
                const bool ok = (int)m < m_end && c0 + cl < p.cx && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const float* g = ok ? p.x + (((long long)b * p.H + iy) * p.W + ix) * p.cx + c0 + cl : zero;
                lds_dma16_w(g, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(32 * BN * 4 + j * 1024))));
            }
        }
    };

    f32x16 acc[TN][TC];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradient = column sums of dY: the MFMA operand a lane reads from the dY tile is dY[k = 2q + (lane >> 5)][n = lane & 31],
    // so a running sum of those registers over the k-pairs IS the column sum of half the rows -- TN adds per k-pair in the
    // workgroups of the first channel tile and tap, no separate pass over dY
    const bool do_bias = p.bias_slab != nullptr && blockIdx.y == 0 && tap == 0;
    float bsum[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) bsum[i] = 0.f;

    const int a_off = wn * TN * 32 + (lane & 31) + (lane >> 5) * BN;
    const int b_off = 32 * BN + wc * TC * 32 + (lane & 31) + (lane >> 5) * BC;

    if (n_it > 0) issue(0, m_begin);
    {   // keep SMEM out of the k-loop (see conv_igemm.hip consume_epilogue_args): epilogue-only arguments are consumed here
        const int nt = p.ntaps, NN = p.N, cxx = p.cx;
        asm volatile("" ::"s"(p.slab), "s"(p.bias_slab), "s"(nt), "s"(NN), "s"(cxx));
    }
    for (int k = 0; k < n_it; ++k) {
        const int stage = k & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (k + 1 < n_it) issue(stage ^ 1, m_begin + (k + 1) * 32);
        const float* As = smem + stage * STAGE + a_off;
        const float* Bs = smem + stage * STAGE + b_off;
        // operands of k-pair q+2 are requested before the MFMAs of k-pair q (two pairs = 2*TN*TC MFMAs of cover for the
        // ds_read latency); the fences keep hipcc from sinking the reads back to their first use
        float a[3][TN], b[3][TC];
        auto load_pair = [&](int slot, int kk) {
#pragma unroll
            for (int i = 0; i < TN; ++i) a[slot][i] = As[kk * BN + i * 32];
#pragma unroll
            for (int j = 0; j < TC; ++j) b[slot][j] = Bs[kk * BC + j * 32];
        };
        load_pair(0, 0);
        load_pair(1, 2);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cur = q % 3, nxt = (q + 2) % 3;
            if (q + 2 < 16) load_pair(nxt, 2 * (q + 2));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TN; ++i) bsum[i] += a[cur][i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (do_bias && wc == 0) {                          // even-k half + odd-k half; the waves of one wn hold the same sums
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const float t = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int n = n0 + (wn * TN + i) * 32 + (lane & 31);
            if (lane < 32 && n < p.N) p.bias_slab[(long long)split * p.N + n] = t;
        }
    }

    // slab[split][n][tap][c]: col (c) = lane & 31 -> 128-byte contiguous stores
    float* outp = p.slab + (long long)split * p.N * p.ntaps * p.cx;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (wn * TN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n >= p.N) continue;
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                const int c = c0 + (wc * TC + j) * 32 + (lane & 31);
                if (c < p.cx) outp[((long long)n * p.ntaps + tap) * p.cx + c] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 weight gradient with the INPUT HALO staged once per 32-pixel chunk and ALL 9 TAPS accumulated by the same
// workgroup (the per-tap kernel above re-loads dY for each of its 9 tap workgroups and streams 32 KB per 4096 MFMA
// cycles -- the CU's LDS-DMA rate, see conv_igemm.hip).  Tile: 64 output channels (n) x 64 input channels (c) x 9 taps;
// 4 waves (2x2), each 32 x 32 x 9 taps = 9 accumulators.  A k-chunk is 32 consecutive output pixels = R = 32/WC whole
// rows of width WC (W <= 32), or a 32-pixel segment of one row (W > 32, WC = 32).  Per chunk the workgroup DMAs
//   dYs[32 px][64 n]                                   8 KB
//   Xh[R][3 rows][WC+2 cols][64 c]   (zero outside)    26-36 KB
// and runs 16 k-pairs x 9 taps MFMAs per wave: 34-44 KB per 9216 MFMA cycles instead of 72 KB.  The X operand of tap
// (dy, dx) for pixel k is halo pixel hb(k) + (dy+1)*(WC+2) + (dx+1): a compile-time immediate on top of a per-lane base.
struct WgHaloParams {
    const float* x;    // [B][H][W][cx]
    const float* dy;   // [M][N]
    float* slab;       // [splits][N][9][cx]
    float* bias_slab;  // [splits][N] column sums of dy, or null
    int cx, N, B, H, W, M;
    int splits, chunks_per_split, n_chunks;
    FastDiv dH_, dW_;
};

template <int WC>
__global__ __launch_bounds__(256) void wgrad3x3_halo_kernel(const WgHaloParams p) {
    constexpr int R = 32 / WC;                  // rows per chunk (W <= 32); 1 when WC == 32 (also covers W > 32)
    constexpr int WP = WC + 2;
    constexpr int HALO_PX = R * 3 * WP;
    constexpr int X_PIECES = (HALO_PX + 3) / 4; // 1-KiB pieces of 4 halo pixels x 64 channels
    constexpr int STAGE = 32 * 64 + X_PIECES * 4 * 64;   // floats
    constexpr int PIECES = 8 + X_PIECES;
    constexpr int PW = (PIECES + 3) / 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int wn = wid >> 1, wc = wid & 1;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64, split = blockIdx.z;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const int g_begin = split * p.chunks_per_split;
    const int g_end = min(p.n_chunks, g_begin + p.chunks_per_split);
    const int n_it = g_end - g_begin;
    const float* zero = g_zero_page_w + (lane & 7) * 4;
    const int ppx = lane >> 4, pc4 = (lane & 15) * 4;    // this lane's pixel inside a piece and its channel quad

    auto issue = [&](int stage, int g) {
        const unsigned st = lds_base + (unsigned)(stage * STAGE * 4);
        const int m0 = g * 32;
        // chunk origin: global row (b*H + y) of its first pixel and first column
        int grow0, x0;
        if (WC == 32) { const unsigned q = fdiv((unsigned)m0, p.dW_); grow0 = (int)q; x0 = m0 - (int)q * p.W; }
        else { grow0 = m0 / WC; x0 = 0; }
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const int j = wid_u * PW + jj;            // wave-uniform piece index
            if (j < 8) {
                const int m = m0 + j * 4 + ppx;
                const float* g_ = m < p.M ? p.dy + (long long)m * p.N + n0 + pc4 : zero;
                lds_dma16_w(g_, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(j * 1024))));
            } else if (j < PIECES) {
                const int hp = (j - 8) * 4 + ppx;
                const int r = hp / (3 * WP), rem = hp - r * (3 * WP);
                const int dyi = rem / WP, xx = rem - dyi * WP;
                const unsigned grow = (unsigned)(grow0 + r);
                const unsigned b = fdiv(grow, p.dH_);
                const int yy = (int)(grow - b * (unsigned)p.H) + dyi - 1;
                const int xc = x0 + xx - 1;
                const bool ok = hp < HALO_PX && (int)b < p.B && (unsigned)yy < (unsigned)p.H && (unsigned)xc < (unsigned)p.W;
                const float* g_ = ok ? p.x + (((long long)b * p.H + yy) * p.W + xc) * p.cx + c0 + pc4 : zero;
                lds_dma16_w(g_, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(32 * 64 * 4 + (j - 8) * 1024))));
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = p.bias_slab != nullptr && blockIdx.y == 0;      // column sums of dY on the side (see wgrad_kernel)
    float bsum = 0.f;

    // per-lane operand offsets (floats): k = 2*kk + (lane >> 5)
    const int fh = lane >> 5, l31 = lane & 31;
    const int a_off = fh * 64 + wn * 32 + l31;                      // + kk * 128
    int hb_off[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = 2 * kk + fh;
        const int r = k / WC, xl = k - r * WC;
        hb_off[kk] = 32 * 64 + (r * 3 * WP + xl) * 64 + wc * 32 + l31;   // tap (dy,dx) adds ((dy+1)*WP + dx+1) * 64
    }

    if (n_it > 0) issue(0, g_begin);
    {   // keep SMEM out of the k-loop (conv_igemm.hip consume_epilogue_args)
        const int NN = p.N, cxx = p.cx, sp = p.splits;
        asm volatile("" ::"s"(p.slab), "s"(p.bias_slab), "s"(NN), "s"(cxx), "s"(sp));
    }
    for (int it = 0; it < n_it; ++it) {
        const int stage = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < n_it) issue(stage ^ 1, g_begin + it + 1);
        const float* S = smem + stage * STAGE;
        float a[2], b[2][9];
        auto load_pair = [&](int slot, int kk) {
            a[slot] = S[a_off + kk * 128];
#pragma unroll
            for (int t = 0; t < 9; ++t) b[slot][t] = S[hb_off[kk] + ((t / 3) * WP + (t % 3)) * 64];
        };
        load_pair(0, 0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < 16) load_pair(cur ^ 1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur], b[cur][t], acc[t], 0, 0, 0);
            if (do_bias) bsum += a[cur];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (do_bias && wc == 0) {
        const float t = bsum + __shfl_xor(bsum, 32, 64);
        const int n = n0 + wn * 32 + l31;
        if (lane < 32 && n < p.N) p.bias_slab[(long long)split * p.N + n] = t;
    }

    // epilogue: per tap, this wave's 32x32 block -> LDS (pitch 40) -> float4 rows -> slab[split][n][tap][c]
    __syncthreads();
    float* Es = smem + wid * 32 * 40;
    float* outp = p.slab + (long long)split * p.N * 9 * p.cx;
    const int row_l = lane >> 3, col4 = (lane & 7) * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * fh) * 40 + l31] = acc[t][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + row_l;
            const int n = n0 + wn * 32 + row;
            const float4 v = *reinterpret_cast<const float4*>(Es + row * 40 + col4);
            if (n < p.N) *reinterpret_cast<float4*>(outp + ((long long)n * 9 + t) * p.cx + c0 + wc * 32 + col4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// grad[(n*cw + c_off + c)*ntaps + tap] += sum_s slab[s][n][tap][c]   (only the first c_real channels: the input may be padded)
// One workgroup per (n, channel block, tap): row groups of cxb lanes read the [cxb c] row of every G-th slab coalesced, four
// loads in flight each, and the G partial sums are added in a fixed order -- deterministic.
// Grid row y == ntaps (present when the launch carries a bias gradient) sums the [splits][N] column-sum slabs into grad_b.
// Channels of one (n, tap) row a workgroup sums: the widest of 256 / 128 / 64 / 32 that divides cx -- the 256 threads are
// 256 / cxb row groups x cxb channels.  (Round 4: with 32 channels per workgroup whatever cx, a 256 -> 256 3x3 weight was 18 432
// workgroups of 32 outputs each and the launch was bound by workgroup dispatch, not by its few MB of slabs: 232 launches of 4-7 us
// per cfg3 optimiser step, and batching them into one launch changed nothing.)
// ... but a thread should not walk more than ~16 slabs on its own (4 iterations of 4 loads in flight: one workgroup per (n, block)
// summing 256 x 9 rows serially took 38 us, profiles/r03_train_wgrad.txt): many splits -> more row groups, narrower blocks.
__host__ __device__ static inline int wg_cxb(int cx, int splits) {
    int cxb = splits <= 16 ? 256 : splits <= 32 ? 128 : splits <= 64 ? 64 : 32;
    while (cx % cxb) cxb >>= 1;
    return cxb;
}

__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slab, int splits, long long slab_stride, float* __restrict__ grad,
                                                   int N, int ntaps, int cx, int c_real, int cw, int c_off,
                                                   const float* __restrict__ bias_slab, float* __restrict__ grad_b, int bx, int by,
                                                   float (&part)[256]) {
    const bool bias_row = by == ntaps;                      // extra grid row: grad_b[n] += sum_s bias_slab[s][n], 32 n per workgroup
    if (bias_row) {
        if (bx * 32 >= N) return;
        const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
        const float* src = bias_slab + bx * 32 + c;
        float s = 0.f;
        for (int k = g; k < splits; k += 8) s += src[(long long)k * N];
        part[g * 32 + c] = s;
        __syncthreads();
        if (threadIdx.x < 32) {
            float t = part[c];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += part[j * 32 + c];
            grad_b[bx * 32 + c] += t;
        }
        return;
    }
    const int cxb = wg_cxb(cx, splits), G = 256 / cxb, cblocks = cx / cxb;
    const int c = threadIdx.x % cxb, g = threadIdx.x / cxb;
    const int n = bx / cblocks, cb = bx % cblocks, tap = by;
    const float* src = slab + ((long long)n * ntaps + tap) * cx + cb * cxb + c;
    float s = 0.f;
    int k = g;
    for (; k + 3 * G < splits; k += 4 * G) {                // four loads in flight per thread
        const float a0 = src[k * slab_stride], a1 = src[(k + G) * slab_stride];
        const float a2 = src[(k + 2 * G) * slab_stride], a3 = src[(k + 3 * G) * slab_stride];
        s += (a0 + a1) + (a2 + a3);
    }
    for (; k < splits; k += G) s += src[k * slab_stride];
    part[g * cxb + c] = s;
    __syncthreads();
    if (threadIdx.x < cxb) {
        float t = part[c];
        for (int j = 1; j < G; ++j) t += part[j * cxb + c];                      // fixed order: deterministic
        if (cb * cxb + c < c_real) grad[((long long)n * cw + c_off + cb * cxb + c) * ntaps + tap] += t;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, long long slab_stride,
                                                           float* __restrict__ grad, int N, int ntaps, int cx, int c_real, int cw,
                                                           int c_off, long long total, const float* __restrict__ bias_slab,
                                                           float* __restrict__ grad_b) {
    __shared__ float part[256];
    (void)total;
    wgrad_reduce_block(slab, splits, slab_stride, grad, N, ntaps, cx, c_real, cw, c_off, bias_slab, grad_b, (int)blockIdx.x, (int)blockIdx.y, part);
}

// The slab reduces of a whole backward pass in ONE launch (ddk_wgrad_reduce_jobs): every weight gradient of the pass left its slabs
// in a buffer of its own (ddk_conv_wgrad_defer) and a job record; the records travel as KERNEL ARGUMENTS (48 per launch: no device
// table, nothing to copy inside a graph capture), a block finds its job by binary search over the jobs' first-block prefix sums and
// then is exactly one block of wgrad_reduce_kernel -- same summation order, same bits.  232 reduce launches of 4-7 us
// per cfg3 optimiser step before (profiles/r04_train_cfg3_kernel_stats.csv), most of them launch-bound on a few MB of slabs.
constexpr int WG_JOBS_PER_LAUNCH = 48;      // 48 x 80 B + 8 = 3848 B of kernel arguments (limit 4 KB)
struct WgradJobPack {
    int n, pad;
    ddk_wgrad_reduce_job j[WG_JOBS_PER_LAUNCH];
};
__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const WgradJobPack pk) {
    __shared__ float part[256];
    const long long blk = blockIdx.x;
    int lo = 0, hi = pk.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pk.j[mid].block0 <= blk) lo = mid;
        else hi = mid - 1;
    }
    const ddk_wgrad_reduce_job& j = pk.j[lo];
    const int local = (int)(blk - j.block0);
    const int gx = j.N * (j.cx / wg_cxb(j.cx, j.splits));
    wgrad_reduce_block(j.slab, j.splits, j.slab_stride, j.grad, j.N, j.ntaps, j.cx, j.c_real, j.cw, j.c_off, j.bias_slab, j.grad_b, local % gx,
                       local / gx, part);
}

// bias gradient: partial[s][n] = sum over the s-th row range of dy[m][n]; then grad[n] += sum_s partial
// part[blockIdx.x][n] = sum over this workgroup's row range of dy[m][n].  A thread owns one float4 column group and
// every RG-th row (RG = 256 / (N/4) row groups), 4 independent accumulators keep loads in flight; row groups are then
// added in fixed order through LDS.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, float* __restrict__ part, long long M, int N,
                                                             long long rows_per) {
    __shared__ float4 red[256];
    const int cq = N >> 2;                    // float4 column groups (<= 256)
    const int rg_n = 256 / cq;                // row groups
    const int c = threadIdx.x % cq, rg = threadIdx.x / cq;
    const long long m0 = blockIdx.x * rows_per, m1 = (m0 + rows_per < M) ? m0 + rows_per : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rg < rg_n) {
        const float4* p = reinterpret_cast<const float4*>(dy) + c;
#pragma unroll 4
        for (long long m = m0 + rg; m < m1; m += rg_n) {
            const float4 v = p[m * cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < cq) {
        float4 t = red[threadIdx.x];
        for (int i = 1; i < rg_n; ++i) {
            const float4 v = red[i * cq + threadIdx.x];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        reinterpret_cast<float4*>(part + (long long)blockIdx.x * N)[threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ grad, int N,
                                                           int accumulate) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int i = rg; i < nparts; i += 4) s += part[(long long)i * N + n];
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && n < N) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        grad[n] = accumulate ? grad[n] + t : t;
    }
}

// dst[i][t][o] = w[o][i][T-1-t] (i < I), zero rows up to i_pad; o padded to o_pad with zeros: pack_elems.h
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I, int taps,
                                                         int i_pad, int o_pad, long long total) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        pack_dgrad_elem(w, dst, idx, O, I, taps, o_pad);
}

// out[b][2y][2x][c] = in[b][y][x][c], zeros elsewhere; out is [B][Ho][Wo][C] (Ho in {2H-1, 2H})
__global__ __launch_bounds__(256) void zero_stuff2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int Ho,
                                                          int Wo, int C, long long total4) {
    const int c4 = C >> 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % c4);
        long long pp = i / c4;
        const int xo = (int)(pp % Wo); pp /= Wo;
        const int yo = (int)(pp % Ho);
        const long long b = pp / Ho;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(yo & 1) && !(xo & 1) && (yo >> 1) < H && (xo >> 1) < W)
            v = *reinterpret_cast<const float4*>(in + ((b * H + (yo >> 1)) * W + (xo >> 1)) * C + cq * 4);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ host
struct WgChoice { int bn, bc, splits, rows_per; };

static int tile32(int v) { return v >= 128 ? 128 : (v >= 64 ? 64 : 32); }

static WgChoice wgrad_choice(int N, int cx, int ntaps, long long M) {
    WgChoice c;
    c.bn = tile32(N);
    c.bc = tile32(cx);
    int target_waves = 2048;
    // Few pixels (the UNet levels below 32x32 at batch 64): a 128x128 tile is one wave per SIMD walking 8 chunks with a drain per chunk;
    // 64x64 tiles give four times the workgroups to hide that latency behind -- the eight per-tap shapes of the cfg3 step 316 -> 222 us
    // (tools/wgrad_bench_cfg3.py, profiles/r03_train_wgrad.txt).  From 65536 pixels on the larger tile wins (101 vs 110 us).
    if (M < 65536) { if (c.bn > 64) c.bn = 64; if (c.bc > 64) c.bc = 64; }
#ifdef DDK_TUNING
    if (const char* e = getenv("DDK_WGRAD_TILE")) { const int t = atoi(e); if (c.bn > t) c.bn = t; if (c.bc > t) c.bc = t; }   // A/B knobs
    if (const char* e = getenv("DDK_WGRAD_WAVES")) target_waves = atoi(e);
#endif
    const long long tiles = ceil_div(N, c.bn) * ceil_div(cx, c.bc) * ntaps;
    // The loop is a 2-stage ring with a drain per 32-pixel chunk: a wave hides its DMA latency only behind OTHER waves of
    // its SIMD.  4-wave tiles want ~512 workgroups (two per CU); the one- and two-wave tiles of narrow convs (the 32 / 64
    // channel encoder / decoder of the dDDPM at up to 262144 pixels) ran 64 splits = 64-513 waves on 1024 SIMDs at 1.5 us
    // per chunk (235 us for 4.8 GFLOP); they get the same ~2048 waves now (profiles/r03_train_wgrad.txt).
    const int waves = (c.bn >= 64 && c.bc >= 64) ? 4 : ((c.bn * c.bc) / 1024 >= 4 ? 4 : (c.bn * c.bc) / 1024);
    long long s = ceil_div(target_waves / waves, tiles);
    const long long max_s = M / 256 > 0 ? M / 256 : 1;   // at least 8 k-chunks per split
    if (s > max_s) s = max_s;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    // Round 6 (profiles/r06_wgrad64_pmc.json: 576 workgroups = 2.25 per CU, so the CUs with three set the time at 0.75 of the fill):
    // among this and the next few split counts take the one whose workgroup count fills whole multiples of the 256 CUs best
    if (waves == 4 && tiles * s > 256) {
        auto fill = [&](long long ss) { const long long w = tiles * ss; return (double)w / (double)(ceil_div(w, 256) * 256); };
        long long best = s;
        for (long long cand = s + 1; cand <= s + 4 && cand <= max_s && cand <= 256; ++cand)
            if (fill(cand) > fill(best) + 0.04) best = cand;
        s = best;
    }
    c.rows_per = (int)(ceil_div(ceil_div(M, s), 32) * 32);
    c.splits = (int)ceil_div(M, c.rows_per);
    return c;
}

static bool wgrad_geometry(int kind, int H, int W, int& Hm, int& Wm, int& stride, int& tapmode, int& ntaps) {
    switch (kind) {
        case DDK_CONV3X3_S1: Hm = H; Wm = W; stride = 1; tapmode = 1; ntaps = 9; return true;
        case DDK_CONV3X3_S2: Hm = (H - 1) / 2 + 1; Wm = (W - 1) / 2 + 1; stride = 2; tapmode = 1; ntaps = 9; return true;
        case DDK_CONV1X1: Hm = H; Wm = W; stride = 1; tapmode = 0; ntaps = 1; return true;
        case DDK_CONV4X4_S2: Hm = H / 2; Wm = W / 2; stride = 2; tapmode = 3; ntaps = 16; return H % 2 == 0 && W % 2 == 0;
        default: return false;
    }
}

// halo weight-gradient kernel: eligibility and split over pixel chunks (~512 workgroups, >= 8 chunks each)
static bool wgrad_halo_plan(int kind, int B, int H, int W, int cx, int N, int& wc, int& splits, int& cps, int& n_chunks) {
#ifdef DDK_TUNING
    const bool off = getenv("DDK_NO_WGRAD_HALO") != nullptr;   // A/B knob (tuning build only)
#else
    const bool off = false;
#endif
    if (off || kind != DDK_CONV3X3_S1 || N % 64 || cx % 64) return false;
    if (W <= 32) { if (W < 4 || 32 % W) return false; wc = W; }
    else { if (W % 32) return false; wc = 32; }
    const long long M = (long long)B * H * W;
    if (M % 32) return false;                      // whole chunks only (rows / images never straddle a chunk boundary badly)
    if (W < 32 && ((long long)B * H) % (32 / W)) return false;
    n_chunks = (int)(M / 32);
    if (n_chunks < 128) return false;              // tiny maps: the per-tap kernel's 9x more workgroups win
    const long long tiles = (long long)(N / 64) * (cx / 64);
#ifdef DDK_TUNING
    const int target = getenv("DDK_WGRAD_HALO_WGS") ? atoi(getenv("DDK_WGRAD_HALO_WGS")) : (n_chunks >= 1024 ? 512 : 256);
#else
    // 512 workgroups are 5-8 % faster than 256 on the large maps (19 GFLOP layers: 194 -> 179 us, 186 -> 177; round 2 saw no
    // difference before the reduce kernel was widened); on small ones 256 halve the slabs for the same time
    const int target = n_chunks >= 1024 ? 512 : 256;
#endif
    long long s = ceil_div(target, tiles);
    const long long max_s = n_chunks / 4 > 0 ? n_chunks / 4 : 1;   // at least 4 chunks per split (8 left 128 -> 256 @8x8 at 128 workgroups)
    if (s > max_s) s = max_s;
    if (s > 128) s = 128;
    cps = (int)ceil_div(n_chunks, s);
    splits = (int)ceil_div(n_chunks, cps);
    return true;
}

template <int WC>
constexpr size_t wgrad_halo_lds_bytes() {
    constexpr int HALO_PX = (32 / WC) * 3 * (WC + 2);
    return 2 * (size_t)(32 * 64 + ((HALO_PX + 3) / 4) * 4 * 64) * sizeof(float);
}

template <int WC>
static int launch_wgrad_halo(const WgHaloParams& p, hipStream_t st) {
    constexpr size_t lds = wgrad_halo_lds_bytes<WC>();
    dim3 grid((unsigned)(p.N / 64), (unsigned)(p.cx / 64), (unsigned)p.splits);
    hipLaunchKernelGGL((wgrad3x3_halo_kernel<WC>), grid, dim3(256), lds, st, p);
    return check_launch("wgrad3x3_halo_kernel");
}

template <int BN, int BC, int WN, int WC>
static int launch_wgrad(const WgradParams& p, int ntiles_n, int ntiles_c, hipStream_t st) {
    constexpr size_t lds = 2 * 32 * (size_t)(BN + BC) * sizeof(float);
    dim3 grid((unsigned)ntiles_n, (unsigned)ntiles_c, (unsigned)(p.ntaps * p.splits));
    hipLaunchKernelGGL((wgrad_kernel<BN, BC, WN, WC>), grid, dim3(WN * WC * 64), lds, st, p);
    return check_launch("wgrad_kernel");
}


// ------------------------------------------------------------------------------------------------
// The same halo form for the NARROW 3x3 convs (N == 32, C == 32: the dDDPM encoder / decoder blocks at up to 262 144 pixels), where
// the 64x64 tile above does not apply and the per-tap kernel re-reads both operands nine times (604 MB through the fabric for the
// 64x64x64-pixel layer, 99 us).  One (n, c) tile = the whole filter, so the grid splits only the pixels: a workgroup is 4 INDEPENDENT
// waves, each walking its own chunks (32 pixels: dY [32][32] + the chunk's input halo, double-buffered in the wave's private LDS,
// no workgroup barrier in the loop) with all 9 taps accumulated per chunk (9 accumulators); at the end the four waves' accumulators
// are summed in order through LDS and leave as ONE slab per workgroup.
template <int WC>
struct WgHalo32 {
    static constexpr int R = 32 / WC, WP = WC + 2, HALO_PX = R * 3 * WP;
    static constexpr int X_PIECES = (HALO_PX + 7) / 8;                 // 1-KiB pieces of 8 halo pixels x 32 channels
    static constexpr int STAGE = 32 * 32 + X_PIECES * 8 * 32;          // floats
    static constexpr size_t lds_bytes = (size_t)4 * 2 * STAGE * sizeof(float);
};

template <int WC>
__global__ __launch_bounds__(256) void wgrad3x3_halo32_kernel(const WgHaloParams p) {
    using G = WgHalo32<WC>;
    constexpr int WP = G::WP, STAGE = G::STAGE, X_PIECES = G::X_PIECES, HALO_PX = G::HALO_PX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    float* wbuf = smem + wave * 2 * STAGE;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)wbuf;
    const int g_begin = split * p.chunks_per_split;
    const int g_end = min(p.n_chunks, g_begin + p.chunks_per_split);
    const float* zero = g_zero_page_w + (lane & 7) * 4;
    const int ppx = lane >> 3, pc4 = (lane & 7) * 4;                   // this lane's pixel inside a piece and its channel quad

    auto issue = [&](int stage, int g) {
        const unsigned st = lds_base + (unsigned)(stage * STAGE * 4);
        const int m0 = g * 32;
        int grow0, x0;
        if (WC == 32) { const unsigned q = fdiv((unsigned)m0, p.dW_); grow0 = (int)q; x0 = m0 - (int)q * p.W; }
        else { grow0 = m0 / WC; x0 = 0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                  // dY chunk: [32 px][32 n]
            const int m = m0 + j * 8 + ppx;
            const float* g_ = m < p.M ? p.dy + (long long)m * 32 + pc4 : zero;
            lds_dma16_w(g_, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(j * 1024))));
        }
#pragma unroll
        for (int j = 0; j < X_PIECES; ++j) {                           // the chunk's input halo: [R][3 rows][WC + 2][32 c]
            const int hp = j * 8 + ppx;
            const int r = hp / (3 * WP), rem = hp - r * (3 * WP);
            const int dyi = rem / WP, xx = rem - dyi * WP;
            const unsigned grow = (unsigned)(grow0 + r);
            const unsigned b = fdiv(grow, p.dH_);
            const int yy = (int)(grow - b * (unsigned)p.H) + dyi - 1;
            const int xc = x0 + xx - 1;
            const bool ok = hp < HALO_PX && (int)b < p.B && (unsigned)yy < (unsigned)p.H && (unsigned)xc < (unsigned)p.W;
            const float* g_ = ok ? p.x + (((long long)b * p.H + yy) * p.W + xc) * 32 + pc4 : zero;
            lds_dma16_w(g_, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(32 * 32 * 4 + j * 1024))));
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = p.bias_slab != nullptr;
    const int fh = lane >> 5, l31 = lane & 31;
    int hb_off[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = 2 * kk + fh;
        const int r = k / WC, xl = k - r * WC;
        hb_off[kk] = 32 * 32 + (r * 3 * WP + xl) * 32 + l31;           // tap (dy, dx) adds ((dy + 1) * WP + dx + 1) * 32
    }
    {
        const int NN = p.N, cxx = p.cx;
        asm volatile("" ::"s"(p.slab), "s"(p.bias_slab), "s"(NN), "s"(cxx));
    }
    int g = g_begin + wave;                                            // this wave's chunks: g_begin + wave, + 4, ...
    if (g < g_end) issue(0, g);
    for (int it = 0; g < g_end; g += 4, ++it) {
        const int stage = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's own pieces (its buffers are private)
        if (g + 4 < g_end) issue(stage ^ 1, g + 4);
        const float* S = wbuf + stage * STAGE;
        float a[2], b[2][9];
        auto load_pair = [&](int slot, int kk) {
            a[slot] = S[fh * 32 + l31 + kk * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) b[slot][t] = S[hb_off[kk] + ((t / 3) * WP + (t % 3)) * 32];
        };
        load_pair(0, 0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < 16) load_pair(cur ^ 1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur], b[cur][t], acc[t], 0, 0, 0);
            if (do_bias) bsum += a[cur];
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // the four waves' sums meet in LDS, tap by tap (fixed order: wave 0..3), and leave as slab[split][n][tap][c]
    __syncthreads();                                                   // every wave is done with its buffers
    float* red = smem;                                                 // [4 waves][32 n][32 c]
    float* outp = p.slab + (long long)split * 32 * 9 * 32;
#pragma unroll                                                         // (a rolled loop would index acc[] dynamically = scratch memory)
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + l31] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256, n = e >> 5, c = e & 31;
            const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
            outp[((long long)n * 9 + t) * 32 + c] = v;
        }
        __syncthreads();
    }
    if (do_bias) {
        const float tsum = bsum + __shfl_xor(bsum, 32, 64);
        if (lane < 32) red[wave * 32 + lane] = tsum;
        __syncthreads();
        if (tid < 32) p.bias_slab[(long long)split * 32 + tid] = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
    }
}

// plan of the narrow halo kernel: <= 256 workgroups of 4 waves, at least 2 chunks per wave
static bool wgrad_halo32_plan(int kind, int B, int H, int W, int cx, int N, int& wc, int& splits, int& cps, int& n_chunks) {
#ifdef DDK_TUNING
    if (getenv("DDK_NO_WGRAD_HALO32")) return false;
#endif
    if (kind != DDK_CONV3X3_S1 || N != 32 || cx != 32) return false;
    if (W <= 32) { if (W < 8 || 32 % W) return false; wc = W; }
    else { if (W % 32) return false; wc = 32; }
    const long long M = (long long)B * H * W;
    if (M % 32) return false;
    if (W < 32 && ((long long)B * H) % (32 / W)) return false;
    n_chunks = (int)(M / 32);
    if (n_chunks < 256) return false;
    long long s = n_chunks / 8;
    if (s > 256) s = 256;
    cps = (int)ceil_div(n_chunks, s);
    splits = (int)ceil_div(n_chunks, cps);
    return true;
}

template <int WC>
static int launch_wgrad_halo32(const WgHaloParams& p, hipStream_t st) {
    hipLaunchKernelGGL((wgrad3x3_halo32_kernel<WC>), dim3((unsigned)p.splits), dim3(256), WgHalo32<WC>::lds_bytes, st, p);
    return check_launch("wgrad3x3_halo32_kernel");
}

// > 64 KB of dynamic LDS needs the attribute, once per device (ensure_device_init, core.hip)
int wgrad_init_device() {
#define WG_ATTR(WC)                                                                                                               \
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_halo_kernel<WC>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)wgrad_halo_lds_bytes<WC>()));
    WG_ATTR(32) WG_ATTR(16) WG_ATTR(8) WG_ATTR(4)
#undef WG_ATTR
#define WG_ATTR32(WC)                                                                                                               \
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_halo32_kernel<WC>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)WgHalo32<WC>::lds_bytes));
    WG_ATTR32(32) WG_ATTR32(16) WG_ATTR32(8)
#undef WG_ATTR32
    return DDK_OK;
}

}  // namespace ddk

using namespace ddk;

extern "C" {

size_t ddk_conv_wgrad_workspace_bytes(int kind, int B, int H, int W, int cx, int N) {
    int Hm, Wm, stride, tapmode, ntaps;
    if (!wgrad_geometry(kind, H, W, Hm, Wm, stride, tapmode, ntaps)) return 0;
    int wc, hs, cps, nch;
    // + [splits][N] for the column sums of dy (ddk_conv_wgrad_bias)
    if (wgrad_halo32_plan(kind, B, H, W, cx, N, wc, hs, cps, nch)) return (size_t)hs * N * (9 * cx + 1) * sizeof(float);
    if (wgrad_halo_plan(kind, B, H, W, cx, N, wc, hs, cps, nch)) return (size_t)hs * N * (9 * cx + 1) * sizeof(float);
    const WgChoice c = wgrad_choice(N, cx, ntaps, (long long)B * Hm * Wm);
    return (size_t)c.splits * N * (ntaps * cx + 1) * sizeof(float);
}

/* grad_w[(n*cw + c_off + c)*taps + tap] += sum_m dy[m][n] x[pix(m)+tap][c],  c < c_real <= cx.
 * x: [B][H][W][cx] (the conv's input, cx % 32 == 0), dy: [B][Hm][Wm][N] (N % 32 == 0).
 * For a conv over a channel concat call once per source with that source's c_off; cw = total input channels of
 * the weight tensor.  For ConvTranspose2d(k4,s2,p1) call with kind = DDK_CONV4X4_S2, x = dY, dy = X: the result
 * is laid out like its (I,O,4,4) weight. */
int ddk_conv_wgrad(int kind, const float* x, const float* dy, float* grad_w, int B, int H, int W, int cx, int c_real, int cw,
                   int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    return ddk_conv_wgrad_bias(kind, x, dy, grad_w, nullptr, B, H, W, cx, c_real, cw, c_off, N, workspace, workspace_bytes, s);
}

/* ddk_conv_wgrad that also accumulates the bias gradient when grad_b != null: grad_b[n] += sum_m dy[m][n], computed by the same
 * launches (the column sums ride on the weight-gradient GEMM as dY^T 1; no separate pass over dy). */
static int conv_wgrad_impl(int kind, const float* x, const float* dy, float* grad_w, float* grad_b, int B, int H, int W, int cx, int c_real,
                           int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s, ddk_wgrad_reduce_job* job_out);

int ddk_conv_wgrad_bias(int kind, const float* x, const float* dy, float* grad_w, float* grad_b, int B, int H, int W, int cx, int c_real,
                        int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    return conv_wgrad_impl(kind, x, dy, grad_w, grad_b, B, H, W, cx, c_real, cw, c_off, N, workspace, workspace_bytes, s, nullptr);
}

/* The same without its reduce launch: the slabs stay in `workspace` (which must then be this call's own until the reduce has run) and
 * *job_out describes the reduce; ddk_wgrad_reduce_jobs runs the reduces of many such calls in one launch. */
int ddk_conv_wgrad_defer(int kind, const float* x, const float* dy, float* grad_w, float* grad_b, int B, int H, int W, int cx, int c_real,
                         int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_wgrad_reduce_job* job_out, ddk_stream_t s) {
    DDK_REQUIRE(job_out, "conv_wgrad_defer: job_out");
    return conv_wgrad_impl(kind, x, dy, grad_w, grad_b, B, H, W, cx, c_real, cw, c_off, N, workspace, workspace_bytes, s, job_out);
}

int ddk_wgrad_reduce_jobs(const ddk_wgrad_reduce_job* jobs, int n, ddk_stream_t s) {
    DDK_REQUIRE(jobs && n > 0, "wgrad_reduce_jobs: arguments");
    for (int k0 = 0; k0 < n; k0 += WG_JOBS_PER_LAUNCH) {
        WgradJobPack pk{};
        pk.n = n - k0 < WG_JOBS_PER_LAUNCH ? n - k0 : WG_JOBS_PER_LAUNCH;
        long long blocks = 0;
        for (int k = 0; k < pk.n; ++k) {
            ddk_wgrad_reduce_job j = jobs[k0 + k];
            DDK_REQUIRE(j.slab && j.grad && j.splits > 0 && j.N > 0 && j.ntaps > 0 && j.cx > 0 && j.cx % 32 == 0 && j.c_real > 0 && j.c_real <= j.cx &&
                            j.c_off >= 0 && j.c_off + j.c_real <= j.cw && (!j.grad_b || j.bias_slab),
                        "wgrad_reduce_jobs: job");
            j.block0 = blocks;
            blocks += (long long)j.N * (j.cx / wg_cxb(j.cx, j.splits)) * (j.ntaps + (j.grad_b ? 1 : 0));
            pk.j[k] = j;
        }
        DDK_REQUIRE(blocks < (1LL << 31), "wgrad_reduce_jobs: too many blocks");
        hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), pk);
        DDK_TRY(check_launch("wgrad_reduce_jobs_kernel"));
    }
    return DDK_OK;
}

static int conv_wgrad_impl(int kind, const float* x, const float* dy, float* grad_w, float* grad_b, int B, int H, int W, int cx, int c_real,
                           int cw, int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s, ddk_wgrad_reduce_job* job_out) {
    int Hm, Wm, stride, tapmode, ntaps;
    DDK_REQUIRE(wgrad_geometry(kind, H, W, Hm, Wm, stride, tapmode, ntaps), "conv_wgrad: kind");
    DDK_REQUIRE(x && dy && grad_w && workspace, "conv_wgrad: null pointer");
    DDK_REQUIRE(B > 0 && cx > 0 && cx % 32 == 0 && N > 0 && N % 32 == 0, "conv_wgrad: cx and N must be multiples of 32");
    DDK_REQUIRE(c_real > 0 && c_real <= cx && c_off >= 0 && c_off + c_real <= cw, "conv_wgrad: channel window");
    DDK_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(workspace), "conv_wgrad: alignment");
    const long long M = (long long)B * Hm * Wm;
    DDK_REQUIRE(M < (1LL << 31) && (long long)B * H * W * cx < (1LL << 31), "conv_wgrad: tensor too large");
    hipStream_t st = as_stream(s);
    DDK_TRY(ensure_device_init());
    {
        int wc, hs, cps, nch;
        const bool narrow = wgrad_halo32_plan(kind, B, H, W, cx, N, wc, hs, cps, nch);
        if (narrow || wgrad_halo_plan(kind, B, H, W, cx, N, wc, hs, cps, nch)) {
            const size_t need_h = (size_t)hs * N * (9 * cx + (grad_b ? 1 : 0)) * sizeof(float);
            if (workspace_bytes < need_h) {
                set_error("conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need_h);
                return DDK_ERR_WORKSPACE;
            }
            WgHaloParams hp{};
            hp.x = x; hp.dy = dy; hp.slab = static_cast<float*>(workspace);
            hp.bias_slab = grad_b ? hp.slab + (size_t)hs * N * 9 * cx : nullptr;
            hp.cx = cx; hp.N = N; hp.B = B; hp.H = H; hp.W = W; hp.M = (int)M;
            hp.splits = hs; hp.chunks_per_split = cps; hp.n_chunks = nch;
            hp.dH_ = make_fastdiv((unsigned)H); hp.dW_ = make_fastdiv((unsigned)W);
            int rc;
            if (narrow) rc = wc == 32 ? launch_wgrad_halo32<32>(hp, st) : (wc == 16 ? launch_wgrad_halo32<16>(hp, st) : launch_wgrad_halo32<8>(hp, st));
            else if (wc == 32) rc = launch_wgrad_halo<32>(hp, st);
            else if (wc == 16) rc = launch_wgrad_halo<16>(hp, st);
            else if (wc == 8) rc = launch_wgrad_halo<8>(hp, st);
            else rc = launch_wgrad_halo<4>(hp, st);
            DDK_TRY(rc);
            const long long total = (long long)N * 9 * cx;
            if (job_out) {
                *job_out = ddk_wgrad_reduce_job{static_cast<const float*>(workspace), grad_w, static_cast<const float*>(hp.bias_slab), grad_b, total, 0,
                                                hs, N, 9, cx, c_real, cw, c_off, 0};
                return DDK_OK;
            }
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(N * (cx / wg_cxb(cx, hs))), grad_b ? 10 : 9), dim3(256), 0, st,
                               static_cast<const float*>(workspace), hs, total, grad_w, N, 9, cx, c_real, cw, c_off, total,
                               static_cast<const float*>(hp.bias_slab), grad_b);
            return check_launch("wgrad_reduce_kernel");
        }
    }
    const WgChoice c = wgrad_choice(N, cx, ntaps, M);
    const size_t need = (size_t)c.splits * N * (ntaps * cx + (grad_b ? 1 : 0)) * sizeof(float);
    if (workspace_bytes < need) {
        set_error("conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
        return DDK_ERR_WORKSPACE;
    }
    WgradParams p{};
    p.x = x; p.dy = dy; p.slab = static_cast<float*>(workspace);
    p.bias_slab = grad_b ? p.slab + (size_t)c.splits * N * ntaps * cx : nullptr;
    p.cx = cx; p.N = N; p.H = H; p.W = W; p.Hm = Hm; p.Wm = Wm; p.M = (int)M;
    p.in_stride = stride; p.tapmode = tapmode; p.ntaps = ntaps;
    p.splits = c.splits; p.rows_per_split = c.rows_per;
    p.dW_ = make_fastdiv((unsigned)Wm); p.dH_ = make_fastdiv((unsigned)Hm);
    const int tn = (int)ceil_div(N, c.bn), tc = (int)ceil_div(cx, c.bc);
    int rc;
#define WG(BN_, BC_, WN_, WC_) rc = launch_wgrad<BN_, BC_, WN_, WC_>(p, tn, tc, st)
    if (c.bn == 128 && c.bc == 128) WG(128, 128, 2, 2);
    else if (c.bn == 128 && c.bc == 64) WG(128, 64, 2, 2);
    else if (c.bn == 128 && c.bc == 32) WG(128, 32, 4, 1);
    else if (c.bn == 64 && c.bc == 128) WG(64, 128, 2, 2);
    else if (c.bn == 64 && c.bc == 64) WG(64, 64, 2, 2);
    else if (c.bn == 64 && c.bc == 32) WG(64, 32, 2, 1);
    else if (c.bn == 32 && c.bc == 128) WG(32, 128, 1, 4);
    else if (c.bn == 32 && c.bc == 64) WG(32, 64, 1, 2);
    else WG(32, 32, 1, 1);
#undef WG
    DDK_TRY(rc);
    const long long total = (long long)N * ntaps * cx;
    if (job_out) {
        *job_out = ddk_wgrad_reduce_job{static_cast<const float*>(workspace), grad_w, static_cast<const float*>(p.bias_slab), grad_b, total, 0,
                                        c.splits, N, ntaps, cx, c_real, cw, c_off, 0};
        return DDK_OK;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(N * (cx / wg_cxb(cx, c.splits))), (unsigned)(ntaps + (grad_b ? 1 : 0))), dim3(256), 0, st,
                       static_cast<const float*>(workspace), c.splits, total, grad_w, N, ntaps, cx, c_real, cw, c_off, total,
                       static_cast<const float*>(p.bias_slab), grad_b);
    return check_launch("wgrad_reduce_kernel");
}

/* grad_b[n] (+)= sum_m dy[m][n]; workspace >= 256*N floats */
int ddk_bias_grad(const float* dy, float* grad_b, long long M, int N, int accumulate, void* workspace, size_t workspace_bytes,
                  ddk_stream_t s) {
    DDK_REQUIRE(dy && grad_b && workspace && M > 0 && N > 0, "bias_grad: arguments");
    DDK_REQUIRE(N % 4 == 0 && N <= 1024 && aligned16(dy) && aligned16(workspace), "bias_grad: N % 4 == 0, N <= 1024, 16-byte aligned dy");
    // ~256 workgroups, each at least 64 rows
    long long parts = ceil_div(M, 64);
    if (parts > 256) parts = 256;
    DDK_REQUIRE(workspace_bytes >= (size_t)parts * N * sizeof(float), "bias_grad: workspace too small (need 256*N floats)");
    const long long rows_per = ceil_div(M, parts);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)parts), dim3(256), 0, as_stream(s), dy, part, M, N, rows_per);
    DDK_TRY(check_launch("colsum_partial_kernel"));
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(256), 0, as_stream(s), part, (int)parts, grad_b, N,
                       accumulate);
    return check_launch("colsum_final_kernel");
}

/* Conv2d weight OIHW -> input-gradient operand [I_pad][KH*KW][O_pad] with flipped taps */
int ddk_pack_conv_weight_dgrad(const float* w_oihw, float* dst, int O, int I, int KH, int KW, int i_pad, int o_pad, ddk_stream_t s) {
    DDK_REQUIRE(w_oihw && dst && O > 0 && I > 0 && KH > 0 && KW > 0 && i_pad >= I && o_pad >= O, "pack_conv_weight_dgrad: arguments");
    const long long total = (long long)i_pad * KH * KW * o_pad;
    DDK_REQUIRE(total < (1LL << 31), "pack_conv_weight_dgrad: 2^31 elements or more");
    const int blocks = (int)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(s), w_oihw, dst, O, I, KH * KW, i_pad, o_pad, total);
    return check_launch("pack_dgrad_kernel");
}

/* [B][H][W][C] -> [B][Ho][Wo][C] with the values on the even grid, zeros elsewhere */
int ddk_zero_stuff2(const float* in, float* out, int B, int H, int W, int Ho, int Wo, int C, ddk_stream_t s) {
    DDK_REQUIRE(in && out && B > 0 && C % 4 == 0 && Ho >= 2 * H - 1 && Ho <= 2 * H && Wo >= 2 * W - 1 && Wo <= 2 * W,
                "zero_stuff2: arguments");
    DDK_REQUIRE(aligned16(in) && aligned16(out), "zero_stuff2: alignment");
    const long long total4 = (long long)B * Ho * Wo * (C / 4);
    const int blocks = (int)(ceil_div(total4, 256) < 4096 ? ceil_div(total4, 256) : 4096);
    hipLaunchKernelGGL(zero_stuff2_kernel, dim3(blocks), dim3(256), 0, as_stream(s), in, out, H, W, Ho, Wo, C, total4);
    return check_launch("zero_stuff2_kernel");
}
}
