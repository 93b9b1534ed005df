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

    const int a_off = wn * TN * 32 + (lane & 31) + (lane >> 5) * BN;
    const int b_off = 32 * BN + wc * TC * 32 + (lane & 31) + (lane >> 5) * BC;

    if (n_it > 0) issue(0, m_begin);
    {   // keep SMEM out of the k-loop (see conv_igemm.hip consume_epilogue_args): epilogue-only arguments are consumed here
        const int nt = p.ntaps, NN = p.N, cxx = p.cx;
        asm volatile("" ::"s"(p.slab), "s"(nt), "s"(NN), "s"(cxx));
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
            __builtin_amdgcn_sched_barrier(0);
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

// grad[(n*cw + c_off + c)*ntaps + tap] += sum_s slab[s][n][tap][c]   (only the first c_real channels: the input may be padded)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, long long slab_stride,
                                                           float* __restrict__ grad, int N, int ntaps, int cx, int c_real, int cw,
                                                           int c_off, long long total) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cx);
        long long r = i / cx;
        const int tap = (int)(r % ntaps);
        const int n = (int)(r / ntaps);
        if (c >= c_real) continue;
        float s = slab[i];
        for (int k = 1; k < splits; ++k) s += slab[k * slab_stride + i];
        grad[((long long)n * cw + c_off + c) * ntaps + tap] += s;
    }
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

// dst[i][t][o] = w[o][i][T-1-t] (i < I), zero rows up to i_pad; o padded to o_pad with zeros
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I, int taps,
                                                         int i_pad, int o_pad, long long total) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int o = (int)(idx % o_pad);
        long long r = idx / o_pad;
        const int t = (int)(r % taps);
        const int i = (int)(r / taps);
        dst[idx] = (i < I && o < O) ? w[((long long)o * I + i) * taps + (taps - 1 - t)] : 0.f;
    }
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
    const long long tiles = ceil_div(N, c.bn) * ceil_div(cx, c.bc) * ntaps;
    long long s = ceil_div(512, tiles);
    const long long max_s = M / 256 > 0 ? M / 256 : 1;   // at least 8 k-chunks per split
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
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

template <int BN, int BC, int WN, int WC>
static int launch_wgrad(const WgradParams& p, int ntiles_n, int ntiles_c, hipStream_t st) {
    constexpr size_t lds = 2 * 32 * (size_t)(BN + BC) * sizeof(float);
    dim3 grid((unsigned)ntiles_n, (unsigned)ntiles_c, (unsigned)(p.ntaps * p.splits));
    hipLaunchKernelGGL((wgrad_kernel<BN, BC, WN, WC>), grid, dim3(WN * WC * 64), lds, st, p);
    return check_launch("wgrad_kernel");
}

}  // namespace ddk

using namespace ddk;

extern "C" {

size_t ddk_conv_wgrad_workspace_bytes(int kind, int B, int H, int W, int cx, int N) {
    int Hm, Wm, stride, tapmode, ntaps;
    if (!wgrad_geometry(kind, H, W, Hm, Wm, stride, tapmode, ntaps)) return 0;
    const WgChoice c = wgrad_choice(N, cx, ntaps, (long long)B * Hm * Wm);
    return (size_t)c.splits * N * ntaps * cx * sizeof(float);
}

/* grad_w[(n*cw + c_off + c)*taps + tap] += sum_m dy[m][n] x[pix(m)+tap][c],  c < c_real <= cx.
 * x: [B][H][W][cx] (the conv's input, cx % 32 == 0), dy: [B][Hm][Wm][N] (N % 32 == 0).
 * For a conv over a channel concat call once per source with that source's c_off; cw = total input channels of
 * the weight tensor.  For ConvTranspose2d(k4,s2,p1) call with kind = DDK_CONV4X4_S2, x = dY, dy = X: the result
 * is laid out like its (I,O,4,4) weight. */
int ddk_conv_wgrad(int kind, const float* x, const float* dy, float* grad_w, int B, int H, int W, int cx, int c_real, int cw,
                   int c_off, int N, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    int Hm, Wm, stride, tapmode, ntaps;
    DDK_REQUIRE(wgrad_geometry(kind, H, W, Hm, Wm, stride, tapmode, ntaps), "conv_wgrad: kind");
    DDK_REQUIRE(x && dy && grad_w && workspace, "conv_wgrad: null pointer");
    DDK_REQUIRE(B > 0 && cx > 0 && cx % 32 == 0 && N > 0 && N % 32 == 0, "conv_wgrad: cx and N must be multiples of 32");
    DDK_REQUIRE(c_real > 0 && c_real <= cx && c_off >= 0 && c_off + c_real <= cw, "conv_wgrad: channel window");
    DDK_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(workspace), "conv_wgrad: alignment");
    const long long M = (long long)B * Hm * Wm;
    DDK_REQUIRE(M < (1LL << 31) && (long long)B * H * W * cx < (1LL << 31), "conv_wgrad: tensor too large");
    const WgChoice c = wgrad_choice(N, cx, ntaps, M);
    const size_t need = (size_t)c.splits * N * ntaps * cx * sizeof(float);
    if (workspace_bytes < need) {
        set_error("conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
        return DDK_ERR_WORKSPACE;
    }
    WgradParams p{};
    p.x = x; p.dy = dy; p.slab = static_cast<float*>(workspace);
    p.cx = cx; p.N = N; p.H = H; p.W = W; p.Hm = Hm; p.Wm = Wm; p.M = (int)M;
    p.in_stride = stride; p.tapmode = tapmode; p.ntaps = ntaps;
    p.splits = c.splits; p.rows_per_split = c.rows_per;
    p.dW_ = make_fastdiv((unsigned)Wm); p.dH_ = make_fastdiv((unsigned)Hm);
    hipStream_t st = as_stream(s);
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
    const int blocks = (int)(ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(workspace), c.splits, total, grad_w,
                       N, ntaps, cx, c_real, cw, c_off, total);
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
