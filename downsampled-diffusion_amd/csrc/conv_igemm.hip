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
    signed char dy[4][9];
    signed char dx[4][9];
};

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
        const int dy = p.dy[phase][tap], dx = p.dx[phase][tap];
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
    const long long target = 200;
    TileId order_wide[] = {T128x128, T128x64, T64x64};
    TileId order_n64[] = {T128x64, T64x64};
    TileId order_n32[] = {T128x32, T64x32};
    TileId* order;
    int n_order;
    if (N % 64 != 0) { order = order_n32; n_order = 2; }
    else if (N % 128 != 0 && N < 128) { order = order_n64; n_order = 2; }
    else { order = order_wide; n_order = 3; }
    Choice c{order[n_order - 1], 1, kiters};
    long long tiles = 0;
    for (int i = 0; i < n_order; ++i) {
        int bm, bn;
        tile_dims(order[i], bm, bn);
        tiles = ceil_div(M, bm) * ceil_div(N, bn) * nphase;
        if (tiles >= target || i == n_order - 1) { c.tile = order[i]; break; }
    }
    if (tiles < target) {
        long long want = ceil_div(256, tiles);
        long long max_by_k = kiters / 4 > 0 ? kiters / 4 : 1;
        long long s = want < max_by_k ? want : max_by_k;
        if (s > 16) s = 16;
        if (s < 1) s = 1;
        c.kps = (int)ceil_div(kiters, s);
        c.splits = (int)ceil_div(kiters, c.kps);
    }
    return c;
}

template <int BM, int BN, int WM, int WN>
static int launch_tile(const IgemmParams& p, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * LDK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.M, BM), (unsigned)ceil_div(p.N, BN), (unsigned)(p.nphase * p.splits));
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
    p.post_mish = a.post_mish;
    p.slab_stride = (long long)a.B * g.Ho * g.Wo * a.N;
    if (a.kind == DDK_CONV3X3_S1 || a.kind == DDK_CONV3X3_S2) {
        for (int t = 0; t < 9; ++t) { p.dy[0][t] = (signed char)(t / 3 - 1); p.dx[0][t] = (signed char)(t % 3 - 1); }
    } else if (a.kind == DDK_CONVT4X4_S2) {
        for (int ph = 0; ph < 4; ++ph)
            for (int t = 0; t < 4; ++t) {
                p.dy[ph][t] = (signed char)((ph >> 1) - (t >> 1));
                p.dx[ph][t] = (signed char)((ph & 1) - (t & 1));
            }
    }
    const Choice c = choose_tile(p.M, p.N, p.nphase, p.kiters);
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
    if (c.splits > 1) {
        const long long n4 = p.slab_stride / 4;
        const int blocks = (int)(ceil_div(n4, 256) < 2048 ? ceil_div(n4, 256) : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a.workspace),
                           c.splits, p.slab_stride, a.bias, a.resid, final_out, n4, p.N, p.post_mish);
        DDK_TRY(check_launch("splitk_reduce_kernel"));
    }
    return DDK_OK;
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

extern "C" int ddk_conv_forward(const ddk_conv_args* a, ddk_stream_t s) {
    if (!a) return ddk::fail_arg("conv: null args");
    return ddk::conv_forward(*a, ddk::as_stream(s));
}
