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
};

constexpr int LOC_PP = 36;   // pitch (floats) of a partial-accumulator row: 4 rows apart = 16 banks apart

__host__ __device__ static inline size_t local_lds_bytes(int MT, int cin) {
    const size_t a = (size_t)(MT + 1) * (cin + 4) * 4;
    const size_t p = (size_t)8 * MT * LOC_PP * 4;
    return (a > p ? a : p) + 256;
}

template <int MT>
__global__ __launch_bounds__(512) void conv3x3_gn_local_kernel(const LocalParams p) {
    extern __shared__ __align__(16) float lds[];
    constexpr int MB = MT / 16;
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

    // ---- the image: [MT rows][cin] into LDS, row MT = zeros
    {
        const int q4 = cin >> 2;
        const long long row0 = (long long)b * MT;
        for (int i = tid; i < MT * q4; i += 512) {
            const int row = i / q4, c = (i - row * q4) << 2;
            const float4 v = c < p.c0 ? *reinterpret_cast<const float4*>(p.src0 + (row0 + row) * p.c0 + c)
                                      : *reinterpret_cast<const float4*>(p.src1 + (row0 + row) * p.c1 + (c - p.c0));
            *reinterpret_cast<float4*>(lds + row * pitch + c) = v;
        }
        for (int i = tid; i < q4; i += 512) *reinterpret_cast<float4*>(lds + MT * pitch + (i << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // ---- k loop: this wave's units u = wave, wave + 8, ...
    f32x4 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int py[MB], px[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int r = i * 16 + m;
        py[i] = r / p.W;
        px[i] = r - py[i] * p.W;
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
            const int srow = ok ? yy * p.W + xx : MT;
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
    while (ctap < 9) {                // scalar loop control; loads past the end re-read the last unit (no load under a condition)
        compute(bA);
        load_b(bA);
        if (ctap < 9) compute(bB);
        load_b(bB);
        if (ctap < 9) compute(bC);
        load_b(bC);
    }

    // ---- the 8 waves' partial tiles meet in LDS (the image is no longer needed)
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
    const int col = tid & 31, row = tid >> 5;
    const int c = n0 + col;
    float v[MB];
    const float cb = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        float s = lds[(row + 16 * i) * LOC_PP + col];
#pragma unroll
        for (int w = 1; w < 8; ++w) s += lds[(w * MT + row + 16 * i) * LOC_PP + col];
        v[i] = s + cb;
    }

    // ---- GroupNorm statistics of (this image, group of column col): two passes over the registers, like torch's
    //      native_group_norm.  Wave level: lanes that share the group (xor masks below cpg, and 32 = the other row of the
    //      wave); workgroup level: the 8 waves' sums in fixed order.
    const int gl = col / p.cpg;                 // group within the tile: 0 .. 32/cpg - 1
    auto group_sum = [&](float s) {
        for (int o = 1; o < p.cpg; o <<= 1) s += __shfl_xor(s, o, 64);
        s += __shfl_xor(s, 32, 64);
        __syncthreads();                        // red free again
        if ((lane & 32) == 0 && (col & (p.cpg - 1)) == 0) red[wave * 4 + gl] = s;
        __syncthreads();
        float t = red[gl];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += red[w * 4 + gl];
        return t;
    };
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i) s += v[i];
    const float inv_n = 1.0f / (float)(MT * p.cpg);
    const float mean = group_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float var = group_sum(q) * inv_n;
    const float rstd = 1.0f / sqrtf(var + p.eps);

    const float ga = p.gamma[c], be = p.beta[c];
    float sh = 0.f;
    if (p.temb) {
        const long long tr = p.temb_rows ? p.temb_rows[b] : b;
        sh = p.temb[tr * p.temb_stride + c];
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const long long o = ((long long)b * MT + row + 16 * i) * p.N + c;
        float y = mish_f((v[i] - mean) * rstd * ga + be) + sh;
        if (p.addend) y += p.addend[o];
        p.out[o] = y;
    }
}

// dst[n tile][tap][chunk][n block][k half][lane = kq * 16 + n][j] = w[o = 32 nt + 16 nb + n][i = 32 chunk + 8 kq + 4 half + j][tap]
__global__ __launch_bounds__(256) void pack_conv_weight_local_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I,
                                                                     int i_pad, long long total) {
    const int nch = i_pad >> 5;
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63), half = (int)((idx >> 8) & 1), nb = (int)((idx >> 9) & 1);
        long long r = idx >> 10;
        const int chunk = (int)(r % nch); r /= nch;
        const int tap = (int)(r % 9);
        const int nt = (int)(r / 9);
        const int o = nt * 32 + nb * 16 + (lane & 15);
        const int i = chunk * 32 + (lane >> 4) * 8 + half * 4 + j;
        dst[idx] = i < I ? w[((long long)o * I + i) * 9 + tap] : 0.f;
    }
}

bool conv_gn_local_ok(int H, int W, int cin, int c0, int N, int groups) {
    const int HW = H * W;
    if (HW != 16 && HW != 64) return false;
    if (cin % 32 || c0 % 4 || (cin - c0) % 4 || N % 32 || N % groups) return false;
    const int cpg = N / groups;
    if (cpg != 8 && cpg != 16 && cpg != 32) return false;
    return local_lds_bytes(HW, cin) <= 160 * 1024;
}

int conv_gn_local_init_device() {
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_gn_local_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    return DDK_OK;
}

int conv_gn_local(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* gamma,
                  const float* beta, const float* temb, int temb_stride, const long long* temb_rows, const float* addend, float* out,
                  int B, int H, int W, int N, int groups, float eps, hipStream_t st) {
    DDK_REQUIRE(src0 && w && gamma && beta && out, "conv_gn_local: null pointer");
    DDK_REQUIRE(B > 0 && groups > 0, "conv_gn_local: B and groups must be positive");
    DDK_REQUIRE(c1 == 0 || src1, "conv_gn_local: second source missing");
    DDK_REQUIRE(conv_gn_local_ok(H, W, c0 + c1, c0, N, groups), "conv_gn_local: shape not eligible (needs H*W in {16, 64}, "
                "cin % 32 == 0, N % 32 == 0, channels per group in {8, 16, 32})");
    DDK_REQUIRE(aligned16(src0) && aligned16(src1) && aligned16(w), "conv_gn_local: sources and weights must be 16-byte aligned");
    DDK_TRY(ensure_device_init());
    LocalParams p{src0, src1, c0, c1, w, bias, gamma, beta, temb, temb_stride, temb_rows, addend, out, H, W, N, N / groups, eps};
    const int HW = H * W;
    const size_t ldsb = local_lds_bytes(HW, c0 + c1);
    const dim3 grid((unsigned)((long long)B * (N / 32)));
    if (HW == 16)
        hipLaunchKernelGGL(conv3x3_gn_local_kernel<16>, grid, dim3(512), ldsb, st, p);
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
                       dst, O, I, i_pad, total);
    return check_launch("pack_conv_weight_local_kernel");
}

int ddk_conv3x3_gn_mish_ok(int H, int W, int cin, int c0, int N, int groups) { return ddk::conv_gn_local_ok(H, W, cin, c0, N, groups) ? 1 : 0; }

int ddk_conv3x3_gn_mish(const float* src0, int c0, const float* src1, int c1, const float* weight, const float* bias,
                        const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend, float* out,
                        int B, int H, int W, int N, int groups, float eps, ddk_stream_t s) {
    return ddk::conv_gn_local(src0, c0, src1, c1, weight, bias, gamma, beta, temb, temb_stride, nullptr, addend, out, B, H, W, N, groups,
                              eps, ddk::as_stream(s));
}

}  // extern "C"
