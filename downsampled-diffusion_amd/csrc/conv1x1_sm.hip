// conv1x1_sm.hip -- 1x1 conv (+ bias + residual) on SMALL maps: M = B*H*W <= a few thousand pixels, K = 128 .. 512 input channels.
//
// Where: to_out of the attention blocks (blocks.py:124 + the Residual of :13-14) and res_conv (blocks.py:103) on the 4x4 and 8x8 maps
// -- 33 .. 270 MFLOP each.  On the im2col tile kernel (64x64 tiles, a 4-stage ring built for long contractions) they were 32 .. 128
// workgroups and 8 .. 11 us apiece (+ a 5 us slab reduce where the chooser split K): seven launches, 68 us of a 1350 us step, all of
// it latency -- the launch, a ring prologue, a k loop of 4 .. 16 steps behind one barrier each, an epilogue through LDS.
//
// Here a workgroup owns a 32-pixel x 32-channel tile (128 .. 768 workgroups) and its four waves split the CONTRACTION: wave w DMAs
// rows [w K/4, (w+1) K/4) of both operands into its own LDS region in one burst (4 .. 16 pieces each), waits for its OWN pieces only
// (no workgroup barrier in front of the matrix work), and multiplies K/8 MFMAs; the four 32x32 partial tiles meet in LDS behind the
// single barrier of the kernel and leave with bias and residual as float4 rows.  No ring, no k loop across barriers, no slabs.
// LDS rows are XOR-swizzled at 16-byte granularity (the source quad a DMA lane fetches is chosen accordingly) so that the 16 lanes of a
// ds_read_b128 group hit 16 different slots: rows of 128 B alternate bank halves ((row >> 1) & 7), wider rows all start on bank 0
// (row & 15).
#include "conv_common.h"

namespace ddk {

struct SmParams {
    const float* src0;
    const float* src1;
    int c0, c1;            // K = c0 + c1
    const float* w;        // [N][K]
    const float* bias;     // [N] or null
    const float* resid;    // [M][N] or null
    float* out;            // [M][N]
    int M, N, K;
    // channel LayerNorm folded into the conv (blocks.py:57-60 in front of to_qkv): w is W o g, ln_c1 = W g, ln_c2 = W b; with
    // r = 1 / (std + eps) of the pixel, out = r (w x) - r mean ln_c1 + ln_c2 (the arithmetic of the tile kernel's folded form)
    const float* ln_c1;    // null: no LayerNorm
    const float* ln_c2;
    float ln_eps;
};

template <int KQ, bool LN>         // K / 4: 32, 64 or 128 floats per row and wave; LN: folded channel LayerNorm
__global__ __launch_bounds__(256) void conv1x1_sm_kernel(const SmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QPR = KQ / 4;                         // 16-byte quads per row
    constexpr int PIECES = 32 * KQ / 256;               // 1-KB DMA pieces per operand and wave
    constexpr int WAVE_FLOATS = 2 * 32 * KQ;            // A rows then B rows
    constexpr int RED_PITCH = 36;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    auto swz = [](int row) { return QPR == 8 ? (row >> 1) & 7 : row & 15; };

    // ---- this wave's quarter of the contraction: k in [wid KQ, (wid + 1) KQ) -- entirely inside one source (host: c0 % KQ == 0)
    const int k0 = wid * KQ;
    const bool first = k0 < p.c0;                       // wave-uniform
    const float* asrc = first ? p.src0 : p.src1;
    const int acs = first ? p.c0 : p.c1, ak0 = first ? k0 : k0 - p.c0;
    const unsigned lds_a = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + (unsigned)(wid * WAVE_FLOATS * 4);
    const unsigned lds_b = lds_a + 32 * KQ * 4;
#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) {
        const int s = pc * 64 + lane;                   // 16-byte slot of the [32][KQ] block
        const int row = s / QPR, q = (s % QPR) ^ swz(row);
        lds_dma16(asrc + (long long)(m0 + row) * acs + ak0 + q * 4, lds_a + (unsigned)(pc * 1024));
        lds_dma16(p.w + (long long)(n0 + row) * p.K + k0 + q * 4, lds_b + (unsigned)(pc * 1024));
    }
    // the epilogue's operands of this thread: row tid / 8, channel quad tid % 8
    const int er = tid >> 3, ec = (tid & 7) * 4;
    const long long eo = (long long)(m0 + er) * p.N + n0 + ec;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), rr = bb;
    if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + n0 + ec);
    if (p.resid) rr = *reinterpret_cast<const float4*>(p.resid + eo);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int l31 = lane & 31, kh = lane >> 5;
    const float4* A4 = reinterpret_cast<const float4*>(smem + wid * WAVE_FLOATS) + l31 * QPR;
    const float4* B4 = A4 + 32 * QPR;
    const int sw = swz(l31);
    wait_vmcnt<0>();                                    // this wave's own pieces (and the epilogue operands): no barrier needed
    float ln_s = 0.f, ln_q = 0.f;                       // LN: sum and sum of squares of this lane's share of pixel row l31
#pragma unroll
    for (int j = 0; j < KQ / 8; ++j) {
        const float4 a = A4[(2 * j + kh) ^ sw], b = B4[(2 * j + kh) ^ sw];
        if (LN) {
            ln_s += (a.x + a.y) + (a.z + a.w);
            ln_q += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    // ---- the four partial tiles meet behind the operand regions: red[wave][32 rows][36]
    float* red = smem + 4 * WAVE_FLOATS + wid * 32 * RED_PITCH;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * kh) * RED_PITCH + l31] = acc[r];
    float* stat = smem + 4 * WAVE_FLOATS + 4 * 32 * RED_PITCH;     // [wave][32 rows][2]
    if (LN) {
        ln_s += __shfl_xor(ln_s, 32, 64);
        ln_q += __shfl_xor(ln_q, 32, 64);
        if (kh == 0) { stat[(wid * 32 + l31) * 2] = ln_s; stat[(wid * 32 + l31) * 2 + 1] = ln_q; }
    }
    __syncthreads();
    const float* rp = smem + 4 * WAVE_FLOATS + er * RED_PITCH + ec;
    float4 v = *reinterpret_cast<const float4*>(rp);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(rp + w * 32 * RED_PITCH);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (LN) {      // biased variance, eps added to the std (blocks.py:57-60); one pass, as the tile kernel's folded form
        float sx = 0.f, sq = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sx += stat[(w * 32 + er) * 2]; sq += stat[(w * 32 + er) * 2 + 1]; }
        const float inv_c = 1.0f / (float)p.K;
        const float mean = sx * inv_c;
        const float var = fmaxf(sq * inv_c - mean * mean, 0.f);
        const float r = 1.0f / (sqrtf(var) + p.ln_eps), rm = r * mean;
        const float4 l1 = *reinterpret_cast<const float4*>(p.ln_c1 + n0 + ec), l2 = *reinterpret_cast<const float4*>(p.ln_c2 + n0 + ec);
        v.x = r * v.x - rm * l1.x + l2.x; v.y = r * v.y - rm * l1.y + l2.y;
        v.z = r * v.z - rm * l1.z + l2.z; v.w = r * v.w - rm * l1.w + l2.w;
    }
    v.x += bb.x + rr.x; v.y += bb.y + rr.y; v.z += bb.z + rr.z; v.w += bb.w + rr.w;
    *reinterpret_cast<float4*>(p.out + eo) = v;
}

static size_t sm_lds_bytes(int KQ) { return ((size_t)4 * 2 * 32 * KQ + 4 * 32 * 36 + 4 * 32 * 2) * sizeof(float); }

bool conv1x1_sm_ok(long long M, int c0, int c1, int N) {
    const int K = c0 + c1;
    if (M <= 0 || M % 32 || N <= 0 || N % 32 || (K != 128 && K != 256 && K != 512)) return false;
    const int KQ = K / 4;
    if (c0 % KQ || (c1 && c1 % KQ)) return false;
    const long long tiles = (M / 32) * (N / 32);
    // small maps only: every tile re-reads its operands from L2, which the 64x64 tile kernel's reuse beats once the map is large
    // (measured at batch 32: 512 -> 128 on 16x16, M K = 4 M: 21.2 vs 16.6 us; 128 -> 128 on 16x16, M K = 1 M: 7.8 vs 9.2 us)
    return tiles >= 64 && tiles <= 1024 && M * K <= (1LL << 20);
}

int conv1x1_sm_init_device() {
#define SM_ATTR(KQ, LN) \
    DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_sm_kernel<KQ, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_lds_bytes(KQ)));
    SM_ATTR(128, false) SM_ATTR(128, true) SM_ATTR(64, false) SM_ATTR(64, true)
#undef SM_ATTR
    return DDK_OK;
}

int conv1x1_sm(const float* src0, int c0, const float* src1, int c1, const float* w, const float* bias, const float* resid, float* out,
               long long M, int N, const ConvLnFold* ln, hipStream_t st) {
    DDK_REQUIRE(src0 && w && out && conv1x1_sm_ok(M, c0, c1, N) && (c1 == 0 || src1), "conv1x1_sm: arguments");
    DDK_REQUIRE(aligned16(src0) && aligned16(src1) && aligned16(w) && aligned16(bias) && aligned16(resid) && aligned16(out), "conv1x1_sm: alignment");
    DDK_REQUIRE(!ln || (ln->c1 && ln->c2 && aligned16(ln->c1) && aligned16(ln->c2) && c1 == 0), "conv1x1_sm: LayerNorm folding vectors (single source)");
    DDK_TRY(ensure_device_init());
    SmParams p{src0, src1, c0, c1, w, bias, resid, out, (int)M, N, c0 + c1, ln ? ln->c1 : nullptr, ln ? ln->c2 : nullptr, ln ? ln->eps : 0.f};
    const dim3 grid((unsigned)(M / 32), (unsigned)(N / 32));
    const int KQ = p.K / 4;
#define SM_LAUNCH(KQ_) \
    do { \
        if (ln) hipLaunchKernelGGL((conv1x1_sm_kernel<KQ_, true>), grid, dim3(256), sm_lds_bytes(KQ_), st, p); \
        else hipLaunchKernelGGL((conv1x1_sm_kernel<KQ_, false>), grid, dim3(256), sm_lds_bytes(KQ_), st, p); \
    } while (0)
    if (KQ == 32) SM_LAUNCH(32);
    else if (KQ == 64) SM_LAUNCH(64);
    else SM_LAUNCH(128);
#undef SM_LAUNCH
    return check_launch("conv1x1_sm_kernel");
}

}  // namespace ddk
