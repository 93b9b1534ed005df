// conv1x1_stream.hip -- 1x1 conv between NARROW tensors (32 or 64 channels on either side) on LARGE maps: the c1 / c4 convs of the
// dDDPM encoder / decoder blocks (convblocks.py:112-130) and their input-gradient convs, 16 384 .. 262 144 pixels at batch 64.
//
// These layers are memory streams: 2 .. 8 KFLOP per pixel against 0.4 .. 0.9 KB of tensor traffic (input, output, and in training the
// residual, the Mish' source and the second Mish output) -- 134 .. 235 MB per launch on the 64x64x64-pixel maps, 21 .. 39 us at the
// ~6 TB/s a streaming kernel reaches.  On the im2col tile kernel (one 64x64 or 128x32 tile per workgroup: a DMA into LDS, a barrier,
// one k step, an epilogue) they ran at 3.3 .. 4.3 TB/s: 4096 workgroups of almost no work each, every one paying the ring prologue.
//
// Here the wave is the unit and nothing goes through LDS:
//   * out^T[ch][px] = W[ch][:] . x[px][:] on v_mfma_f32_16x16x4: the 16 x 16 result tile holds, per lane, FOUR CONSECUTIVE CHANNELS
//     of one pixel -- the epilogue (bias, x Mish'(src), + residual, Mish, second output) is float4 loads and stores on NHWC rows;
//   * the weights (32x64 or 64x32 floats) sit in registers as A fragments for the whole kernel;
//   * the contraction order is free, so lane (pixel p, group g = lane / 16) takes the CONTIGUOUS quarter row x[p][g K/4 .. (g+1) K/4)
//     as float4 loads (k step q multiplies channel g K/4 + q of every group);
//   * a wave walks 16-pixel tiles grid-stride; the loads of tile t+1 and the epilogue operands of tile t are issued before the MFMAs of
//     tile t, 8 waves per CU: ~100 KB in flight per CU.
#include "conv_common.h"

namespace ddk {

typedef float f32x4s __attribute__((ext_vector_type(4)));

struct StreamParams {
    const float* x;          // [M][KIN]
    const float* w;          // [NOUT][KIN]
    const float* bias;       // [NOUT] or null
    const float* dmish_src;  // [M][NOUT] or null: out = (w x + bias) * Mish'(dmish_src)
    const float* resid;      // [M][NOUT] or null
    float* out;              // [M][NOUT]
    float* mish_out;         // [M][NOUT] or null: also receives Mish(out)
    int tiles;               // M / 16
    int pre_mish, post_mish;
};

template <int KIN, int NOUT>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const StreamParams p) {
    constexpr int KQ = KIN / 4;        // channels per lane group = k steps
    constexpr int NB = NOUT / 16;      // 16-channel output blocks
    const int lane = threadIdx.x & 63, px = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), waves = gridDim.x * 4;

    // A fragments: wf[cb][q] = W[cb*16 + (lane & 15)][g*KQ + q]
    float wf[NB][KQ];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int j = 0; j < KQ / 4; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(p.w + (cb * 16 + px) * KIN + g * KQ + j * 4);
            wf[cb][4 * j] = t.x; wf[cb][4 * j + 1] = t.y; wf[cb][4 * j + 2] = t.z; wf[cb][4 * j + 3] = t.w;
        }
    float4 bias4[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
        bias4[cb] = p.bias ? *reinterpret_cast<const float4*>(p.bias + cb * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_x = [&](int tile, float4 (&xr)[KQ / 4]) {
        const float* src = p.x + ((long long)tile * 16 + px) * KIN + g * KQ;
#pragma unroll
        for (int j = 0; j < KQ / 4; ++j) xr[j] = *reinterpret_cast<const float4*>(src + j * 4);
    };

    float4 xc[KQ / 4], xn[KQ / 4];
    int tile = wave;
    if (tile < p.tiles) load_x(tile, xc);
    for (; tile < p.tiles; tile += waves) {
        const int nxt = tile + waves;
        if (nxt < p.tiles) load_x(nxt, xn);
        const long long o = ((long long)tile * 16 + px) * NOUT + g * 4;      // this lane's float4 of block cb sits at o + cb*16
        float4 dm[NB], rs[NB];
        if (p.dmish_src) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) dm[cb] = *reinterpret_cast<const float4*>(p.dmish_src + o + cb * 16);
        }
        if (p.resid) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) rs[cb] = *reinterpret_cast<const float4*>(p.resid + o + cb * 16);
        }
        float xs[KQ];
#pragma unroll
        for (int j = 0; j < KQ / 4; ++j) { xs[4 * j] = xc[j].x; xs[4 * j + 1] = xc[j].y; xs[4 * j + 2] = xc[j].z; xs[4 * j + 3] = xc[j].w; }
        if (p.pre_mish) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) xs[q] = mish_f(xs[q]);
        }
        f32x4s acc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cb][q], xs[q], acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            float4 v = make_float4(acc[cb][0] + bias4[cb].x, acc[cb][1] + bias4[cb].y, acc[cb][2] + bias4[cb].z, acc[cb][3] + bias4[cb].w);
            if (p.dmish_src) {
                v.x *= mish_grad_f(dm[cb].x); v.y *= mish_grad_f(dm[cb].y); v.z *= mish_grad_f(dm[cb].z); v.w *= mish_grad_f(dm[cb].w);
            }
            if (p.resid) { v.x += rs[cb].x; v.y += rs[cb].y; v.z += rs[cb].z; v.w += rs[cb].w; }
            if (p.post_mish) { v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w); }
            *reinterpret_cast<float4*>(p.out + o + cb * 16) = v;
            if (p.mish_out)
                *reinterpret_cast<float4*>(p.mish_out + o + cb * 16) = make_float4(mish_f(v.x), mish_f(v.y), mish_f(v.z), mish_f(v.w));
        }
#pragma unroll
        for (int j = 0; j < KQ / 4; ++j) xc[j] = xn[j];
    }
}

// 32 / 64 channels on both sides, whole 16-pixel tiles, a map large enough to be a stream (below ~16K pixels the tile kernels' few
// hundred workgroups are as good and the small-map kernel covers K >= 128)
bool conv1x1_stream_ok(long long M, int cin, int N) {
    return (cin == 32 || cin == 64) && (N == 32 || N == 64) && M % 16 == 0 && M >= 16384 && M / 16 < (1LL << 31);
}

int conv1x1_stream(const float* x, int cin, const float* w, const float* bias, const float* dmish_src, const float* resid, float* out,
                   float* mish_out, long long M, int N, int pre_mish, int post_mish, hipStream_t st) {
    DDK_REQUIRE(conv1x1_stream_ok(M, cin, N), "conv1x1_stream: internal: shape not eligible");
    DDK_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out) && (!bias || aligned16(bias)) && (!dmish_src || aligned16(dmish_src)) &&
                    (!resid || aligned16(resid)) && (!mish_out || aligned16(mish_out)),
                "conv1x1_stream: alignment");
    StreamParams p{x, w, bias, dmish_src, resid, out, mish_out, (int)(M / 16), pre_mish, post_mish};
    // 2 workgroups of 4 waves per CU; at least two tiles per wave where the map allows
    long long wgs = ceil_div((long long)p.tiles, 8);
    if (wgs > 512) wgs = 512;
    if (wgs < 1) wgs = 1;
    const dim3 grid((unsigned)wgs);
    if (cin == 64 && N == 64) hipLaunchKernelGGL((conv1x1_stream_kernel<64, 64>), grid, dim3(256), 0, st, p);
    else if (cin == 64 && N == 32) hipLaunchKernelGGL((conv1x1_stream_kernel<64, 32>), grid, dim3(256), 0, st, p);
    else if (cin == 32 && N == 64) hipLaunchKernelGGL((conv1x1_stream_kernel<32, 64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv1x1_stream_kernel<32, 32>), grid, dim3(256), 0, st, p);
    return check_launch("conv1x1_stream_kernel");
}

}  // namespace ddk
