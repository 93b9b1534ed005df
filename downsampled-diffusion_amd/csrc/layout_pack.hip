// layout_pack.hip -- NCHW <-> NHWC boundary conversions, weight repacking, and the small-N 1x1 projection.
//
// The reference keeps NCHW tensors and OIHW weights (models/unet/unet.py:74).  The HIP
// path computes in NHWC with [out][tap][in] weights; the canonical tensors stay what state_dict() holds
// and these kernels derive the packed copies.
#include "ddk_internal.h"
#include "pack_elems.h"

namespace ddk {

static int grid1d(long long n) {
    const long long b = ceil_div(n > 0 ? n : 1, 256);
    return (int)(b < 4096 ? b : 4096);
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                           int c_pad, long long total) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c_pad);
        const long long pix = i / c_pad;
        const long long hw = (long long)H * W;
        const long long b = pix / hw, r = pix - b * hw;
        dst[i] = c < C ? src[(b * C + c) * hw + r] : 0.f;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                           int c_stride, long long total) {
    const long long hw = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i % hw;
        const long long bc = i / hw;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        dst[i] = src[(b * hw + r) * c_stride + c];
    }
}

__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int c_pad,
                                                           long long total) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c_pad);
        const long long m = i / c_pad;
        dst[i] = c < C ? src[m * C + c] : 0.f;
    }
}

// dst[o][tap][i_pad] <- w[o][i][ky][kx] (two separately padded sources at generic widths): pack_elems.h
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I,
                                                               int taps, int i_pad, long long total, int split, int split_pad) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        pack_conv_weight_elem(w, dst, idx, I, taps, i_pad, split, split_pad);
}

// dst[phase][o][tap][i] <- w[i][o][ky][kx]: pack_elems.h
__global__ __launch_bounds__(256) void pack_convT_weight_kernel(const float* __restrict__ w, float* __restrict__ dst, int I, int O,
                                                                long long total, int Ip, int Op) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        pack_convT_weight_elem(w, dst, idx, I, O, Ip, Op);
}

// dst[i][col0 + o] <- w[o][i]   (dst is [I][ld])
__global__ __launch_bounds__(256) void pack_linear_T_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I, int ld,
                                                            int col0, long long total) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int o = (int)(idx % O);
        const long long i = idx / O;
        dst[i * ld + col0 + o] = w[(long long)o * I + i];
    }
}

// Small-N 1x1 conv: LPP = C/4 lanes (<= 64) share a pixel; each lane holds VPL float4 of the pixel and dots
// it with every output row of W, then the lane group xor-reduces.  HBM-bound on reading x once.
template <int LPP, int VPL>
__global__ __launch_bounds__(256) void conv1x1_small_n_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out, long long M,
                                                              int C, int n_out) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63;
    const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
    const long long pix = wave * PPW + lane / LPP;
    const int sub = lane % LPP;
    const bool ok = pix < M;
    float4 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        v[i] = ok ? *reinterpret_cast<const float4*>(x + pix * C + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int co = 0; co < n_out; ++co) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float4 ww = *reinterpret_cast<const float4*>(w + (long long)co * C + (sub + i * LPP) * 4);
            s += (v[i].x * ww.x + v[i].y * ww.y) + (v[i].z * ww.z + v[i].w * ww.w);
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (ok && sub == 0) out[pix * n_out + co] = s + (bias ? bias[co] : 0.f);
    }
}

// n_out <= 8 (the UNet's final projection to the latent / image channels): the whole weight matrix lives in registers
// (8 x VPL float4 per lane), waves walk the pixels grid-stride -- no weight reload per pixel, x is read exactly once.
template <int LPP, int VPL>
__global__ __launch_bounds__(256) void conv1x1_n8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, long long M, int C,
                                                         int n_out) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, sub = lane % LPP, pl = lane / LPP;
    float4 ww[8][VPL];
    const bool h2 = sub & (LPP / 2), h4 = sub & (LPP / 4), h8 = sub & (LPP / 8);
    const int my_co = (h2 ? 4 : 0) + (h4 ? 2 : 0) + (h8 ? 1 : 0);      // the output channel this lane ends up holding (see below)
    const float my_bias = (bias && my_co < n_out) ? bias[my_co] : 0.f;
#pragma unroll
    for (int co = 0; co < 8; ++co) {
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            ww[co][i] = co < n_out ? *reinterpret_cast<const float4*>(w + (long long)co * C + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6); wave * PPW < M; wave += nwaves) {
        const long long pix = wave * PPW + pl;
        const bool ok = pix < M;
        float4 v[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            v[i] = ok ? *reinterpret_cast<const float4*>(x + pix * C + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float s[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) t += (v[i].x * ww[co][i].x + v[i].y * ww[co][i].y) + (v[i].z * ww[co][i].z + v[i].w * ww[co][i].w);
            s[co] = t;
        }
        // Reduce the 8 partial dot products over the pixel's LPP lanes with 4 + 2 + 1 + (log2(LPP) - 3) shuffles instead of
        // 8 log2(LPP): each of the first three butterfly steps also halves the number of outputs a lane carries (the upper half
        // of the lane group keeps the upper half of the outputs), so afterwards lane l holds output
        // co = 4 bit(LPP/2) + 2 bit(LPP/4) + bit(LPP/8) of its lane index, summed over 8 lanes; the remaining steps are plain sums.
        float t4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = h2 ? s[4 + i] : s[i], send = h2 ? s[i] : s[4 + i];
            t4[i] = keep + __shfl_xor(send, LPP / 2, 64);
        }
        float t2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = h4 ? t4[2 + i] : t4[i], send = h4 ? t4[i] : t4[2 + i];
            t2[i] = keep + __shfl_xor(send, LPP / 4, 64);
        }
        float r = (h8 ? t2[1] : t2[0]) + __shfl_xor(h8 ? t2[0] : t2[1], LPP / 8, 64);
#pragma unroll
        for (int o = LPP / 16; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if (ok && (sub & (LPP / 8 - 1)) == 0 && my_co < n_out) out[pix * n_out + my_co] = r + my_bias;
    }
}

int conv1x1_small_n(const float* x, const float* w, const float* bias, float* out, long long M, int C, int n_out, hipStream_t st) {
    DDK_REQUIRE(x && w && out && M > 0 && n_out > 0 && n_out <= 64, "conv1x1_small_n: arguments (n_out <= 64)");
    DDK_REQUIRE(aligned16(x) && aligned16(w), "conv1x1_small_n: alignment");
    if (n_out <= 8 && (C == 64 || C == 128 || C == 256)) {
        const long long want = ceil_div(M, 8);
        const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
        if (C == 64) hipLaunchKernelGGL((conv1x1_n8_kernel<16, 1>), dim3(blocks), dim3(256), 0, st, x, w, bias, out, M, C, n_out);
        else if (C == 128) hipLaunchKernelGGL((conv1x1_n8_kernel<32, 1>), dim3(blocks), dim3(256), 0, st, x, w, bias, out, M, C, n_out);
        else hipLaunchKernelGGL((conv1x1_n8_kernel<32, 2>), dim3(blocks), dim3(256), 0, st, x, w, bias, out, M, C, n_out);
        return check_launch("conv1x1_n8_kernel");
    }
#define C1_CASE(LPP, VPL)                                                                                                       \
    do {                                                                                                                        \
        const long long waves = ceil_div(M, 64 / LPP);                                                                          \
        hipLaunchKernelGGL((conv1x1_small_n_kernel<LPP, VPL>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, st, x, w, bias, out, \
                           M, C, n_out);                                                                                        \
        return check_launch("conv1x1_small_n_kernel");                                                                          \
    } while (0)
    switch (C) {
        case 32: C1_CASE(8, 1);
        case 64: C1_CASE(16, 1);
        case 96: C1_CASE(8, 3);
        case 128: C1_CASE(32, 1);
        case 192: C1_CASE(16, 3);
        case 256: C1_CASE(64, 1);
        case 384: C1_CASE(32, 3);
        case 512: C1_CASE(64, 2);
        // the other multiples of 32 up to 512 (padded pitches of widths that are not multiples of 32; no tuning)
        case 160: C1_CASE(8, 5);
        case 224: C1_CASE(8, 7);
        case 288: C1_CASE(8, 9);
        case 320: C1_CASE(16, 5);
        case 352: C1_CASE(8, 11);
        case 416: C1_CASE(8, 13);
        case 448: C1_CASE(16, 7);
        case 480: C1_CASE(8, 15);
        default: break;
    }
#undef C1_CASE
    return fail_arg("conv1x1_small_n: unsupported channel count");
}

}  // namespace ddk

using namespace ddk;

extern "C" {

int ddk_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int c_pad, ddk_stream_t s) {
    DDK_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && c_pad >= C, "nchw_to_nhwc: arguments");
    const long long total = (long long)B * H * W * c_pad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), src, dst, C, H, W, c_pad, total);
    return check_launch("nchw_to_nhwc_kernel");
}

int ddk_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int c_stride, ddk_stream_t s) {
    DDK_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && c_stride >= C, "nhwc_to_nchw: arguments");
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), src, dst, C, H, W, c_stride, total);
    return check_launch("nhwc_to_nchw_kernel");
}

int ddk_pad_channels(const float* src, float* dst, long long M, int C, int c_pad, ddk_stream_t s) {
    DDK_REQUIRE(src && dst && M > 0 && C > 0 && c_pad >= C, "pad_channels: arguments");
    const long long total = M * c_pad;
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), src, dst, C, c_pad, total);
    return check_launch("pad_channels_kernel");
}

int ddk_pack_conv_weight(const float* w, float* dst, int O, int I, int KH, int KW, int i_pad, ddk_stream_t s) {
    DDK_REQUIRE(w && dst && O > 0 && I > 0 && KH > 0 && KW > 0 && i_pad >= I, "pack_conv_weight: arguments");
    const long long total = (long long)O * KH * KW * i_pad;
    DDK_REQUIRE(total < (1LL << 31), "pack_conv_weight: 2^31 elements or more");
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), w, dst, O, I, KH * KW, i_pad, total, I, i_pad);
    return check_launch("pack_conv_weight_kernel");
}

int ddk_pack_conv_weight_split(const float* w, float* dst, int O, int I, int KH, int KW, int i_pad, int split, int split_pad, ddk_stream_t s) {
    DDK_REQUIRE(w && dst && O > 0 && I > 0 && KH > 0 && KW > 0 && split > 0 && split <= I && split_pad >= split &&
                    i_pad >= split_pad + (I - split), "pack_conv_weight_split: arguments");
    const long long total = (long long)O * KH * KW * i_pad;
    DDK_REQUIRE(total < (1LL << 31), "pack_conv_weight_split: 2^31 elements or more");
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), w, dst, O, I, KH * KW, i_pad, total, split,
                       split_pad);
    return check_launch("pack_conv_weight_kernel");
}

int ddk_pack_convT_weight(const float* w, float* dst, int I, int O, ddk_stream_t s) {
    DDK_REQUIRE(w && dst && I > 0 && O > 0, "pack_convT_weight: arguments");
    const long long total = 16LL * I * O;
    DDK_REQUIRE(total < (1LL << 31), "pack_convT_weight: 2^31 elements or more");
    hipLaunchKernelGGL(pack_convT_weight_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), w, dst, I, O, total, I, O);
    return check_launch("pack_convT_weight_kernel");
}

int ddk_pack_convT_weight_padded(const float* w, float* dst, int I, int O, int c_pad, ddk_stream_t s) {
    DDK_REQUIRE(w && dst && I > 0 && O > 0 && c_pad >= I && c_pad >= O && c_pad % 32 == 0, "pack_convT_weight_padded: arguments");
    const long long total = 16LL * c_pad * c_pad;
    hipLaunchKernelGGL(pack_convT_weight_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), w, dst, I, O, total, c_pad, c_pad);
    return check_launch("pack_convT_weight_kernel");
}

int ddk_pack_linear_T(const float* w, float* dst, int O, int I, int ld, int col0, ddk_stream_t s) {
    DDK_REQUIRE(w && dst && O > 0 && I > 0 && ld >= col0 + O && col0 >= 0, "pack_linear_T: arguments");
    const long long total = (long long)O * I;
    hipLaunchKernelGGL(pack_linear_T_kernel, dim3(grid1d(total)), dim3(256), 0, as_stream(s), w, dst, O, I, ld, col0, total);
    return check_launch("pack_linear_T_kernel");
}

int ddk_conv1x1_small_n(const float* x, const float* w, const float* bias, float* out, long long M, int C, int n_out, ddk_stream_t s) {
    return conv1x1_small_n(x, w, bias, out, M, C, n_out, as_stream(s));
}
}
