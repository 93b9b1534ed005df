// time_embed.hip -- timestep embedding path of the UNet.
//
// Reference: models/unet/blocks.py:22-29 (SinusoidalPosEmb), models/unet/unet.py:30-35 (Linear -> Mish ->
// Linear), models/unet/blocks.py:92-95,106-109 (per-ResnetBlock Mish -> Linear(time_dim, C_out)).
// The 17 per-block Linears all consume the same Mish(time_mlp(t)) vector, so they are evaluated as ONE
// [dim] x [dim][sum C_out] product (ddk_time_proj) whose row b / column slice is the shift the
// GroupNorm+Mish kernel adds for that block.  Weights are held transposed ([in][out]) so consecutive
// threads read consecutive addresses.
#include "ddk_internal.h"

namespace ddk {

__global__ __launch_bounds__(256) void time_mlp_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                                       const float* __restrict__ w1t, const float* __restrict__ b1,
                                                       const float* __restrict__ w2t, const float* __restrict__ b2,
                                                       float* __restrict__ act, float* __restrict__ raw, int dim) {
    extern __shared__ float sm[];
    float* e = sm;          // [dim]
    float* h1 = sm + dim;   // [4 dim]
    const int b = blockIdx.x, half = dim >> 1, hid = dim * 4;
    const float tf = (float)t[b];  // int64 * fp32 promotes to fp32 in torch (blocks.py:27)
    for (int j = threadIdx.x; j < dim; j += blockDim.x) {
        const float a = tf * freqs[j < half ? j : j - half];
        e[j] = j < half ? sinf(a) : cosf(a);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < hid; j += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < dim; ++k) s += e[k] * w1t[(long long)k * hid + j];
        h1[j] = mish_f(s + b1[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < hid; ++j) s += h1[j] * w2t[(long long)j * dim + i];
        s += b2[i];
        if (raw) raw[(long long)b * dim + i] = s;
        act[(long long)b * dim + i] = mish_f(s);
    }
}

// out[b][j] = sum_k act[b][k] wt[k][j] + bias[j]; 8 samples share one pass over the weights.
__global__ __launch_bounds__(256) void time_proj_kernel(const float* __restrict__ act, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ out, int B, int dim,
                                                        int n_out) {
    extern __shared__ float a_s[];  // [8][dim]
    const int b0 = blockIdx.y * 8;
    const int nb = min(8, B - b0);
    for (int i = threadIdx.x; i < 8 * dim; i += blockDim.x) {
        const int bb = i / dim;
        a_s[i] = bb < nb ? act[(long long)(b0 + bb) * dim + (i - bb * dim)] : 0.f;
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    float acc[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) acc[bb] = 0.f;
    for (int k = 0; k < dim; ++k) {
        const float w = wt[(long long)k * n_out + j];
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) acc[bb] += a_s[bb * dim + k] * w;
    }
    const float bj = bias[j];
    for (int bb = 0; bb < nb; ++bb) out[(long long)(b0 + bb) * n_out + j] = acc[bb] + bj;
}

int time_mlp(const int64_t* t, const float* freqs, const float* w1t, const float* b1, const float* w2t, const float* b2,
             float* act, float* raw, int B, int dim, hipStream_t st) {
    DDK_REQUIRE(t && freqs && w1t && b1 && w2t && b2 && act, "time_mlp: null pointer");
    DDK_REQUIRE(B > 0 && dim > 0 && dim % 2 == 0 && dim <= 2048, "time_mlp: B / dim");
    hipLaunchKernelGGL(time_mlp_kernel, dim3(B), dim3(256), (size_t)dim * 5 * sizeof(float), st, t, freqs, w1t, b1, w2t, b2, act,
                       raw, dim);
    return check_launch("time_mlp_kernel");
}

int time_proj(const float* act, const float* wt, const float* bias, float* out, int B, int dim, int n_out, hipStream_t st) {
    DDK_REQUIRE(act && wt && bias && out, "time_proj: null pointer");
    DDK_REQUIRE(B > 0 && dim > 0 && n_out > 0 && dim <= 2048, "time_proj: B / dim / n_out");
    dim3 grid((unsigned)ceil_div(n_out, 256), (unsigned)ceil_div(B, 8));
    hipLaunchKernelGGL(time_proj_kernel, grid, dim3(256), (size_t)dim * 8 * sizeof(float), st, act, wt, bias, out, B, dim, n_out);
    return check_launch("time_proj_kernel");
}

}  // namespace ddk

extern "C" {
int ddk_time_mlp(const int64_t* t, const float* freqs, const float* w1t, const float* b1, const float* w2t, const float* b2,
                 float* act, float* raw, int B, int dim, ddk_stream_t s) {
    return ddk::time_mlp(t, freqs, w1t, b1, w2t, b2, act, raw, B, dim, ddk::as_stream(s));
}
int ddk_time_proj(const float* act, const float* wt, const float* bias, float* out, int B, int dim, int n_out, ddk_stream_t s) {
    return ddk::time_proj(act, wt, bias, out, B, dim, n_out, ddk::as_stream(s));
}
}
