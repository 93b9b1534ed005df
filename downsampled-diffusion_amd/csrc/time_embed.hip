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

// One workgroup per sample.  Both matrix-vector products split k over thread groups so that every thread
// streams a short run of independent 16-byte weight loads (a single thread walking a whole 128..512-long
// column is a chain of dependent-latency loads: 80 us instead of a few).
__global__ __launch_bounds__(256) void time_mlp_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                                       const float* __restrict__ w1t, const float* __restrict__ b1,
                                                       const float* __restrict__ w2t, const float* __restrict__ b2,
                                                       float* __restrict__ act, float* __restrict__ raw, int dim) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* e = sm;              // [dim]
    float* h1 = sm + dim;       // [4 dim]
    float* part = sm + 5 * dim; // [2048] partial sums
    const int b = blockIdx.x, half = dim >> 1, hid = dim * 4, tid = threadIdx.x;
    const float tf = (float)t[b];  // int64 * fp32 promotes to fp32 in torch (blocks.py:27)
    for (int j = tid; j < dim; j += 256) {
        const float a = tf * freqs[j < half ? j : j - half];
        e[j] = j < half ? sinf(a) : cosf(a);
    }
    __syncthreads();
    {   // h1 = Mish(W1 e + b1): thread = (k-group, 8 consecutive outputs)
        const int nj = hid >> 3, nkg = 256 / nj, kper = (dim + nkg - 1) / nkg;
        const int kg = tid / nj, jl = tid - kg * nj;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (kg < nkg) {
            const int k1 = min(dim, (kg + 1) * kper);
#pragma unroll 4
            for (int k = kg * kper; k < k1; ++k) {
                const float4 u = *reinterpret_cast<const float4*>(w1t + (long long)k * hid + jl * 8);
                const float4 v = *reinterpret_cast<const float4*>(w1t + (long long)k * hid + jl * 8 + 4);
                const float ek = e[k];
                s[0] += ek * u.x; s[1] += ek * u.y; s[2] += ek * u.z; s[3] += ek * u.w;
                s[4] += ek * v.x; s[5] += ek * v.y; s[6] += ek * v.z; s[7] += ek * v.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) part[kg * hid + jl * 8 + i] = s[i];
        }
        __syncthreads();
        for (int j = tid; j < hid; j += 256) {
            float a = b1[j];
            for (int g = 0; g < nkg; ++g) a += part[g * hid + j];
            h1[j] = mish_f(a);
        }
        __syncthreads();
    }
    {   // tv = W2 h1 + b2: thread = (k-group, 4 consecutive outputs)
        const int ni = dim >> 2, nkg = 256 / ni, kper = (hid + nkg - 1) / nkg;
        const int kg = tid / ni, il = tid - kg * ni;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (kg < nkg) {
            const int k1 = min(hid, (kg + 1) * kper);
#pragma unroll 8
            for (int k = kg * kper; k < k1; ++k) {
                const float4 u = *reinterpret_cast<const float4*>(w2t + (long long)k * dim + il * 4);
                const float hk = h1[k];
                s0 += hk * u.x; s1 += hk * u.y; s2 += hk * u.z; s3 += hk * u.w;
            }
            float* pp = part + kg * dim + il * 4;
            pp[0] = s0; pp[1] = s1; pp[2] = s2; pp[3] = s3;
        }
        __syncthreads();
        for (int i = tid; i < dim; i += 256) {
            float a = b2[i];
            for (int g = 0; g < nkg; ++g) a += part[g * dim + i];
            if (raw) raw[(long long)b * dim + i] = a;
            act[(long long)b * dim + i] = mish_f(a);
        }
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
    DDK_REQUIRE(B > 0 && dim >= 8 && dim % 8 == 0 && dim <= 512, "time_mlp: dim (= unet_chan) must be a multiple of 8, <= 512");
    DDK_REQUIRE(aligned16(w1t) && aligned16(w2t), "time_mlp: weight alignment");
    hipLaunchKernelGGL(time_mlp_kernel, dim3(B), dim3(256), ((size_t)dim * 5 + 2048) * sizeof(float), st, t, freqs, w1t, b1, w2t,
                       b2, act, raw, dim);
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
