// Which ingredient of the conv inner loop slows v_mfma_f32_32x32x2_f32 below its 64-cycle pace?
//   mode 0: 64 MFMAs / iter, operands = 32 distinct VGPRs held in registers (2x2 tiles, 4 q-steps, 4 e)
//   mode 1: same, operands re-read from LDS (ds_read_b128) every q-step, prefetched one q-step ahead
//   mode 2: mode 1 + one __syncthreads() per iteration
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const float* base = lds + (lane & 31) * 32 + (lane >> 5) * 4;
    float4 a[4][2], b[4][2];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 2; ++i) {
        a[q][i] = *reinterpret_cast<const float4*>(base + i * 1024 + q * 8);
        b[q][i] = *reinterpret_cast<const float4*>(base + 2048 + i * 1024 + q * 8);
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) __syncthreads();
        float4 ca[2], cb[2];
        if (MODE >= 1) {
            for (int i = 0; i < 2; ++i) { ca[i] = *reinterpret_cast<const float4*>(base + i * 1024 + ((it & 3) * 8)); cb[i] = *reinterpret_cast<const float4*>(base + 2048 + i * 1024 + ((it & 3) * 8)); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 na[2], nb[2];
            if (MODE >= 1 && q < 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { na[i] = *reinterpret_cast<const float4*>(base + i * 1024 + (((it + q + 1) & 3) * 8)); nb[i] = *reinterpret_cast<const float4*>(base + 2048 + i * 1024 + (((it + q + 1) & 3) * 8)); }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 av = MODE >= 1 ? ca[i] : a[q][i], bv = MODE >= 1 ? cb[j] : b[q][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                }
            if (MODE >= 1 && q < 3) { for (int i = 0; i < 2; ++i) { ca[i] = na[i]; cb[i] = nb[i]; } }
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* out, const float* in, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 5000;
    k<MODE><<<blocks, 256>>>(out, 10, in);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, in);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("mode %d blocks %4d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, best, flop / best / 1e9);
}
int main() {
    float *out, *in; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 8192 * 4);
    static float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) { run<0>(out, in, blocks); run<1>(out, in, blocks); run<2>(out, in, blocks); }
    return 0;
}
