// Sustained fp32 MFMA rate of this chip: register-only v_mfma_f32_32x32x2_f32 loops on every CU.
//   mode 0: 4 accumulators round-robin (consecutive MFMAs independent)
//   mode 1: 4 consecutive MFMAs on the SAME accumulator, then the next accumulator (the conv kernel's old order)
//   mode 2: a single accumulator (fully dependent chain)
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(out, 100, 0.5f, 0.25f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("mode %d blocks %4d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", MODE, blocks, blocks / 256, best, flop / best / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int blocks : {256, 512}) { run<0>(out, blocks); run<1>(out, blocks); run<2>(out, blocks); }
    return 0;
}
