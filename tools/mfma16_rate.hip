// Issue rate of v_mfma_f32_16x16x4_f32 per SIMD: NACC accumulators in rotation, WAVES waves per workgroup (one workgroup per CU).
// hipcc -O3 --offload-arch=gfx950 tools/mfma16_rate.hip -o tools/bin/mfma16_rate && tools/bin/mfma16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void k32(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, int waves, int per_iter) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    launch(out, 10, cyc, waves);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(out, iters, cyc, waves);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * per_iter * waves / 4.0;
    printf("%-34s %2d waves/CU: %7.1f shader-clock ticks per MFMA per SIMD (wave 0), %6.1f ns per MFMA per SIMD by events\n", name, waves,
           (double)h / n_per_simd, ms * 1e6 / n_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int waves : {4, 8, 12}) {
        run("16x16x4 f32, 1 accumulator", [](float* o, int it, unsigned long long* c, int w) { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(64 * w), 0, 0, o, it, c); }, waves, 8);
        run("16x16x4 f32, 2 accumulators", [](float* o, int it, unsigned long long* c, int w) { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(64 * w), 0, 0, o, it, c); }, waves, 16);
        run("16x16x4 f32, 4 accumulators", [](float* o, int it, unsigned long long* c, int w) { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(64 * w), 0, 0, o, it, c); }, waves, 32);
        run("16x16x4 f32, 8 accumulators", [](float* o, int it, unsigned long long* c, int w) { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(64 * w), 0, 0, o, it, c); }, waves, 64);
        run("32x32x2 f32, 2 accumulators", [](float* o, int it, unsigned long long* c, int w) { hipLaunchKernelGGL(k32, dim3(256), dim3(64 * w), 0, 0, o, it, c); }, waves, 16);
    }
    return 0;
}
