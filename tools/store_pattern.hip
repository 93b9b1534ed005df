// Microbenchmark: how fast do 16.8 MB of NHWC output leave the chip under the two store patterns of this library's epilogues?
//   (a) "accumulator order": lane (pixel p = lane & 31, half h = lane >> 5) stores a float4 at out[(p0 + p) * N + c0 + 4 h + 8 q],
//       q = 0..3 -- 32-byte pieces of 32 different 512-byte rows per instruction (conv_first_kernel, conv1x1_ws_kernel)
//   (b) "row order": 8 lanes cover 128 contiguous bytes of one pixel row, 8 rows per instruction (the LDS-transposed epilogue of
//       the im2col kernels)
// hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 128;
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float* __restrict__ out, float v) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long tile = (long long)blockIdx.x * 128;            // 128 pixels x 128 channels per workgroup, as conv_first
    const int pg = wave & 3, par = wave >> 2;
    const float4 val = make_float4(v, v + 1, v + 2, v + 3);
    if (MODE == 0) {
        const int pl = lane & 31, h = lane >> 5;
        for (int nb = par; nb < 4; nb += 2)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(out + (tile + pg * 32 + pl) * N + nb * 32 + 4 * h + 8 * q) = val;
    } else {
        const int row = lane >> 3, c4 = (lane & 7) * 4;                // 8 rows x 128 bytes per instruction
        for (int nb = par; nb < 4; nb += 2)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
                *reinterpret_cast<float4*>(out + (tile + pg * 32 + pass * 8 + row) * N + nb * 32 + c4) = val;
    }
}

template <int MODE>
static float run(float* out, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, out, (float)i);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}

int main() {
    const int pixels = 32 * 32 * 32, blocks = pixels / 128;
    float* out;
    hipMalloc(&out, (size_t)pixels * N * sizeof(float));
    const double mb = (double)pixels * N * 4 / 1e6;
    for (int rep = 0; rep < 2; ++rep) {
        const float ta = run<0>(out, blocks, 200), tb = run<1>(out, blocks, 200);
        printf("%.1f MB of NHWC output, %d workgroups: accumulator-order stores %.2f us (%.0f GB/s) | row-order stores %.2f us (%.0f GB/s)\n", mb,
               blocks, ta, mb / ta * 1e3, tb, mb / tb * 1e3);
    }
    return 0;
}
