// Weight-stream intake of a CU when all 32 workgroups of an XCD stream the SAME bytes at the same time (gfx950) -- the access
// pattern of the image-local conv kernels (conv_local.hip: workgroup = one image x one n tile; blockIdx % 8 = n tile = XCD), which
// tools/dma_rate.hip does not cover (there every wave walks its own stream).  256 workgroups x 8 waves; workgroup b streams region
// (b % 8) of `kb` KiB, wave w takes pieces w, w + 8, ...; DEPTH pieces of 1 KiB in flight per wave.
//   mode 0: global_load_dwordx4 into registers (what conv_local.hip does)      mode 1: LDS-DMA (global_load_lds_dwordx4)
// Build: hipcc -O3 --offload-arch=gfx950 tools/share_rate.hip -o tools/bin/share_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void lds_dma16(const float* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(g) : "memory", "m0");
}

template <int DEPTH, int MODE>
__global__ __launch_bounds__(512) void share_kernel(const float* __restrict__ src, int pieces, int region_floats, int rotate, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int region = rotate ? (blockIdx.x + blockIdx.x / 8) % 8 : blockIdx.x % 8;
    const float* base = src + (size_t)region * region_floats + lane * 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + (unsigned)(wid * DEPTH * 1024);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1) {
        int slot = 0;
        for (int i = wid; i < pieces; i += 8) {
            lds_dma16(base + (size_t)i * 256, lds_base + (unsigned)(slot * 1024));
            slot = slot + 1 == DEPTH ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        float4 r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) r[d] = make_float4(0.f, 0.f, 0.f, 0.f);
        int i = wid;
        for (; i + 8 * (DEPTH - 1) < pieces; i += 8 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = *reinterpret_cast<const float4*>(base + (size_t)(i + 8 * d) * 256);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x += r[d].x; acc.y += r[d].y; acc.z += r[d].z; acc.w += r[d].w; }
        }
    }
    __syncthreads();
    if (sink && acc.x + acc.y + acc.z + acc.w == 12345.f) sink[blockIdx.x] = acc.x + smem[0];
}

template <int DEPTH, int MODE>
static double run(const float* src, int kb, int rotate, float* sink) {
    const int pieces = kb, region_floats = kb * 256;
    const size_t lds = MODE == 1 ? (size_t)8 * DEPTH * 1024 : 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&share_kernel<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    share_kernel<DEPTH, MODE><<<256, 512, lds>>>(src, pieces, region_floats, rotate, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) share_kernel<DEPTH, MODE><<<256, 512, lds>>>(src, pieces, region_floats, rotate, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    return best * 1e3;   // us per launch
}

int main() {
    float *src, *sink;
    hipMalloc(&src, (size_t)8 * 1024 * 1024 * 4);
    hipMemset(src, 0, (size_t)8 * 1024 * 1024 * 4);
    hipMalloc(&sink, 4096);
    for (int kb : {288, 512}) {
        for (int rotate : {0, 1}) {
            printf("%d KiB per workgroup, %s:\n", kb, rotate ? "n tiles rotated over the XCDs" : "one region per XCD (32 workgroups share it)");
            const double a2 = run<2, 0>(src, kb, rotate, sink), a4 = run<4, 0>(src, kb, rotate, sink), a8 = run<8, 0>(src, kb, rotate, sink);
            printf("  registers: depth 2 %6.2f us (%5.1f GB/s/CU)  depth 4 %6.2f us (%5.1f)  depth 8 %6.2f us (%5.1f)\n", a2, kb * 1.024e-3 / a2 * 1e3,
                   a4, kb * 1.024e-3 / a4 * 1e3, a8, kb * 1.024e-3 / a8 * 1e3);
            const double d2 = run<2, 1>(src, kb, rotate, sink), d4 = run<4, 1>(src, kb, rotate, sink), d8 = run<8, 1>(src, kb, rotate, sink),
                         d16 = run<16, 1>(src, kb, rotate, sink);
            printf("  LDS-DMA:   depth 2 %6.2f us (%5.1f GB/s/CU)  depth 4 %6.2f us (%5.1f)  depth 8 %6.2f us (%5.1f)  depth 16 %6.2f us (%5.1f)\n", d2,
                   kb * 1.024e-3 / d2 * 1e3, d4, kb * 1.024e-3 / d4 * 1e3, d8, kb * 1.024e-3 / d8 * 1e3, d16, kb * 1.024e-3 / d16 * 1e3);
            fflush(stdout);
        }
    }
    return 0;
}
