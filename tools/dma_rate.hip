// LDS-DMA intake of one CU against the bytes it keeps in flight (gfx950).  Tuning tool, not product code.
//
// Question it answers: is the ~7-8 B/clk/CU that the im2col conv kernels take in a hardware ceiling of the LDS-DMA path,
// or Little's law at their in-flight depth (one 16-24 KB chunk per workgroup)?  Every wave issues 1-KiB pieces
// (global_load_lds_dwordx4, 64 lanes x 16 B) into a private LDS ring and waits with a counted vmcnt(DEPTH-1): DEPTH
// pieces stay in flight per wave.  No compute.  Patterns:
//   0  contiguous 1-KiB pieces, every wave walks its own stream (weights-like)
//   1  8 rows x 128 B per piece, rows strided by `row_stride` floats inside a window (im2col activations-like)
// Source footprint `mb` MiB: 2 = L2-resident per XCD, 64 = Infinity-Cache resident, 1024 = HBM.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma_rate.hip -o tools/bin/dma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void lds_dma16(const float* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(g) : "memory", "m0");
}

template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const float* __restrict__ src, unsigned long long mask_floats, int iters, int pattern,
                                                  int row_stride, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + (unsigned)(wid * DEPTH * 1024);
    const unsigned long long wave = (unsigned long long)blockIdx.x * nw + wid;
    unsigned long long pos = wave * (unsigned long long)iters * 256ull;   // floats
    const int prow = lane >> 3, ppos = lane & 7;
    int slot = 0;
    for (int i = 0; i < iters; ++i) {
        const float* g;
        if (pattern == 0) {
            g = src + ((pos + (unsigned long long)i * 256ull + lane * 4) & mask_floats);
        } else {
            // piece i of this wave: 8 rows, row r at (base_row + r) * row_stride, 32 floats each (ppos picks the 16-B chunk)
            const unsigned long long row = (wave * 977ull + (unsigned long long)i * 8ull + prow);
            g = src + ((row * (unsigned long long)row_stride + ppos * 4) & mask_floats);
        }
        lds_dma16(g, lds_base + (unsigned)(slot * 1024));
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0 && iters < 0) sink[blockIdx.x] = smem[0];
}

template <int DEPTH>
static double run(const float* src, size_t floats, int blocks, int waves, int iters, int pattern, int row_stride) {
    const size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dma_kernel<DEPTH><<<blocks, waves * 64, lds>>>(src, floats - 1, iters, pattern, row_stride, nullptr);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        dma_kernel<DEPTH><<<blocks, waves * 64, lds>>>(src, floats - 1, iters, pattern, row_stride, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("error: %s\n", hipGetErrorString(e)); return 0; }
    const double bytes = (double)blocks * waves * iters * 1024.0;
    return bytes / (best * 1e-3);   // B/s
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    for (int mb : {2, 64, 1024}) {
        const size_t floats = (size_t)mb << 18;
        float* src;
        if (hipMalloc(&src, floats * 4) != hipSuccess) { printf("alloc %d MiB failed\n", mb); continue; }
        hipMemset(src, 0, floats * 4);
        for (int pattern : {0, 1}) {
            for (int blocks : {256, 512}) {
                for (int waves : {4, 8}) {
                    if (blocks == 512 && waves == 8) continue;
                    printf("src %4d MiB pattern %d  %d WG x %d waves:", mb, pattern, blocks, waves);
                    double r[7];
                    r[0] = run<1>(src, floats, blocks, waves, iters, pattern, 256);
                    r[1] = run<2>(src, floats, blocks, waves, iters, pattern, 256);
                    r[2] = run<4>(src, floats, blocks, waves, iters, pattern, 256);
                    r[3] = run<8>(src, floats, blocks, waves, iters, pattern, 256);
                    r[4] = run<16>(src, floats, blocks, waves, iters, pattern, 256);
                    r[5] = (waves * 32 * 1024 * (blocks / 256) <= 160 * 1024) ? run<32>(src, floats, blocks, waves, iters, pattern, 256) : 0;
                    const int d[6] = {1, 2, 4, 8, 16, 32};
                    for (int k = 0; k < 6; ++k)
                        if (r[k] > 0) printf("  d%-2d %5.2f TB/s (%4.1f GB/s/CU, %2d KiB/CU in flight)", d[k], r[k] / 1e12, r[k] / 256e9,
                                             d[k] * waves * (blocks / 256));
                    printf("\n");
                    fflush(stdout);
                }
            }
        }
        hipFree(src);
    }
    return 0;
}
