// What does the fp32 matrix pipe sustain when its operands come from LDS, with and without a workgroup barrier per 16 MFMAs, at
// one and at two matrix waves per SIMD?  (Round 4: the Winograd kernel's stage runs 74-80 cycles per v_mfma_f32_32x32x2_f32 where the
// register-only loop of mfma_peak.hip runs 64.)  One workgroup per CU (100 KB of LDS), W waves; each wave repeats stages of 16 dependent
// MFMAs on one accumulator, fragments as in conv_wino.hip (one ds_read_b128 of A and of B per 4 MFMAs, read one quarter ahead).
//   bit 0: s_barrier behind every stage   bit 1: fragments from LDS (else constant registers)   bit 2: 4 accumulators round-robin per stage
//   bit 3: first B fragment of a stage read BEHIND the barrier (exposed), as the kernel must
//   RN (second template argument): n blocks per wave that share an A fragment (1 A read + RN B reads per 4 RN MFMAs)
//   W64 (third): every fragment as two ds_read_b64 instead of one ds_read_b128 (same bytes, twice the instructions)
// hipcc -O3 --offload-arch=gfx950 tools/mfma_lds.hip -o tools/bin/mfma_lds && tools/bin/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_cyc[4096];

template <int RN, bool W64>
__global__ __launch_bounds__(512) void kr(float* out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 24 * 1024; i += blockDim.x) smem[i] = (float)((i * 37 + blockIdx.x) & 255) * 1e-3f - 0.1f;
    __syncthreads();
    f32x16 acc[RN];
    for (int j = 0; j < RN; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    int foff[4];
    for (int q = 0; q < 4; ++q) foff[q] = (lane & 31) * 32 + (((2 * q + fh) ^ fsw) << 2);
    const float* As = smem + (wid & 3) * 1024;
    const float* Bs = smem + 8192 + (wid & 3) * 1024;
    auto rd = [&](const float* p) -> float4 {
        if (W64) {
            const float2 lo = *reinterpret_cast<const float2*>(p), hi = *reinterpret_cast<const float2*>(p + 2);
            return make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        return *reinterpret_cast<const float4*>(p);
    };
    float4 a[2], b[2][RN];
    a[0] = rd(As + foff[0]);
    for (int j = 0; j < RN; ++j) b[0][j] = rd(Bs + j * 1024 + foff[0]);
    a[1] = a[0];
    for (int j = 0; j < RN; ++j) b[1][j] = b[0][j];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        const float* A2 = As + (s & 1) * 4096;
        const float* B2 = Bs + (s & 1) * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            a[nxt] = rd(A2 + foff[(q + 1) & 3]);
#pragma unroll
            for (int j = 0; j < RN; ++j) b[nxt][j] = rd(B2 + j * 1024 + foff[(q + 1) & 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].x, b[cur][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].y, b[cur][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].z, b[cur][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].w, b[cur][j].w, acc[j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int j = 0; j < RN; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 512 + tid] = sum;
    if (tid == 0) g_cyc[blockIdx.x] = t1 - t0;
}

template <int RN, bool W64>
void runr(float* out, int waves) {
    const int blocks = 256, stages = 2000 / RN;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kr<RN, W64>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kr<RN, W64><<<blocks, waves * 64, 100 * 1024>>>(out, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kr<RN, W64><<<blocks, waves * 64, 100 * 1024>>>(out, stages);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_cyc), blocks * 8);
    std::sort(c.begin(), c.end());
    const double per_simd = (double)stages * 16 * RN * (waves / 4);
    printf("A shared by %d n blocks (%.4f reads per MFMA%s), barrier per stage, %d waves/SIMD: %6.1f cycles per MFMA on a SIMD, clock %.2f GHz\n", RN,
           (1.0 + RN) / (4.0 * RN), W64 ? ", as ds_read_b64 pairs" : "", waves / 4, c[blocks / 2] / per_simd, c[blocks / 2] / (ms * 1e6));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 24 * 1024; i += blockDim.x) smem[i] = (float)((i * 37 + blockIdx.x) & 255) * 1e-3f - 0.1f;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    int foff[4];
    for (int q = 0; q < 4; ++q) foff[q] = (lane & 31) * 32 + (((2 * q + fh) ^ fsw) << 2);
    const float* As = smem + (wid & 3) * 1024;
    const float* Bs = smem + 8192 + wid * 1024;
    float4 a[2], b[2];
    a[0] = *reinterpret_cast<const float4*>(As + foff[0]);
    b[0] = *reinterpret_cast<const float4*>(Bs + foff[0]);
    a[1] = a[0]; b[1] = b[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        const float* A2 = As + (s & 1) * 4096;
        const float* B2 = Bs + (s & 1) * 4096;
        if (MODE & 8) b[0] = *reinterpret_cast<const float4*>(B2 + foff[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            if (MODE & 2) {
                a[nxt] = *reinterpret_cast<const float4*>(A2 + foff[(q + 1) & 3]);
                if (!((MODE & 8) && q == 3)) b[nxt] = *reinterpret_cast<const float4*>(B2 + foff[(q + 1) & 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16& c = acc[(MODE & 4) ? q : 0];
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].x, b[cur].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].y, b[cur].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].z, b[cur].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].w, b[cur].w, c, 0, 0, 0);
        }
        if (MODE & 1) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 512 + tid] = sum;
    if (tid == 0) g_cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(float* out, int waves) {
    const int blocks = 256, stages = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, waves * 64, 100 * 1024>>>(out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, waves * 64, 100 * 1024>>>(out, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_cyc), blocks * 8);
    std::sort(c.begin(), c.end());
    const double per_simd = (double)stages * 16 * (waves / 4);
    const double flop = (double)blocks * waves * stages * 16 * 4096.0;
    printf("mode %2d (%s%s%s%s) %d waves/SIMD: %6.1f cycles per MFMA on a SIMD (median WG), %.3f ms, %.1f TFLOP/s, clock %.2f GHz\n", MODE,
           (MODE & 2) ? "LDS fragments" : "register operands", (MODE & 1) ? ", barrier per stage" : "", (MODE & 4) ? ", 4 accumulators" : "",
           (MODE & 8) ? ", first B read behind the barrier" : "", waves / 4, c[blocks / 2] / per_simd, ms, flop / ms / 1e9,
           c[blocks / 2] / (ms * 1e6));
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int waves : {4, 8}) {
        run<0>(out, waves); run<2>(out, waves); run<3>(out, waves); run<11>(out, waves); run<6>(out, waves); run<7>(out, waves); run<1>(out, waves);
        runr<1, false>(out, waves); runr<1, true>(out, waves); runr<2, false>(out, waves); runr<4, false>(out, waves); runr<4, true>(out, waves);
    }
    return 0;
}
