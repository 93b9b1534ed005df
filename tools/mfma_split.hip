// Would the Winograd conv's position GEMMs run faster as fp32-accurate bf16 products?  (Round 5, DESIGN.md section 8: an fp32 product
// a b equals, to fp32 accuracy, the six bf16 products hh, hm, mh, mm, hl, lh of three-way splits a = a_h + a_m + a_l; the bf16 matrix
// pipe is 16x the fp32 one.)  This measures the matrix-wave loop of the proposed kernel -- NOT the kernel -- against the present one's,
// on one workgroup per CU, 8 matrix waves (two per SIMD) + 4 waves that write LDS at the rate the loader's V images and the U DMA
// would (the fills share the LDS pipe with the fragment reads):
//   fp32  : a stage = one Winograd position: per wave 32 tiles x 32 channels x 32 n: 8 ds_read_b128 + 16 v_mfma_f32_32x32x2_f32  (today)
//   split : the same GEMM per (n block of 32, half of the 32 tiles): 3 A + 6 B ds_read_b128 (three bf16 pieces, 8 k per lane) and
//           2 n sub-blocks x 6 products = 12 v_mfma_f32_16x16x32_bf16; 1.5x the bytes in LDS (fills 1.5x as well)
// Prints cycles per stage (= per position and workgroup) and the ratio.  Also checks the split's numerics on the device: a 16 x 16 x 32
// product from random fp32 data, six bf16 products vs the fp64 host result.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_split.hip -o tools/bin/mfma_split && tools/bin/mfma_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_cyc[1024];

// ---- the loop of the proposed kernel.  LDS: ring of 4 stages x 30 KB (V 3 pieces x 32 rows x 64 B + U 3 pieces x 128 rows x 64 B)
template <int FILL>
__global__ __launch_bounds__(768) void k_split(float* out, int stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 30 * 1024 * 4 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem_b)[i] = 0x3c003c00u + (unsigned)(i * 2654435761u >> 28);
    __syncthreads();
    constexpr int STAGE = 30 * 1024;
    unsigned long long t0 = 0, t1 = 0;
    if (wid >= 8) {
        // "loader": FILL 16-byte LDS writes per thread and stage (256 threads x 16 B x 8 = 32 KB ~ one stage of split operands; 5 ~ fp32 stage)
        const f32x4 v = {1.f, 2.f, 3.f, 4.f};
        for (int s = 0; s < stages; ++s) {
            unsigned char* dst = smem_b + ((s + 2) & 3) * STAGE + (tid - 512) * 16;
#pragma unroll
            for (int j = 0; j < FILL; ++j) *reinterpret_cast<f32x4*>(dst + (j * 4096) % (STAGE - 4096)) = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    const int nb = wid & 3, mh = wid >> 2;                 // n block of 32 channels, half of the 32 tiles
    // fragment addresses: row r = lane & 15 (+ 16 mh for A), k group g = lane >> 4 (8 bf16 = 16 B), 64-byte rows per piece, 16-byte
    // positions XOR-swizzled with (r >> 2) & 3 so that 16 consecutive rows cover the 64 banks once
    const int r = lane & 15, g = lane >> 4;
    const int a_off = ((mh * 16 + r) * 64) + ((g ^ ((r >> 2) & 3)) << 4);                  // + piece * 2048
    const int b_off = 3 * 2048 + ((nb * 32 + r) * 64) + ((g ^ ((r >> 2) & 3)) << 4);      // + piece * 8192 + sub * 16 * 64
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        const unsigned char* st = smem_b + (s & 3) * STAGE;
        bf16x8 a[3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(st + a_off + p * 2048);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[sub][p] = *reinterpret_cast<const bf16x8*>(st + b_off + p * 8192 + sub * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[sub][0], acc[sub], 0, 0, 0);   // small terms first
            acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[sub][2], acc[sub], 0, 0, 0);
            acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[sub][1], acc[sub], 0, 0, 0);
            acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[sub][0], acc[sub], 0, 0, 0);
            acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[sub][1], acc[sub], 0, 0, 0);
            acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[sub][0], acc[sub], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = acc[0][0] + acc[0][1] + acc[1][2] + acc[1][3];
    if (tid == 0) g_cyc[blockIdx.x] = t1 - t0;
}

// ---- today's loop for the same GEMM: wave = (position pp of two, n block): here one position per stage for a like-for-like count
template <int FILL>
__global__ __launch_bounds__(768) void k_f32(float* out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 30 * 1024; i += blockDim.x) smem[i] = (float)((i * 37 + blockIdx.x) & 255) * 1e-3f - 0.1f;
    __syncthreads();
    constexpr int STAGE = 10 * 1024;                       // floats: two positions x (V 1024 + U 4096)
    if (wid >= 8) {
        const f32x4 v = {1.f, 2.f, 3.f, 4.f};
        for (int s = 0; s < stages; ++s) {
            float* dst = smem + ((s + 2) % 3) * STAGE + (tid - 512) * 4;
#pragma unroll
            for (int j = 0; j < FILL; ++j) *reinterpret_cast<f32x4*>(dst + (j * 1024) % (STAGE - 1024)) = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    const int pp = wid & 1, nb = wid >> 1;
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    int foff[4];
    for (int q = 0; q < 4; ++q) foff[q] = (lane & 31) * 32 + (((2 * q + fh) ^ fsw) << 2);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        const float* As = smem + (s % 3) * STAGE + pp * 1024;
        const float* Bs = smem + (s % 3) * STAGE + 2048 + (pp * 4 + nb) * 1024;
        float4 a[2], b[2];
        a[0] = *reinterpret_cast<const float4*>(As + foff[0]);
        b[0] = *reinterpret_cast<const float4*>(Bs + foff[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            if (q < 3) {
                a[nxt] = *reinterpret_cast<const float4*>(As + foff[q + 1]);
                b[nxt] = *reinterpret_cast<const float4*>(Bs + foff[q + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].x, b[cur].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].y, b[cur].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].z, b[cur].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur].w, b[cur].w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int i = 0; i < 16; ++i) sum += acc[i];
    out[blockIdx.x * 512 + tid] = sum;
    if (tid == 0) g_cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static double run(K kern, float* out, size_t lds, int stages, const char* what, double positions_per_stage) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<256, 768, lds>>>(out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    kern<<<256, 768, lds>>>(out, stages);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(256);
    (void)hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_cyc), 256 * 8);
    std::sort(c.begin(), c.end());
    const double per_pos = (double)c[128] / stages / positions_per_stage;
    printf("%-78s %7.1f cycles per position and workgroup (clock %.2f GHz)\n", what, per_pos, c[128] / (ms * 1e6));
    return per_pos;
}

// ---- numerics on the device: C = A B (16 x 16, K = 32) from fp32 data: six bf16 products vs fp64
__global__ void k_num(const float* A, const float* B, float* C, int trunc) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    bf16x8 ap[3], bp[3];
    for (int j = 0; j < 8; ++j) {
        float x = A[r * 32 + g * 8 + j], y = B[(g * 8 + j) * 16 + r];
        for (int p = 0; p < 3; ++p) {
            __bf16 hx, hy;
            if (trunc) {
                unsigned ux = __float_as_uint(x) & 0xFFFF0000u, uy = __float_as_uint(y) & 0xFFFF0000u;
                float fx = __uint_as_float(ux), fy = __uint_as_float(uy);
                hx = (__bf16)fx; hy = (__bf16)fy;
                x -= fx; y -= fy;
            } else {
                hx = (__bf16)x; hy = (__bf16)y;
                x -= (float)hx; y -= (float)hy;
            }
            ap[p][j] = hx; bp[p][j] = hy;
        }
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], bp[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[0], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = acc[i];      // D[row 4 g + i][column r]
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 768 * 4);
    const int stages = 4000;
    printf("matrix-wave loop of one Winograd position GEMM (32 tiles x 32 channels x 128 n per workgroup), 8 matrix + 4 filling waves:\n");
    const double f0 = run(k_f32<0>, out, 120 * 1024, stages, "fp32 MFMA 32x32x2, operands from LDS, no fills", 2.0);
    const double f5 = run(k_f32<5>, out, 120 * 1024, stages, "fp32 MFMA 32x32x2, + 20 KB of LDS fills per position", 2.0);
    const double s0 = run(k_split<0>, out, 120 * 1024, stages, "six bf16 MFMA 16x16x32 per product block, operands from LDS, no fills", 1.0);
    const double s8 = run(k_split<8>, out, 120 * 1024, stages, "six bf16 MFMA 16x16x32 per product block, + 32 KB of LDS fills per position", 1.0);
    printf("ratio fp32 / split: %.2f without fills, %.2f with fills\n", f0 / s0, f5 / s8);

    std::vector<float> A(16 * 32), B(32 * 16), C(256);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.2f;
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int trunc = 0; trunc < 2; ++trunc) {
        k_num<<<1, 64>>>(dA, dB, dC, trunc);
        (void)hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
        double emax = 0, rmax = 0, e32 = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0; float f = 0;
                for (int k = 0; k < 32; ++k) { ref += (double)A[i * 32 + k] * B[k * 16 + j]; f = fmaf(A[i * 32 + k], B[k * 16 + j], f); }
                emax = std::max(emax, std::fabs(C[i * 16 + j] - ref)); rmax = std::max(rmax, std::fabs(ref)); e32 = std::max(e32, std::fabs((double)f - ref));
            }
        printf("numerics, 16 x 16 x 32 product, pieces by %s: max |split - fp64| / max |fp64| = %.2e   (sequential fp32 FMA: %.2e)\n",
               trunc ? "truncation" : "round-to-nearest", emax / rmax, e32 / rmax);
    }
    return 0;
}
