// tools/chain_hop.hip -- what does ONE dependency hop cost at the geometry of the 4x4 level of the reverse step, as a kernel
// boundary and inside one persistent launch?  (round 6, verdict item 1: "persistent image-group kernel per resolution level")
//
// Geometry = conv3x3_gn_local_kernel's: 256 workgroups of 512 threads = 32 images x 8 channel slices.  Per hop every workgroup
// reads its image's whole activation (16 pixels x 256 channels = 16 KB, produced by the image's 8 workgroups in the hop before),
// optionally streams `stream_kb` of weights from L2 and idles `work` x 64 cycles (the k loop's stand-in), and writes its own
// 16 x 32 slice (2 KB).  Every word of an image equals the hop index, every reader checks every word: a stale or torn read counts.
//
//   L  one launch per hop (N launches in one hipGraph, plain loads / stores)          -- what the step does today
//   C  one persistent launch: sc1 stores -> vmcnt(0) -> barrier -> lane-0 agent atomic add; lane-0 sc1 poll -> barrier -> sc1 loads
//      (row 1 of MI355X_MICROARCH.md's hand-off table; the form conv_wino2's in-launch GroupNorm uses)
//   S  one persistent launch: three rotating slots armed with a NaN-payload sentinel; producers just store (sc1), consumers load
//      (sc1) and retry the 16-byte pieces that still hold a sentinel word; a slot is re-armed by its owner one hop after its use
//
// placement: 0 = an image's 8 workgroups on 8 different XCDs (block = image * 8 + slice: the slice's weights stay in one L2, as in
// the product kernels), 1 = on one XCD (block = slice * 32 + image).
//
//   hipcc -O3 --offload-arch=gfx950 tools/chain_hop.hip -o /tmp/chain_hop && /tmp/chain_hop [hops=200] [work=0] [stream_kb=0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int B = 32, NT = 8, ROWS = 16, C = 256, IMG = ROWS * C;      // floats per image
constexpr unsigned SENT = 0x7FC0DDC5u;                                    // a quiet NaN no arithmetic produces

struct Args {
    float* buf;            // L, C: 2 x [B][IMG]; S: 3 x [B][IMG]
    unsigned* cnt;         // [B] x 32 words apart (own 128-byte line each)
    const float4* wts;     // weights stand-in, [NT][stream bytes]
    unsigned* err;         // mismatching words seen
    float* sink;
    int hops, work, stream16, placement, hop0;   // stream16: float4 loads per thread and hop
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld2_sc1(const float* p0, const float* p1, float4& a, float4& b) {
    f32x4 x, y;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x), "=&v"(y) : "v"(p0), "v"(p1) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w);
    b = make_float4(y.x, y.y, y.z, y.w);
}
__device__ __forceinline__ void st_sc1(float* p, float4 v) {
    const f32x4 x{v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ unsigned ld_u32_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ void decode(const Args& a, int& b, int& nt) {
    if (a.placement == 0) { b = blockIdx.x / NT; nt = blockIdx.x % NT; }
    else { nt = blockIdx.x / B; b = blockIdx.x % B; }
}

__device__ __forceinline__ float body(const Args& a, int nt, int tid) {      // the k loop's stand-in: L2 weight stream + idle cycles
    float acc = 0.f;
    const float4* w = a.wts + (size_t)nt * a.stream16 * 512 + tid;
    for (int i = 0; i < a.stream16; i += 4) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = w[(size_t)(i + j < a.stream16 ? i + j : a.stream16 - 1) * 512];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].w;
    }
    for (int i = 0; i < a.work; ++i) __builtin_amdgcn_s_sleep(1);
    return acc;
}

__device__ __forceinline__ int bad_words(float4 v, float want) { return (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want); }

// ---------------------------------------------------------------------------------------------------- L: one launch per hop
__global__ __launch_bounds__(512) void hop_launch_kernel(const Args a) {
    int b, nt;
    decode(a, b, nt);
    const int tid = threadIdx.x, h = a.hop0;
    const float* src = a.buf + (size_t)(h & 1) * B * IMG + (size_t)b * IMG;
    float* dst = a.buf + (size_t)((h + 1) & 1) * B * IMG + (size_t)b * IMG;
    const float4 v0 = *reinterpret_cast<const float4*>(src + tid * 4), v1 = *reinterpret_cast<const float4*>(src + 2048 + tid * 4);
    const int bad = bad_words(v0, (float)h) + bad_words(v1, (float)h);
    if (bad) atomicAdd(a.err, bad);
    const float acc = body(a, nt, tid);
    if (acc == 12345.678f) a.sink[0] = acc;
    if (tid < 128) {
        const float f = (float)(h + 1);
        *reinterpret_cast<float4*>(dst + (tid >> 3) * C + nt * 32 + (tid & 7) * 4) = make_float4(f, f, f, f);
    }
}

// ---------------------------------------------------------------------------------------------------- C: counter hand-off
__global__ __launch_bounds__(512) void hop_counter_kernel(const Args a) {
    int b, nt;
    decode(a, b, nt);
    const int tid = threadIdx.x;
    unsigned* cnt = a.cnt + b * 32;
    for (int h = 0; h < a.hops; ++h) {
        if (h > 0) {
            if (tid == 0) {
                long long t0 = __builtin_amdgcn_s_memrealtime();
                while (ld_u32_sc1(cnt) < (unsigned)(8 * h)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000LL) { atomicAdd(a.err, 1u << 20); break; }   // 2 s: give up, never hang
                }
            }
            __syncthreads();
        }
        const float* src = a.buf + (size_t)(h & 1) * B * IMG + (size_t)b * IMG;
        float* dst = a.buf + (size_t)((h + 1) & 1) * B * IMG + (size_t)b * IMG;
        float4 v0, v1;
        ld2_sc1(src + tid * 4, src + 2048 + tid * 4, v0, v1);
        const int bad = bad_words(v0, (float)h) + bad_words(v1, (float)h);
        if (bad) atomicAdd(a.err, bad);
        const float acc = body(a, nt, tid);
        if (acc == 12345.678f) a.sink[0] = acc;
        if (tid < 128) {
            const float f = (float)(h + 1);
            st_sc1(dst + (tid >> 3) * C + nt * 32 + (tid & 7) * 4, make_float4(f, f, f, f));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------- S: sentinel slots
__global__ __launch_bounds__(512) void hop_sentinel_kernel(const Args a) {
    int b, nt;
    decode(a, b, nt);
    const int tid = threadIdx.x;
    const float sentf = __builtin_bit_cast(float, SENT);
    for (int h = 0; h < a.hops; ++h) {
        const float* src = a.buf + (size_t)(h % 3) * B * IMG + (size_t)b * IMG;
        float* dst = a.buf + (size_t)((h + 1) % 3) * B * IMG + (size_t)b * IMG;
        float* arm = a.buf + (size_t)((h + 2) % 3) * B * IMG + (size_t)b * IMG;
        float4 v0, v1;
        long long t0 = 0;
        for (int tries = 0;; ++tries) {
            ld2_sc1(src + tid * 4, src + 2048 + tid * 4, v0, v1);
            const bool armed = __builtin_bit_cast(unsigned, v0.x) == SENT || __builtin_bit_cast(unsigned, v0.y) == SENT ||
                               __builtin_bit_cast(unsigned, v0.z) == SENT || __builtin_bit_cast(unsigned, v0.w) == SENT ||
                               __builtin_bit_cast(unsigned, v1.x) == SENT || __builtin_bit_cast(unsigned, v1.y) == SENT ||
                               __builtin_bit_cast(unsigned, v1.z) == SENT || __builtin_bit_cast(unsigned, v1.w) == SENT;
            if (!__builtin_amdgcn_ballot_w64(armed)) break;          // the wave moves on together (its loads are one instruction anyway)
            if (tries == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(1);
            if ((tries & 63) == 63 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000LL) { if ((tid & 63) == 0) atomicAdd(a.err, 1u << 20); break; }
        }
        const int bad = bad_words(v0, (float)h) + bad_words(v1, (float)h);
        if (bad) atomicAdd(a.err, bad);
        __syncthreads();                 // (the product kernel's "image staged in LDS" barrier) every wave of this workgroup has its rows
        // everybody has written hop h's input, hence read hop h - 1's: the slot of hop h - 1 (= the slot of hop h + 2) is free; its owner re-arms
        if (tid < 128) st_sc1(arm + (tid >> 3) * C + nt * 32 + (tid & 7) * 4, make_float4(sentf, sentf, sentf, sentf));
        const float acc = body(a, nt, tid);
        if (acc == 12345.678f) a.sink[0] = acc;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the re-arm has landed before this hop's output can be seen anywhere
        if (tid < 128) {
            const float f = (float)(h + 1);
            st_sc1(dst + (tid >> 3) * C + nt * 32 + (tid & 7) * 4, make_float4(f, f, f, f));
        }
    }
}

__global__ void fill_kernel(float* p, size_t n, float v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v;
}

int main(int argc, char** argv) {
    const int hops = argc > 1 ? atoi(argv[1]) : 200;
    const int work = argc > 2 ? atoi(argv[2]) : 0;
    const int stream_kb = argc > 3 ? atoi(argv[3]) : 0;
    const int stream16 = stream_kb * 1024 / (512 * 16);
    Args a{};
    CK(hipMalloc(&a.buf, sizeof(float) * 3 * B * IMG));
    CK(hipMalloc(&a.cnt, sizeof(unsigned) * B * 32));
    CK(hipMalloc(&a.err, 4));
    CK(hipMalloc(&a.sink, 4));
    float4* w = nullptr;
    CK(hipMalloc(&w, sizeof(float4) * 512 * (size_t)(stream16 > 0 ? stream16 : 1) * NT));
    CK(hipMemset(w, 0, sizeof(float4) * 512 * (size_t)(stream16 > 0 ? stream16 : 1) * NT));
    a.wts = w;
    a.hops = hops; a.work = work; a.stream16 = stream16;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float sentf = [] { float f; unsigned u = SENT; memcpy(&f, &u, 4); return f; }();
    printf("hops %d, work %d x 64 cycles, weight stream %d KB per workgroup and hop\n", hops, work, stream_kb);
    for (int placement = 0; placement < 2; ++placement) {
        a.placement = placement;
        const char* pn = placement ? "image on ONE XCD " : "image over 8 XCDs";
        auto reset = [&](int slots) {
            CK(hipMemsetAsync(a.err, 0, 4, st));
            CK(hipMemsetAsync(a.cnt, 0, sizeof(unsigned) * B * 32, st));
            hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, st, a.buf, (size_t)B * IMG, 0.f);
            hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, st, a.buf + (size_t)B * IMG, (size_t)(slots - 1) * B * IMG, slots == 3 ? sentf : -1.f);
        };
        auto report = [&](const char* name, float ms, int final_slot) {
            unsigned err = 0;
            CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
            std::vector<float> out((size_t)B * IMG);
            CK(hipMemcpy(out.data(), a.buf + (size_t)final_slot * B * IMG, sizeof(float) * B * IMG, hipMemcpyDeviceToHost));
            size_t wrong = 0;
            for (float f : out) wrong += f != (float)hops;
            printf("  %s  %-44s %8.3f us per hop   (bad words read %u, wrong final words %zu)\n", pn, name, ms * 1e3f / hops, err, wrong);
        };
        for (int rep = 0; rep < 2; ++rep) {
            // L: graph of `hops` launches
            reset(2);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int h = 0; h < hops; ++h) { a.hop0 = h; hipLaunchKernelGGL(hop_launch_kernel, dim3(B * NT), dim3(512), 0, st, a); }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("L one launch per hop (graph)", ms, hops & 1);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            // C
            reset(2);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(hop_counter_kernel, dim3(B * NT), dim3(512), 0, st, a);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("C persistent, counter + sc1 hand-off", ms, hops & 1);
            // S
            reset(3);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(hop_sentinel_kernel, dim3(B * NT), dim3(512), 0, st, a);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("S persistent, sentinel-armed rotating slots", ms, hops % 3);
        }
    }
    return 0;
}
