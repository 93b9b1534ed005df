// host_sanitize.h -- the `make asan` build of libddk's HOST half (DDK_HOST_SANITIZE; SURVEY.md section 5, "ASAN build of the host shim").
//
// The plan builder, the choosers, the workspace carve-up and the graph / table caches are ~3000 lines of host C++ doing pointer
// arithmetic over caller-owned arenas; a slip there is a silent device-memory overrun.  GPU sanitizers are not available on this
// pool, so this build checks the host half on a machine WITHOUT a GPU: every file is compiled by hipcc as usual, with
// -fsanitize=address,undefined on the host side, and the macros below reroute
//   * every kernel launch to ddk::san::launch(): nothing runs; the launch geometry is validated and every pointer argument -- also
//     the pointer-sized words of by-value parameter structs -- must lie inside an arena the test registered (ddk_san_register);
//   * the few runtime calls the host half makes to checked stand-ins that work on host memory (memset / memcpy with range checks,
//     fixed device attributes of a whole MI355X).
// Entry points with known tensor extents (conv_forward, the GroupNorm launchers) additionally check base + extent (san::extent).
// tests/host/plan_walk.cpp drives it: plans, packing, one forward and a short eager sampler chain for all five BASELINE
// configurations (tests/test_host_sanitize.py, -m "not gpu").  Nothing of this exists in libddk.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

namespace ddk {
namespace san {

void fail(const char* fmt, ...);                       // counts an error, keeps the first message (core.hip)
bool inside(const void* p, size_t bytes);              // [p, p + bytes) lies in ONE registered arena
bool near_arenas(uint64_t v);                          // v is within 1 GiB of the registered address range
void count_launch();

inline void extent(const char* what, const void* p, long long bytes) {
    if (p && bytes > 0 && !inside(p, (size_t)bytes)) fail("%s: [%p, +%lld) leaves its arena", what, p, bytes);
}

template <class T>
void check_arg(const char* kernel, int idx, const T& v) {
    if constexpr (std::is_pointer_v<T>) {
        if (v != nullptr && !inside(reinterpret_cast<const void*>(v), 1)) fail("%s: pointer argument %d = %p is outside every arena", kernel, idx, (const void*)v);
    } else if constexpr (std::is_class_v<T> && std::is_trivially_copyable_v<T>) {
        // by-value parameter block: every aligned 8-byte word that LOOKS like an address near the arenas must be inside one
        unsigned char raw[sizeof(T)];
        std::memcpy(raw, &v, sizeof(T));
        for (size_t o = 0; o + 8 <= sizeof(T); o += 8) {
            uint64_t w;
            std::memcpy(&w, raw + o, 8);
            if (w != 0 && near_arenas(w) && !inside(reinterpret_cast<const void*>(w), 1))
                fail("%s: parameter block argument %d, word at byte %zu = %#llx points outside every arena", kernel, idx, o, (unsigned long long)w);
        }
    }
}

template <class... A>
void launch(const char* kernel, dim3 grid, dim3 block, size_t lds, hipStream_t, const A&... args) {
    count_launch();
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || grid.y > 65535u || grid.z > 65535u) fail("%s: grid %u x %u x %u", kernel, grid.x, grid.y, grid.z);
    if (threads == 0 || threads > 1024) fail("%s: %llu threads per workgroup", kernel, threads);
    if (lds > 160u * 1024u) fail("%s: %zu bytes of dynamic LDS", kernel, lds);
    int idx = 0;
    (check_arg(kernel, idx++, args), ...);
}

inline hipError_t memset_async(void* p, int v, size_t n) {
    if (!inside(p, n)) { fail("hipMemsetAsync: [%p, +%zu) leaves its arena", p, n); return hipSuccess; }
    std::memset(p, v, n);
    return hipSuccess;
}
inline hipError_t memcpy_async(void* d, const void* s, size_t n, hipMemcpyKind kind) {
    const bool dev_src = kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice;
    const bool dev_dst = kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice;
    if ((dev_src && !inside(s, n)) || (dev_dst && !inside(d, n))) { fail("hipMemcpyAsync: a device range of %zu bytes leaves its arena", n); return hipSuccess; }
    std::memmove(d, s, n);
    return hipSuccess;
}
inline hipError_t memcpy2d_async(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    if (height == 0 || width == 0) return hipSuccess;
    if (!inside(d, dpitch * (height - 1) + width) || !inside(s, spitch * (height - 1) + width)) {
        fail("hipMemcpy2DAsync: a range of %zu rows x %zu bytes leaves its arena", height, width);
        return hipSuccess;
    }
    for (size_t r = 0; r < height; ++r) std::memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
    return hipSuccess;
}
inline hipError_t dev_attr(int* v, hipDeviceAttribute_t a) {
    *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : a == hipDeviceAttributeNumberOfXccs ? 8 : 0;
    return hipSuccess;
}

}  // namespace san
}  // namespace ddk

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    ::ddk::san::launch(#kernel, dim3(grid), dim3(block), (size_t)(lds), (hipStream_t)(stream), ##__VA_ARGS__)
#define hipMemsetAsync(p, v, n, s) ::ddk::san::memset_async((p), (v), (n))
#define hipMemcpyAsync(d, s_, n, kind, st) ::ddk::san::memcpy_async((d), (s_), (n), (kind))
#define hipMemcpy2DAsync(d, dp, s_, sp, w, h, kind, st) ::ddk::san::memcpy2d_async((d), (dp), (s_), (sp), (w), (h), (kind))
#define hipGetDevice(pd) (*(pd) = 0, hipSuccess)
#define hipGetDeviceCount(pn) (*(pn) = 1, hipSuccess)
#define hipFuncSetAttribute(f, a, v) (hipSuccess)
#define hipGetLastError() (hipSuccess)
#define hipDeviceGetAttribute(pv, attr, dev) ::ddk::san::dev_attr((pv), (attr))
#define hipDeviceSynchronize() (hipSuccess)
#define hipStreamSynchronize(s) (hipSuccess)
#define hipMemcpyFromSymbol(...) (hipSuccess)
#define hipMemcpyToSymbol(...) (hipSuccess)
