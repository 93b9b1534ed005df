// core.hip -- error reporting, version and device probe of libddk.so.
#include <atomic>
#include <cstring>
#include <mutex>

#include "ddk_internal.h"
#include "level_chain.h"

namespace ddk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// One-time, per-device setup (kernel attributes for > 64 KB of dynamic LDS).  Keyed by the CURRENT device, so a process
// that drives several GPUs initialises each; thread-safe; must first run outside stream capture (ddk_unet_create and the
// conv / wgrad entry points call it -- any warm-up call does).
int ensure_device_init() {
    static std::atomic<unsigned long long> done{0};
    static std::mutex mu;
    int dev = 0;
    DDK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail_arg("device index out of range");
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return DDK_OK;
    std::lock_guard<std::mutex> lock(mu);
    if (done.load(std::memory_order_relaxed) & bit) return DDK_OK;
    DDK_TRY(conv_init_device());
    DDK_TRY(conv_wino_init_device());
    DDK_TRY(conv_gn_local_init_device());
    DDK_TRY(level_chain_init_device());
    DDK_TRY(conv_first_init_device());
    DDK_TRY(conv1x1_ws_init_device());
    DDK_TRY(conv1x1_sm_init_device());
    DDK_TRY(linattn_small_qkv_init_device());
    DDK_TRY(wgrad_init_device());
    done.fetch_or(bit, std::memory_order_release);
    return DDK_OK;
}
}  // namespace ddk

#ifdef DDK_HOST_SANITIZE
// ---- state of the host-sanitizer build (host_sanitize.h)
#include <string>
#include <vector>
namespace ddk { namespace san {
struct Arena { uintptr_t base; size_t bytes; std::string name; };
static std::vector<Arena> g_arenas;
static long g_launches = 0, g_errors = 0;
static char g_first[512] = "";
void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g_errors++ == 0) std::snprintf(g_first, sizeof(g_first), "%s", buf);
    std::fprintf(stderr, "[ddk host sanitize] %s\n", buf);
}
bool inside(const void* p, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    for (const Arena& r : g_arenas)
        if (a >= r.base && a <= r.base + r.bytes && bytes <= r.base + r.bytes - a) return true;
    return false;
}
bool near_arenas(uint64_t v) {
    if (g_arenas.empty()) return false;
    uintptr_t lo = ~(uintptr_t)0, hi = 0;
    for (const Arena& r : g_arenas) { if (r.base < lo) lo = r.base; if (r.base + r.bytes > hi) hi = r.base + r.bytes; }
    const uintptr_t slack = (uintptr_t)1 << 30;
    return v + slack >= lo && v <= hi + slack;
}
void count_launch() { ++g_launches; }
}}  // namespace ddk::san
extern "C" void ddk_san_register(const void* base, size_t bytes, const char* name) {
    ddk::san::g_arenas.push_back({reinterpret_cast<uintptr_t>(base), bytes, name ? name : ""});
}
extern "C" void ddk_san_clear(void) { ddk::san::g_arenas.clear(); }
extern "C" void ddk_san_stats(long* launches, long* errors) { *launches = ddk::san::g_launches; *errors = ddk::san::g_errors; }
extern "C" const char* ddk_san_first_error(void) { return ddk::san::g_first; }
#endif

// Clock probe: one record {XCC id, s_memtime (shader cycles), s_memrealtime (100 MHz ticks)} per workgroup.  Two probes around a
// timed region give the shader clock the chip held in it: d(memtime) / d(realtime) x 100 MHz, per XCC (the counters of
// different XCCs are not comparable with each other, their rates are).
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + 0] = xcc & 0xfu;
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 3] = 1;
    }
}

extern "C" int ddk_debug_clock_probe(unsigned long long* out, int workgroups, ddk_stream_t s) {
    DDK_REQUIRE(out && workgroups > 0 && workgroups <= 1024, "debug_clock_probe: arguments");
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)workgroups), dim3(64), 0, ddk::as_stream(s), out);
    return ddk::check_launch("clock_probe_kernel");
}

extern "C" int ddk_version(void) { return 400; }  // 0.4.0: ddk_unet_cluster_check & co., cluster workspace layout

extern "C" const char* ddk_last_error(void) { return ddk::g_err; }

extern "C" int ddk_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        ddk::set_error("no HIP device visible");
        return 0;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        ddk::set_error("cannot query HIP device");
        return 0;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ddk::set_error("device is %s, libddk.so is built for gfx950 only", prop.gcnArchName);
        return 0;
    }
    return 1;
}
