// core.hip -- error reporting, version and device probe of libddk.so.
#include <cstring>

#include "ddk_internal.h"

namespace ddk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ddk

extern "C" int ddk_version(void) { return 100; }  // 0.1.0

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
