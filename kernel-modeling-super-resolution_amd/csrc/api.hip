// Version / error plumbing of libkmsr_hip.so.
#include "kmsr_common.h"

#include <cstring>

namespace kmsr {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace kmsr

extern "C" int kmsr_version(void) { return KMSR_VERSION; }

extern "C" const char* kmsr_last_error(void) { return kmsr::g_err; }

extern "C" int kmsr_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return kmsr::fail(KMSR_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}
