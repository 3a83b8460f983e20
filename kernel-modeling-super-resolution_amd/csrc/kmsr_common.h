// Shared host-side helpers for libkmsr_hip.so (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/kmsr.h"

namespace kmsr {

void set_error(const char* fmt, ...);  // thread-local message store (api.hip)

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

#define KMSR_HIP_CHECK(expr)                                                                          \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return ::kmsr::fail(KMSR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                __LINE__);                                                            \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace kmsr
