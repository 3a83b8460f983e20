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

// acc[o] += w[o] * v for C consecutive (wave-uniform) weights: channel PAIRS go through v_pk_fma_f32 (two fp32 FMAs per lane and
// issue slot; the SGPR pair {w[2q], w[2q+1]} is one operand, v is broadcast to both halves by op_sel), an odd last channel
// through v_fmac_f32.  acc2[q] = channels (2q, 2q+1), accr = channel C-1 when C is odd.
typedef float kmsr_f32x2 __attribute__((ext_vector_type(2)));
template <int C>
__device__ __forceinline__ void fma_row(const float* __restrict__ w, float v, kmsr_f32x2 (&acc2)[(C + 1) / 2], float& accr) {
#pragma unroll
    for (int q = 0; q < C / 2; ++q)
        acc2[q] = __builtin_elementwise_fma(kmsr_f32x2{w[2 * q], w[2 * q + 1]}, kmsr_f32x2{v, v}, acc2[q]);
    if constexpr (C & 1) accr = fmaf(w[C - 1], v, accr);
}
template <int C>
__device__ __forceinline__ float fma_row_get(const kmsr_f32x2 (&acc2)[(C + 1) / 2], float accr, int o) {
    return (C & 1) && o == C - 1 ? accr : (o & 1 ? acc2[o >> 1].y : acc2[o >> 1].x);
}

}  // namespace kmsr
