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

// ---- the named legs SURVEY 8(b) lists as the minimum export set: thin forms of the entry points that implement them
extern "C" int kmsr_conv3x3_dgrad_f32(const kmsr_conv3x3_desc* d, const float* dy, const float* packed_wT, float* dx, void* stream) {
    if (!d) return kmsr::fail(KMSR_E_BADARG, "null descriptor");
    if (d->stride != 1 || d->pixel_shuffle) return kmsr::fail(KMSR_E_UNSUPPORTED, "conv3x3 dgrad: stride 1, gradient in the conv-output layout");
    kmsr_conv3x3_desc g = *d;                 // the data gradient is the same convolution from Cout to Cin channels
    g.Cin = d->Cout;
    g.Cout = d->Cin;
    g.relu = 0;
    g.flags = 0;
    return kmsr_conv3x3_fwd_f32(&g, dy, packed_wT, nullptr, nullptr, dx, stream);
}

extern "C" int kmsr_conv3x3_pixelshuffle_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                                                 float* out, void* stream) {
    if (!d) return kmsr::fail(KMSR_E_BADARG, "null descriptor");
    kmsr_conv3x3_desc g = *d;
    g.pixel_shuffle = 1;
    return kmsr_conv3x3_fwd_f32(&g, x, packed_w, bias, nullptr, out, stream);
}

extern "C" int kmsr_conv3x3_pixelshuffle_fwd_bf16(const kmsr_conv3x3_desc* d, const void* x, const void* packed_w, const float* bias,
                                                  void* out, void* stream) {
    if (!d) return kmsr::fail(KMSR_E_BADARG, "null descriptor");
    kmsr_conv3x3_desc g = *d;
    g.pixel_shuffle = 1;
    return kmsr_conv3x3_fwd_bf16(&g, x, packed_w, bias, nullptr, out, stream);
}
