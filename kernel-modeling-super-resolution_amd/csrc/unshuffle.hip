// Backward of the PixelShuffle(2) that the F -> 4F convs fuse into their store: the gradient arrives in the shuffled layout
// g[N,C,2H,2W] and the data / weight gradient kernels want the conv-output layout gu[N,4C,H,W],
//   gu[n, 4c + 2i + j, y, x] = g[n, c, 2y + i, 2x + j],
// together with the bias gradient sum_{n,y,x} gu[n,k,y,x].  One pass does both (semantics: include/kmsr.h,
// kmsr_pixel_unshuffle2_sum_f32): a thread reads two float4 of an input row (8 columns = both sub-pixel phases of 4 output
// columns), writes one float4 per phase, and keeps 4 running sums; per-block partial sums are written in a fixed order
// (the caller adds the [N * row blocks] partials per channel).
#include "kmsr_common.h"

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int UROWS = 32;     // output rows per block

__global__ void __launch_bounds__(256) pixel_unshuffle2_sum_kernel(const float* __restrict__ g, float* __restrict__ gu,
                                                                   float* __restrict__ part, int C, int H, int W, int rblocks) {
    const int plane = blockIdx.x / rblocks, rb = blockIdx.x % rblocks;         // plane = n * C + c
    const int n = plane / C, c = plane % C;
    const int qw = W / 4;                                                      // float4 per output row
    const long long W2 = 2ll * W;
    const float* __restrict__ gp = g + (long long)plane * (2ll * H) * W2;
    float* __restrict__ op = gu + ((long long)n * 4 * C + 4 * c) * H * W;
    const long long HW = (long long)H * W;
    const int y0 = rb * UROWS, y1 = min(H, y0 + UROWS);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e = threadIdx.x; e < (y1 - y0) * qw; e += 256) {
        const int y = y0 + e / qw, k = e % qw;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4* src = reinterpret_cast<const f32x4*>(gp + (2ll * y + i) * W2 + 8 * k);
            const f32x4 a = src[0], b = src[1];
            const f32x4 p0 = f32x4{a.x, a.z, b.x, b.z}, p1 = f32x4{a.y, a.w, b.y, b.w};
            *reinterpret_cast<f32x4*>(op + (2 * i + 0) * HW + (long long)y * W + 4 * k) = p0;
            *reinterpret_cast<f32x4*>(op + (2 * i + 1) * HW + (long long)y * W + 4 * k) = p1;
            s[2 * i + 0] += (p0.x + p0.y) + (p0.z + p0.w);
            s[2 * i + 1] += (p1.x + p1.y) + (p1.z + p1.w);
        }
    }
    __shared__ float red[4][256];
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) part[((long long)n * rblocks + rb) * 4 * C + 4 * c + threadIdx.x] = red[threadIdx.x][0];
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_pixel_unshuffle2_sum_partials(int32_t N, int32_t C, int32_t H) {
    if (N <= 0 || C <= 0 || H <= 0) return 0;
    return (size_t)N * ceil_div(H, UROWS) * 4 * C;
}

extern "C" int kmsr_pixel_unshuffle2_sum_f32(const float* g, int32_t N, int32_t C, int32_t H, int32_t W, float* gu, float* partials,
                                             void* stream) {
    if (!g || !gu || !partials) return fail(KMSR_E_BADARG, "pixel_unshuffle2_sum: null pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(KMSR_E_BADARG, "pixel_unshuffle2_sum: bad shape");
    if (W % 4 || reinterpret_cast<uintptr_t>(g) % 16 || reinterpret_cast<uintptr_t>(gu) % 16)
        return fail(KMSR_E_UNSUPPORTED, "pixel_unshuffle2_sum: needs W %% 4 == 0 (W=%d) and 16-byte aligned tensors", W);
    const int rblocks = ceil_div(H, UROWS);
    const long long blocks = (long long)N * C * rblocks;
    if (blocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "pixel_unshuffle2_sum: grid too large");
    pixel_unshuffle2_sum_kernel<<<(int)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(g, gu, partials, C, H, W, rblocks);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
