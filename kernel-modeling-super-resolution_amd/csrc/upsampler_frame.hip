// Outer frame of the collapsed x4 upsampler (semantics: include/kmsr.h, kmsr_upsampler_frame_f32).
//
// Away from the border  up1 -> PixelShuffle -> up2 -> PixelShuffle -> tail  is one 5x5 convolution on the LR grid followed by
// PixelShuffle(4) (conv5x5_ps4.hip).  On the outermost LR row / column the intermediate layers' zero padding truncates some paths,
// which gives a DIFFERENT linear 5x5 map per border class (top, bottom, left, right, four corners; host side: sr.py reads the 8
// weight sets off the layers' impulse responses).  This kernel evaluates those maps for the 2W + 2(H-2) border LR pixels of
// every image and writes their 4x4 HR blocks into the output of the interior launch.  1.5 GFLOP at config 2: plain VALU code;
// workgroup = 16 border pixels of ONE class x 16 sub-pixels, a thread owns its sub-pixel of every band; the class's weights and
// the pixels' 5x5 patches stream through LDS in chunks of 8 input channels.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int FPX = 16;            // border pixels per workgroup
constexpr int FCK = 8;             // input channels per LDS chunk
constexpr int FMAXB = 8;           // bands
constexpr int FK = FCK * 25;       // 200 (ci, tap) rows per chunk

struct FrameParams {
    int N, F, H, W, bands;
};

// class c: 0 top, 1 bottom, 2 left, 3 right (corners excluded), 4 top-left, 5 top-right, 6 bottom-left, 7 bottom-right
__device__ __forceinline__ int class_pixels(int c, int H, int W) { return c < 2 ? W - 2 : (c < 4 ? H - 2 : 1); }
__device__ __forceinline__ void class_pixel(int c, int t, int H, int W, int& i, int& j) {
    switch (c) {
        case 0: i = 0; j = 1 + t; break;
        case 1: i = H - 1; j = 1 + t; break;
        case 2: i = 1 + t; j = 0; break;
        case 3: i = 1 + t; j = W - 1; break;
        case 4: i = 0; j = 0; break;
        case 5: i = 0; j = W - 1; break;
        case 6: i = H - 1; j = 0; break;
        default: i = H - 1; j = W - 1; break;
    }
}

template <int BANDS>
__global__ void __launch_bounds__(256) upsampler_frame_kernel(const float* __restrict__ h, const float* __restrict__ wcls,
                                                              const float* __restrict__ bcls, float* __restrict__ y,
                                                              FrameParams p) {
    constexpr int CO = 16 * BANDS, BP = BANDS <= 4 ? 4 : 8;            // bands padded to whole 16-byte reads
    __shared__ __attribute__((aligned(16))) float ws[FK * 16 * BP];    // [ci 8][tap 25][sub-pixel 16][band BP]
    __shared__ __attribute__((aligned(16))) float xs[FPX * FK];        // [pixel][ci 8][tap 25]
    const int c = blockIdx.y, tid = threadIdx.x;
    const int H = p.H, W = p.W;
    const int per_img = class_pixels(c, H, W), total = per_img * p.N;
    const int first = blockIdx.x * FPX;
    if (first >= total) return;                    // block-uniform
    const int pl = tid >> 4, q = tid & 15;         // this thread's pixel slot and sub-pixel (a = q >> 2, b = q & 3)
    const int me = first + pl;
    int n = 0, i = 0, j = 0;
    if (me < total) {
        n = me / per_img;
        class_pixel(c, me % per_img, H, W, i, j);
    }
    float acc[BANDS];
#pragma unroll
    for (int b = 0; b < BANDS; ++b) acc[b] = 0.f;
    const float* __restrict__ wc = wcls + (long long)c * p.F * 25 * CO;
    const long long HW = (long long)H * W;
    for (int c0 = 0; c0 < p.F; c0 += FCK) {
        __syncthreads();
        // this chunk of the class's weights (contiguous in [ci][tap][co]): 10 loads in flight per thread before the first LDS store
        // (a rolled load -> store loop pays one L2 round trip per element: 0.39 ms in the first version)
        constexpr int NW = (FK * CO + 255) / 256;
#pragma unroll
        for (int b0 = 0; b0 < NW; b0 += 10) {
            float v[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = tid + 256 * (b0 + u);
                const bool ok = e < FK * CO && c0 + e / (25 * CO) < p.F;
                v[u] = ok ? wc[(long long)c0 * 25 * CO + e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 10; ++u)
            {
                const int e = tid + 256 * (b0 + u);                      // e = k * CO + band * 16 + sub-pixel
                const int k = e / CO, co = e - k * CO;
                if (e < FK * CO) ws[(k * 16 + (co & 15)) * BP + (co >> 4)] = v[u];
            }
        }
        // 5x5 patches (zero outside the image / past the pixel list): the 16 threads of a pixel slot fetch its 200 (ci, tap) values
        const float* __restrict__ hp = h + ((long long)n * p.F + c0) * HW;
#pragma unroll
        for (int t = 0; t < (FK + 15) / 16; ++t) {
            const int l = q + 16 * t;                                    // (ci, tap) index: compile-time divisions only
            if (l < FK) {
                const int ci = l / 25, tap = l - ci * 25;
                const int yy = i + tap / 5 - 2, xx = j + tap % 5 - 2;
                const bool ok = me < total && c0 + ci < p.F && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const float v = hp[ok ? (long long)ci * HW + (long long)yy * W + xx : 0];
                xs[pl * FK + l] = ok ? v : 0.f;
            }
        }
        __syncthreads();
        const f32x4* xp = reinterpret_cast<const f32x4*>(xs + pl * FK);
        const float* wp = ws + q * BP;
#pragma unroll 2
        for (int k4 = 0; k4 < FK / 4; ++k4) {                            // 4 (ci, tap) rows per step: 1 + 4 (or 8) 16-byte LDS reads per 4 BANDS FMAs
            const f32x4 x4 = xp[k4];
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float* wk = wp + (4 * k4 + kk) * 16 * BP;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, 0.f, 0.f, 0.f, 0.f};
                float wv2[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BP == 8) {
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wk + 4);
                    wv2[0] = w1.x; wv2[1] = w1.y; wv2[2] = w1.z; wv2[3] = w1.w;
                }
#pragma unroll
                for (int b = 0; b < BANDS; ++b) acc[b] = fmaf(xv[kk], b < 4 ? wv[b] : wv2[b - 4], acc[b]);
            }
        }
    }
    if (me < total) {
        const long long H4 = 4ll * H, W4 = 4ll * W;
#pragma unroll
        for (int b = 0; b < BANDS; ++b)
            y[(((long long)n * BANDS + b) * H4 + 4 * i + (q >> 2)) * W4 + 4 * j + (q & 3)] = acc[b] + bcls[c * CO + b * 16 + q];
    }
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_upsampler_frame_f32(const float* h, int32_t N, int32_t F, int32_t H, int32_t W, int32_t bands, const float* wcls,
                                        const float* bcls, float* y, void* stream) {
    if (!h || !wcls || !bcls || !y) return fail(KMSR_E_BADARG, "upsampler_frame: null pointer");
    if (N <= 0 || F <= 0 || H < 4 || W < 4) return fail(KMSR_E_BADARG, "upsampler_frame: bad shape (H, W >= 4)");
    if (bands < 1 || bands > FMAXB) return fail(KMSR_E_UNSUPPORTED, "upsampler_frame: 1..%d bands (got %d)", FMAXB, bands);
    FrameParams p{N, F, H, W, bands};
    const int longest = std::max(H, W) - 2;
    const dim3 grid((unsigned)ceil_div(longest * N, FPX), 8);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (bands) {
#define KMSR_FR(B_) case B_: upsampler_frame_kernel<B_><<<grid, 256, 0, st>>>(h, wcls, bcls, y, p); break;
        KMSR_FR(1) KMSR_FR(2) KMSR_FR(3) KMSR_FR(4) KMSR_FR(5) KMSR_FR(6) KMSR_FR(7) KMSR_FR(8)
#undef KMSR_FR
    }
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
