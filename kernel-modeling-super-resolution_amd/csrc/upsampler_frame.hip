// Outer frame of the collapsed x4 upsampler (semantics: include/kmsr.h, kmsr_upsampler_frame_f32).
//
// Away from the border  up1 -> PixelShuffle -> up2 -> PixelShuffle -> tail  is one 5x5 convolution on the LR grid followed by
// PixelShuffle(4) (conv5x5_ps4.hip).  On the outermost LR row / column the intermediate layers' zero padding truncates some paths,
// which gives a DIFFERENT linear 5x5 map per border class (top, bottom, left, right, four corners; host side: sr.py reads the 8
// weight sets off the layers' impulse responses).  This kernel evaluates those maps for the 2W + 2(H-2) border LR pixels of
// every image and writes their 4x4 HR blocks into the output of the interior launch.
//
// Per class a GEMM  D[sub-pixel 16 x band][pixel] = W[k][.]^T x[pixel][k],  k = (input channel, tap) = 25 F, on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32; round 2's VALU kernel spent 0.26 ms per step of config 2 on LDS reads: one 16-byte weight read per 3
// FMAs).  Workgroup = 64 border pixels of ONE class, wave = 16 of them x all bands; K streams through LDS in chunks of 160 rows,
// both operands stored so that one ds_read_b128 feeds FOUR MFMAs:
//   ws[kblk][band][kk][q][s]  = W[16 kblk + 4 s + kk][band * 16 + q]      (A operand of lane (q, kk), steps s = 0..3)
//   xs[kblk][kk][pixel][s]    = x[pixel][16 kblk + 4 s + kk]              (B operand of lane (pixel, kk))
// D comes out with lane (pixel, a) holding the four HR columns of HR row a of its pixel: one 16-byte store per band.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int FPX = 64;            // border pixels per workgroup (16 per wave)
constexpr int FKC = 160;           // K rows (input channel, tap) per LDS chunk = 10 blocks of 16
constexpr int FMAXB = 8;           // bands

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
    constexpr int CO = 16 * BANDS;
    __shared__ __attribute__((aligned(16))) float ws[FKC * CO];         // 160 x 48 floats = 30 KB at 3 bands
    __shared__ __attribute__((aligned(16))) float xs[FKC * FPX];        // 40 KB
    const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int per_img = class_pixels(c, H, W), total = per_img * p.N;
    const int first = blockIdx.x * FPX;
    if (first >= total) return;                    // block-uniform
    const int K = p.F * 25;
    const long long HW = (long long)H * W;
    const float* __restrict__ wc = wcls + (long long)c * K * CO;

    // staging role: pixel slot sm = tid & 63 (coalesced along a top / bottom border), K rows (tid >> 6) + 4 r
    const int sm = tid & 63, sme = first + sm;
    int sn = 0, si = 0, sj = 0;
    if (sme < total) {
        sn = sme / per_img;
        class_pixel(c, sme % per_img, H, W, si, sj);
    }
    const float* __restrict__ hn = h + (long long)sn * p.F * HW;

    f32x4 acc[BANDS];
#pragma unroll
    for (int b = 0; b < BANDS; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += FKC) {
        __syncthreads();
        // ---- weights of this chunk: contiguous [k][co] in global memory; 10 loads in flight per thread before the first LDS store
        constexpr int NW = (FKC * CO + 255) / 256;
#pragma unroll
        for (int b0 = 0; b0 < NW; b0 += 10) {
            float v[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = tid + 256 * (b0 + u);
                const bool ok = b0 + u < NW && e < FKC * CO && k0 + e / CO < K;
                v[u] = ok ? wc[(long long)k0 * CO + e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = tid + 256 * (b0 + u);                      // e = kl * CO + band * 16 + q
                if (b0 + u < NW && e < FKC * CO) {
                    const int kl = e / CO, co = e - kl * CO;
                    ws[((((kl >> 4) * BANDS + (co >> 4)) * 4 + (kl & 3)) * 16 + (co & 15)) * 4 + ((kl >> 2) & 3)] = v[u];
                }
            }
        }
        // ---- 5x5 patches (zero outside the image / past the pixel list / past K): thread (sm, tid >> 6) walks rows kl = (tid >> 6) + 4 r
        {
            int kg = k0 + (tid >> 6);
            int ci = kg / 25, tap = kg - ci * 25;
#pragma unroll
            for (int r0 = 0; r0 < FKC / 4; r0 += 10) {
                float v[10];
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int yy = si + tap / 5 - 2, xx = sj + tap % 5 - 2;
                    const bool ok = sme < total && ci < p.F && yy >= 0 && yy < H && xx >= 0 && xx < W;
                    v[u] = ok ? hn[(long long)ci * HW + (long long)yy * W + xx] : 0.f;
                    tap += 4;
                    if (tap >= 25) { tap -= 25; ++ci; }
                }
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int kl = (tid >> 6) + 4 * (r0 + u);
                    xs[(((kl >> 4) * 4 + (kl & 3)) * FPX + sm) * 4 + ((kl >> 2) & 3)] = v[u];
                }
            }
        }
        __syncthreads();
        // ---- 10 blocks of 16 K rows: one 16-byte read per operand and band, 4 MFMAs each
        const int q = lane & 15, kk = lane >> 4;
#pragma unroll 2
        for (int kb = 0; kb < FKC / 16; ++kb) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + ((kb * 4 + kk) * FPX + wv * 16 + q) * 4);
#pragma unroll
            for (int b = 0; b < BANDS; ++b) {
                const f32x4 wvv = *reinterpret_cast<const f32x4*>(ws + (((kb * BANDS + b) * 4 + kk) * 16 + q) * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv[s], xv[s], acc[b], 0, 0, 0);
            }
        }
    }
    // lane (pixel q of this wave, HR row a = lane >> 4): acc[b][r] = sub-pixel (a, r) of band b
    const int me = first + wv * 16 + (lane & 15), a = lane >> 4;
    if (me < total) {
        const int n = me / per_img;
        int i, j;
        class_pixel(c, me % per_img, H, W, i, j);
        const long long H4 = 4ll * H, W4 = 4ll * W;
#pragma unroll
        for (int b = 0; b < BANDS; ++b) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bcls + c * CO + b * 16 + 4 * a);
            *reinterpret_cast<f32x4*>(y + (((long long)n * BANDS + b) * H4 + 4 * i + a) * W4 + 4 * j) = acc[b] + bv;
        }
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
    if (reinterpret_cast<uintptr_t>(bcls) % 16 || reinterpret_cast<uintptr_t>(y) % 16)
        return fail(KMSR_E_BADARG, "upsampler_frame: class biases and output must be 16-byte aligned");
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
