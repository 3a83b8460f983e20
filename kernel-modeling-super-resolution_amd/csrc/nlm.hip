// Fast-mode non-local means, one band per image (semantics and the PARITY UNPINNED caveat: include/kmsr.h).
// Workgroup = 16x16 output pixels; the reflect-padded neighbourhood (halo 13 before / 14 after: box (-3,+3] plus shifts
// of +-d) sits in LDS; each thread keeps its 6x6 reference patch in registers and walks the (2d+1)^2 shifts.
#include "kmsr_common.h"

#include <algorithm>
#include <cmath>

namespace kmsr {
namespace {

constexpr int NT_ = 16;          // output tile edge
constexpr int OFF = 3;           // patch_size 7 -> offset 3 -> 6x6 box rows (r-3, r+3]

__device__ __forceinline__ int refl(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}

__global__ void __launch_bounds__(256) nlm_kernel(const float* __restrict__ img, int H, int W, int d,
                                                  const float* __restrict__ hh, const float* __restrict__ sg,
                                                  float* __restrict__ out, int tiles_x, int tiles) {
    extern __shared__ float tile[];                       // [TS][TS], TS = 16 + 2d + 6 - 1
    const int lo = d + OFF - 1, TS = NT_ + 2 * d + 2 * OFF - 1;
    const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int ty0 = (t / tiles_x) * NT_, tx0 = (t % tiles_x) * NT_;
    const float* __restrict__ im = img + (long long)b * H * W;
    for (int e0 = 0; e0 < TS * TS; e0 += 256 * 4) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + (int)threadIdx.x + 256 * q, TS * TS - 1);
            const int r = e / TS, c = e - r * TS;
            // clamp first so that positions belonging to pixels beyond the image edge still map to valid addresses
            const int gy = refl(min(max(ty0 - lo + r, -(H - 1)), 2 * (H - 1)), H), gx = refl(min(max(tx0 - lo + c, -(W - 1)), 2 * (W - 1)), W);
            v[q] = im[(long long)gy * W + gx];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + (int)threadIdx.x + 256 * q;
            if (e < TS * TS) tile[e] = v[q];
        }
    }
    __syncthreads();
    const int py = threadIdx.x / NT_, px = threadIdx.x % NT_;
    const float hv = hh[b], sv = sg[b];
    const float inv_h2s2 = 1.0f / (hv * hv * 49.0f), var36 = 36.0f * 2.0f * sv * sv;
    // reference patch: rows py-2 .. py+3, cols px-2 .. px+3 (tile coordinates: + lo)
    float ref[6][6];
    const float* base = tile + (py + lo - 2) * TS + (px + lo - 2);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) ref[a][c] = base[a * TS + c];
    float num = 0.f, den = 0.f;
    for (int tr = -d; tr <= d; ++tr) {
        for (int tc = -d; tc <= d; ++tc) {
            const float* sh = base + tr * TS + tc;
            float D = 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float df = ref[a][c] - sh[a * TS + c];
                    D = fmaf(df, df, D);
                }
            const float dist = fmaxf(D - var36, 0.f) * inv_h2s2;
            if (dist <= 5.0f) {
                const float w = expf(-dist);
                num = fmaf(w, sh[2 * TS + 2], num);       // v[p + t]
                den += w;
            }
        }
    }
    const int y = ty0 + py, x = tx0 + px;
    if (y < H && x < W) out[((long long)b * H + y) * W + x] = num / den;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_nlm_f32(const float* img, int32_t B, int32_t H, int32_t W, int32_t patch_size, int32_t patch_distance,
                            const float* h, const float* sigma, float* out, void* stream) {
    if (!img || !h || !sigma || !out) return fail(KMSR_E_BADARG, "nlm: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return fail(KMSR_E_BADARG, "nlm: bad shape");
    if (patch_size != 7 && patch_size != 6) return fail(KMSR_E_UNSUPPORTED, "nlm: patch_size %d not implemented (7, or 6 which skimage makes 7)", patch_size);
    if (patch_distance < 0 || patch_distance > 20) return fail(KMSR_E_UNSUPPORTED, "nlm: patch_distance %d not in 0..20", patch_distance);
    if (std::min(H, W) <= patch_distance + 4) return fail(KMSR_E_UNSUPPORTED, "nlm: image %dx%d too small for the reflect padding of %d", H, W, patch_distance + 4);
    const int tiles_x = ceil_div(W, NT_), tiles = tiles_x * ceil_div(H, NT_);
    const int TS = NT_ + 2 * patch_distance + 2 * OFF - 1;
    nlm_kernel<<<B * tiles, 256, (size_t)TS * TS * sizeof(float), static_cast<hipStream_t>(stream)>>>(
        img, H, W, patch_distance, h, sigma, out, tiles_x, tiles);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
