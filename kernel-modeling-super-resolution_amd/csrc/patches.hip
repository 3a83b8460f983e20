// Water mask + patch cutting (semantics: include/kmsr.h; A_00_patch_cutter_universal.py:89-197).  Pure bandwidth work
// with exact integer reductions: wave ballots / popcounts, one integer atomic per block.
#include "kmsr_common.h"

#include <algorithm>
#include <cmath>

namespace kmsr {
namespace {

__global__ void water_mask_kernel(const float* __restrict__ data, int C, long long HW, int nir, float invalid, float tmin,
                                  float tmax, float* __restrict__ masked, unsigned long long* __restrict__ counts) {
    unsigned long long valid = 0, water = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        float nv = data[(long long)nir * HW + i];
        if (nv == invalid) nv = NAN;
        const bool w = nv >= tmin && nv <= tmax;          // NaN compares false
        valid += !isnan(nv);
        water += w;
        for (int c = 0; c < C; ++c) {
            float v = data[(long long)c * HW + i];
            if (v == invalid) v = NAN;
            masked[(long long)c * HW + i] = w ? v : NAN;
        }
    }
    // exact integer block reduction -> one atomic per block and counter
    __shared__ unsigned long long sv[256], sw[256];
    sv[threadIdx.x] = valid; sw[threadIdx.x] = water;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sv[threadIdx.x] += sv[threadIdx.x + s]; sw[threadIdx.x] += sw[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&counts[0], sv[0]); atomicAdd(&counts[1], sw[0]); }
}

// one block per patch
__global__ void patch_nan_counts_kernel(const float* __restrict__ data, int C, int H, int W, int ps, int stride, int wp,
                                        int* __restrict__ out) {
    const int i = blockIdx.x / wp, j = blockIdx.x % wp;
    const int y0 = i * stride, x0 = j * stride;
    int cnt = 0;
    const int n = C * ps * ps;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int c = e / (ps * ps), rem = e - c * ps * ps, y = rem / ps, x = rem - y * ps;
        cnt += isnan(data[((long long)c * H + y0 + y) * W + x0 + x]) ? 1 : 0;
    }
    __shared__ int sc[256];
    sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sc[threadIdx.x] += sc[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sc[0];
}

__global__ void gather_patches_kernel(const float* __restrict__ data, int C, int H, int W, int ps,
                                      const int* __restrict__ tl, long long total, float* __restrict__ out) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % ps), y = (int)((idx / ps) % ps);
        const int c = (int)((idx / ((long long)ps * ps)) % C), t = (int)(idx / ((long long)ps * ps * C));
        out[idx] = data[((long long)c * H + tl[2 * t] + y) * W + tl[2 * t + 1] + x];
    }
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_water_mask_f32(const float* data, int32_t C, int32_t H, int32_t W, int32_t nir_index,
                                   float invalid_value, float tmin, float tmax, float* masked, uint64_t* counts,
                                   void* stream_) {
    if (!data || !masked || !counts) return fail(KMSR_E_BADARG, "water_mask: null pointer");
    if (C <= 0 || H <= 0 || W <= 0 || nir_index < 0 || nir_index >= C) return fail(KMSR_E_BADARG, "water_mask: bad shape / NIR index");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    KMSR_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), st));
    const long long HW = (long long)H * W;
    const int blocks = (int)std::min<long long>((HW + 255) / 256, 256 * 16);
    water_mask_kernel<<<blocks, 256, 0, st>>>(data, C, HW, nir_index, invalid_value, tmin, tmax, masked,
                                              reinterpret_cast<unsigned long long*>(counts));
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_patch_nan_counts_f32(const float* data, int32_t C, int32_t H, int32_t W, int32_t ps, int32_t stride,
                                         int32_t* nan_counts, void* stream_) {
    if (!data || !nan_counts) return fail(KMSR_E_BADARG, "patch_nan_counts: null pointer");
    if (C <= 0 || ps <= 0 || stride <= 0 || H < ps || W < ps) return fail(KMSR_E_SHAPE, "patch_nan_counts: patch %d does not fit %dx%d (stride %d)", ps, H, W, stride);
    if ((long long)C * ps * ps >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "patch too large");
    const int hp = (H - ps) / stride + 1, wp = (W - ps) / stride + 1;
    patch_nan_counts_kernel<<<hp * wp, 256, 0, static_cast<hipStream_t>(stream_)>>>(data, C, H, W, ps, stride, wp, nan_counts);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_gather_patches_f32(const float* data, int32_t C, int32_t H, int32_t W, int32_t ps,
                                       const int32_t* top_left, int32_t n, float* out, void* stream_) {
    if (n == 0) return KMSR_OK;
    if (!data || !top_left || !out) return fail(KMSR_E_BADARG, "gather_patches: null pointer");
    if (C <= 0 || ps <= 0 || H < ps || W < ps || n < 0) return fail(KMSR_E_SHAPE, "gather_patches: bad shape");
    const long long total = (long long)n * C * ps * ps;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    gather_patches_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream_)>>>(data, C, H, W, ps, top_left, total, out);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
