// L1 loss of the SR training step (the SR network and its loss are build-defined; the reference has no SR model) -- value and
// gradient in ONE pass over the two HR-sized tensors:  loss = mean |a - b|,  grad = sign(a - b) / n  (the gradient w.r.t. a for an upstream
// gradient of 1; kmsr_scale_by_device_scalar_f32 rescales it in place when the upstream gradient is something else).  torch's l1_loss +
// autograd walk the 0.67 GB config-3 output five times (sub, abs, mean, sign, mul: 1.1 ms); this is 3 tensor passes (0.45 ms).  HBM-bound.
// Deterministic: per-block partial sums in a fixed grid, added in a fixed order by the second launch.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kLossBlocks = 2048;

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }     // sign(0) = 0 (torch.sign); a NaN difference gives 0 here, NaN in torch -- the loss value itself is NaN either way

__global__ void __launch_bounds__(256) l1_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float inv_n,
                                                      float* __restrict__ grad, float* __restrict__ partial) {
    const long long n4 = n >> 2;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
        const f32x4 d = x - y;
        acc += (fabsf(d.x) + fabsf(d.y)) + (fabsf(d.z) + fabsf(d.w));
        if (grad) reinterpret_cast<f32x4*>(grad)[i] = f32x4{sgn(d.x) * inv_n, sgn(d.y) * inv_n, sgn(d.z) * inv_n, sgn(d.w) * inv_n};
    }
    if (blockIdx.x == 0) {                                   // the n % 4 tail
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float d = a[i] - b[i];
            acc += fabsf(d);
            if (grad) grad[i] = sgn(d) * inv_n;
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) l1_loss_finish_kernel(const float* __restrict__ partial, int nblocks, float inv_n, float* __restrict__ loss) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * inv_n;
}

__global__ void __launch_bounds__(256) scale_by_device_scalar_kernel(float* __restrict__ x, long long n, const float* __restrict__ scale) {
    const float s = *scale;
    if (s == 1.f) return;                                    // block-uniform: the usual case (loss.backward() with no upstream factor) costs a launch
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        reinterpret_cast<f32x4*>(x)[i] = v * s;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) x[i] *= s;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_l1_loss_workspace_floats(void) { return kLossBlocks; }

extern "C" int kmsr_l1_loss_f32(const float* a, const float* b, int64_t n, float* loss, float* grad, float* workspace, void* stream_) {
    if (!a || !b || !loss || !workspace) return fail(KMSR_E_BADARG, "l1_loss: null pointer");
    if (n <= 0) return fail(KMSR_E_BADARG, "l1_loss: n = %lld", (long long)n);
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(grad)) & 15)
        return fail(KMSR_E_UNSUPPORTED, "l1_loss: a, b and grad must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int blocks = (int)std::min<long long>(((n >> 2) + 255) / 256 + 1, kLossBlocks);
    const float inv_n = (float)(1.0 / (double)n);
    l1_loss_kernel<<<blocks, 256, 0, stream>>>(a, b, (long long)n, inv_n, grad, workspace);
    KMSR_HIP_CHECK(hipGetLastError());
    l1_loss_finish_kernel<<<1, 256, 0, stream>>>(workspace, blocks, inv_n, loss);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_scale_by_device_scalar_f32(float* x, int64_t n, const float* scale, void* stream_) {
    if (!x || !scale) return fail(KMSR_E_BADARG, "scale_by_device_scalar: null pointer");
    if (n <= 0) return fail(KMSR_E_BADARG, "scale_by_device_scalar: n = %lld", (long long)n);
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail(KMSR_E_UNSUPPORTED, "scale_by_device_scalar: x must be 16-byte aligned");
    const int blocks = (int)std::min<long long>(((n >> 2) + 255) / 256 + 1, 4096);
    scale_by_device_scalar_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream_)>>>(x, (long long)n, scale);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
