// KernelGAN kernel composition: collapse a bias-free deep-linear conv chain into one K x K kernel per (sample, band).
// Semantics and reference citations: include/kmsr.h.  One workgroup per (sample, band); the running kernel stack
// K_l [co_l][size_l^2] lives in LDS (two ping-pong buffers), layer weights stream from L2 (they are shared by all
// samples; only the per-channel scales differ), every thread owns a strided set of output taps (co, y, x).
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

constexpr int kMaxLayers = 8;
constexpr int kMaxCh = 64;
constexpr int kMaxK = 31;

struct ComposeParams {
    int B, C, n_layers;
    int co[kMaxLayers], ci[kMaxLayers], ks[kMaxLayers];
    int woff[kMaxLayers];   // float offset of layer l inside one band's weight block
    int soff[kMaxLayers];   // float offset of layer l inside one (sample, band) scale block
    int wband, sband;       // floats per band (weights) / per (sample, band) (scales)
    int K, buf_floats;
};

__global__ void __launch_bounds__(256) compose_kernels_kernel(const float* __restrict__ weights,
                                                              const float* __restrict__ scales,
                                                              float* __restrict__ out, ComposeParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cur = smem;
    float* nxt = smem + p.buf_floats;
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int band = blockIdx.x % p.C, b = blockIdx.x / p.C;
    const float* __restrict__ wb = weights + (long long)band * p.wband;
    const float* __restrict__ sb = scales ? scales + ((long long)b * p.C + band) * p.sband : nullptr;

    // layer 0: K_0[co][k0 x k0] = s_0[co] * W_0[co][0]
    int size = p.ks[0];
    for (int e = tid; e < p.co[0] * size * size; e += 256) {
        const int o = e / (size * size);
        cur[e] = wb[p.woff[0] + e] * (sb ? sb[p.soff[0] + o] : 1.f);
    }
    __syncthreads();
    int cmid = p.co[0];
    for (int l = 1; l < p.n_layers; ++l) {
        const int k = p.ks[l], nsize = size + k - 1, nco = p.co[l];
        const float* __restrict__ wl = wb + p.woff[l];
        for (int e = tid; e < nco * nsize * nsize; e += 256) {
            const int o = e / (nsize * nsize), rem = e - o * (nsize * nsize), y = rem / nsize, x = rem - y * nsize;
            const float sc = sb ? sb[p.soff[l] + o] : 1.f;
            // valid tap range: 0 <= y-a < size, 0 <= x-bb < size
            const int a0 = max(0, y - size + 1), a1 = min(k - 1, y);
            const int b0 = max(0, x - size + 1), b1 = min(k - 1, x);
            float acc = 0.f;
            for (int m = 0; m < cmid; ++m) {
                const float* __restrict__ A = wl + ((long long)o * cmid + m) * k * k;
                const float* Bm = cur + m * size * size;
                float part = 0.f;   // one conv_kernel(W[co,cm], K_cur[cm]) (networks.py:181-183), then acc += it (:184)
                for (int a = a0; a <= a1; ++a)
                    for (int bb = b0; bb <= b1; ++bb) part = fmaf(A[a * k + bb] * sc, Bm[(y - a) * size + (x - bb)], part);
                acc += part;
            }
            nxt[e] = acc;
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
        size = nsize;
        cmid = nco;
    }
    // mean over the last layer's maps, clamp, normalise
    const int n = size * size;
    float local = 0.f;
    for (int e = tid; e < n; e += 256) {
        float v = 0.f;
        for (int m = 0; m < cmid; ++m) v += cur[m * n + e];
        v /= (float)cmid;
        v = fmaxf(v, 0.f);
        nxt[e] = v;
        local += v;
    }
    red[tid] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    float ssum = red[0];
    if (ssum <= 1e-12f) ssum = 1.f;
    float* o = out + ((long long)b * p.C + band) * n;
    for (int e = tid; e < n; e += 256) o[e] = nxt[e] / ssum;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_compose_kernels_f32(const float* weights, const float* scales, int32_t B, int32_t C, int32_t n_layers,
                                        const int32_t* co, const int32_t* ci, const int32_t* ks, float* out,
                                        void* stream) {
    if (!weights || !out || !co || !ci || !ks) return fail(KMSR_E_BADARG, "compose: null pointer");
    if (B <= 0 || C <= 0) return fail(KMSR_E_BADARG, "compose: bad batch/band count");
    if (n_layers < 1 || n_layers > kMaxLayers) return fail(KMSR_E_UNSUPPORTED, "compose: n_layers %d not in 1..%d", n_layers, kMaxLayers);
    if (ci[0] != 1) return fail(KMSR_E_UNSUPPORTED, "compose: first layer must have one input channel (got %d)", ci[0]);
    ComposeParams p{};
    p.B = B; p.C = C; p.n_layers = n_layers;
    int woff = 0, soff = 0, K = 1, maxf = 0, size = 1;
    for (int l = 0; l < n_layers; ++l) {
        if (co[l] < 1 || co[l] > kMaxCh || ks[l] < 1 || !(ks[l] & 1)) return fail(KMSR_E_UNSUPPORTED, "compose: layer %d shape", l);
        if (l > 0 && ci[l] != co[l - 1]) return fail(KMSR_E_SHAPE, "compose: layer %d has %d inputs, previous has %d outputs", l, ci[l], co[l - 1]);
        p.co[l] = co[l]; p.ci[l] = ci[l]; p.ks[l] = ks[l];
        p.woff[l] = woff; p.soff[l] = soff;
        woff += co[l] * ci[l] * ks[l] * ks[l];
        soff += co[l];
        K += ks[l] - 1;
        size = K;
        maxf = std::max(maxf, co[l] * size * size);
    }
    if (K > kMaxK) return fail(KMSR_E_UNSUPPORTED, "compose: composed kernel %d > %d", K, kMaxK);
    p.wband = woff; p.sband = soff; p.K = K;
    p.buf_floats = round_up(maxf, 4);
    const size_t shmem = 2 * (size_t)p.buf_floats * sizeof(float);
    if (shmem > 150 * 1024) return fail(KMSR_E_UNSUPPORTED, "compose: %zu B of LDS needed", shmem);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(compose_kernels_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    compose_kernels_kernel<<<B * C, 256, shmem, st>>>(weights, scales, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
