// Small-channel k x k convolution with reflect padding and per-unit weights (semantics: include/kmsr.h).
// One workgroup = one 16x16 output tile of one unit; the reflect-padded input tile of ALL input channels sits in LDS
// (<= 32 x 22 x 22 floats), each thread owns one pixel and all COUT output channels; weights are wave-uniform
// ([ci][tap][co] contiguous per unit) so they arrive through the scalar cache and feed v_fmac as SGPR operands.
#include "kmsr_common.h"

namespace kmsr {
namespace {

constexpr int RT = 16;   // output tile edge

__device__ __forceinline__ int reflect_idx(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}

template <int KS, int COUT>
__global__ void __launch_bounds__(256) conv_reflect_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ out, int Cin, int H, int W, int tiles_x,
                                                           int tiles) {
    constexpr int P = KS / 2, IT = RT + KS - 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [Cin][IT][IT]
    const int tid = threadIdx.x;
    const int u = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int ty0 = (t / tiles_x) * RT, tx0 = (t % tiles_x) * RT;
    const float* __restrict__ xu = x + (long long)u * Cin * H * W;
    // 8 loads in flight per thread before any is written to LDS (a rolled load->store loop pays a round trip per element)
    const int n_el = Cin * IT * IT;
    for (int e0 = 0; e0 < n_el; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = min(e0 + tid + 256 * q, n_el - 1);
            const int c = e / (IT * IT), rem = e - c * (IT * IT), r = rem / IT, cc = rem - r * IT;
            const int gy = reflect_idx(min(ty0 + r - P, H - 1 + P), H), gx = reflect_idx(min(tx0 + cc - P, W - 1 + P), W);
            v[q] = xu[((long long)c * H + gy) * W + gx];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + tid + 256 * q;
            if (e < n_el) tile[e] = v[q];
        }
    }
    __syncthreads();
    const int py = tid / RT, px = tid % RT;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    const float* __restrict__ wu = w + (long long)u * Cin * KS * KS * COUT;    // wave-uniform
    for (int c = 0; c < Cin; ++c) {
        const float* tc = tile + c * IT * IT + py * IT + px;
#pragma unroll
        for (int a = 0; a < KS; ++a)
#pragma unroll
            for (int b = 0; b < KS; ++b) {
                const float v = tc[a * IT + b];
                const float* __restrict__ wt = wu + ((long long)c * KS * KS + a * KS + b) * COUT;
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(wt[o], v, acc[o]);
            }
    }
    const int y = ty0 + py, xx = tx0 + px;
    if (y < H && xx < W) {
        float* ou = out + (long long)u * COUT * H * W + (long long)y * W + xx;
#pragma unroll
        for (int o = 0; o < COUT; ++o) ou[(long long)o * H * W] = acc[o];
    }
}

template <int KS, int COUT>
int launch(const float* x, const float* w, int U, int Cin, int H, int W, float* out, hipStream_t st) {
    const int tiles_x = ceil_div(W, RT), tiles = tiles_x * ceil_div(H, RT);
    const size_t shmem = (size_t)Cin * (RT + KS - 1) * (RT + KS - 1) * sizeof(float);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_reflect_kernel<KS, COUT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    conv_reflect_kernel<KS, COUT><<<U * tiles, 256, shmem, st>>>(x, w, out, Cin, H, W, tiles_x, tiles);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_conv_reflect_f32(const float* x, const float* w, int32_t U, int32_t Cin, int32_t Cout, int32_t H,
                                     int32_t W, int32_t k, float* out, void* stream) {
    if (!x || !w || !out) return fail(KMSR_E_BADARG, "conv_reflect: null pointer");
    if (U <= 0 || Cin <= 0 || Cin > 32 || H <= k / 2 || W <= k / 2) return fail(KMSR_E_BADARG, "conv_reflect: bad shape");
    if ((long long)U * ceil_div(H, RT) * ceil_div(W, RT) > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv_reflect: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define KMSR_CR(K_, C_) if (k == K_ && Cout == C_) return launch<K_, C_>(x, w, U, Cin, H, W, out, st);
    KMSR_CR(1, 1) KMSR_CR(1, 32) KMSR_CR(3, 1) KMSR_CR(3, 32) KMSR_CR(5, 1) KMSR_CR(5, 32) KMSR_CR(7, 1) KMSR_CR(7, 32)
#undef KMSR_CR
    return fail(KMSR_E_UNSUPPORTED, "conv_reflect: k=%d Cout=%d not instantiated (k in 1,3,5,7; Cout in 1,32)", k, Cout);
}
