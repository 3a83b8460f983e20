// Small-channel k x k convolution with reflect (or zero) padding and per-unit (or shared) weights -- forward leg
// (semantics: include/kmsr.h; the data / weight gradient legs are in convk_bwd.hip).
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

template <int KS, int COUT, bool ZERO>
__global__ void __launch_bounds__(256) conv_reflect_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                           int H, int W, int tiles_x, int tiles, long long wstride) {
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
            if constexpr (ZERO) {
                const int gy = ty0 + r - P, gx = tx0 + cc - P;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                v[q] = in ? xu[((long long)c * H + gy) * W + gx] : 0.f;
            } else {
                const int gy = reflect_idx(min(ty0 + r - P, H - 1 + P), H), gx = reflect_idx(min(tx0 + cc - P, W - 1 + P), W);
                v[q] = xu[((long long)c * H + gy) * W + gx];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + tid + 256 * q;
            if (e < n_el) tile[e] = v[q];
        }
    }
    __syncthreads();
    const int py = tid / RT, px = tid % RT;
    kmsr_f32x2 acc2[(COUT + 1) / 2];
    float accr = (COUT & 1) && bias ? bias[COUT - 1] : 0.f;
#pragma unroll
    for (int q = 0; q < COUT / 2; ++q) acc2[q] = bias ? kmsr_f32x2{bias[2 * q], bias[2 * q + 1]} : kmsr_f32x2{0.f, 0.f};
    const float* __restrict__ wu = w + (long long)u * wstride;                 // wave-uniform
    for (int c = 0; c < Cin; ++c) {
        const float* tc = tile + c * IT * IT + py * IT + px;
#pragma unroll
        for (int a = 0; a < KS; ++a)
#pragma unroll
            for (int b = 0; b < KS; ++b) {
                const float v = tc[a * IT + b];
                const float* __restrict__ wt = wu + ((long long)c * KS * KS + a * KS + b) * COUT;
fma_row<COUT>(wt, v, acc2, accr);
            }
    }
    const int y = ty0 + py, xx = tx0 + px;
    if (y < H && xx < W) {
        float* ou = out + (long long)u * COUT * H * W + (long long)y * W + xx;
#pragma unroll
        for (int o = 0; o < COUT; ++o) ou[(long long)o * H * W] = fma_row_get<COUT>(acc2, accr, o);
    }
}

template <int KS, int COUT, bool ZERO>
int launch(const float* x, const float* w, const float* bias, int U, int Cin, int H, int W, bool shared, float* out,
           hipStream_t st) {
    const int tiles_x = ceil_div(W, RT), tiles = tiles_x * ceil_div(H, RT);
    const size_t shmem = (size_t)Cin * (RT + KS - 1) * (RT + KS - 1) * sizeof(float);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_reflect_kernel<KS, COUT, ZERO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    conv_reflect_kernel<KS, COUT, ZERO><<<U * tiles, 256, shmem, st>>>(x, w, bias, out, Cin, H, W, tiles_x, tiles,
                                                                       shared ? 0ll : (long long)Cin * KS * KS * COUT);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

int convk_forward(const kmsr_convk_desc* d, const float* x, const float* w, const float* bias, float* out, hipStream_t st) {
    const int k = d->k, Cout = d->Cout;
    const bool zero = d->pad_mode == KMSR_CONVK_ZERO, shared = d->shared_weights != 0;
#define KMSR_CR(K_, C_)                                                                                         \
    if (k == K_ && Cout == C_)                                                                                  \
        return zero ? launch<K_, C_, true>(x, w, bias, d->U, d->Cin, d->H, d->W, shared, out, st)               \
                    : launch<K_, C_, false>(x, w, bias, d->U, d->Cin, d->H, d->W, shared, out, st);
    KMSR_CR(1, 1) KMSR_CR(1, 32) KMSR_CR(1, 64) KMSR_CR(3, 1) KMSR_CR(3, 32) KMSR_CR(5, 1) KMSR_CR(5, 32) KMSR_CR(7, 1)
    KMSR_CR(7, 5) KMSR_CR(7, 32) KMSR_CR(7, 64)
#undef KMSR_CR
    return fail(KMSR_E_UNSUPPORTED, "convk: k=%d Cout=%d not instantiated (k in 1,3,5,7; Cout in 1,5,32,64)", k, Cout);
}

int check_desc(const kmsr_convk_desc* d, const char* who) {
    if (!d) return fail(KMSR_E_BADARG, "%s: null descriptor", who);
    if (d->U <= 0 || d->Cin <= 0 || d->Cin > 64 || d->Cout <= 0 || d->Cout > 64 || d->k < 1 || d->k > 7 || !(d->k & 1) ||
        d->H <= d->k / 2 || d->W <= d->k / 2)
        return fail(KMSR_E_BADARG, "%s: bad shape U=%d Cin=%d Cout=%d H=%d W=%d k=%d", who, d->U, d->Cin, d->Cout, d->H, d->W, d->k);
    if (d->pad_mode != KMSR_CONVK_REFLECT && d->pad_mode != KMSR_CONVK_ZERO) return fail(KMSR_E_BADARG, "%s: bad pad_mode", who);
    if ((long long)d->U * ceil_div(d->H, RT) * ceil_div(d->W, RT) > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "%s: grid too large", who);
    return KMSR_OK;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_conv_reflect_f32(const float* x, const float* w, int32_t U, int32_t Cin, int32_t Cout, int32_t H,
                                     int32_t W, int32_t k, float* out, void* stream) {
    if (!x || !w || !out) return fail(KMSR_E_BADARG, "conv_reflect: null pointer");
    kmsr_convk_desc d{U, Cin, Cout, H, W, k, KMSR_CONVK_REFLECT, 0};
    if (int rc = check_desc(&d, "conv_reflect")) return rc;
    return convk_forward(&d, x, w, nullptr, out, static_cast<hipStream_t>(stream));
}

extern "C" int kmsr_convk_fwd_f32(const kmsr_convk_desc* d, const float* x, const float* w, const float* bias, float* out,
                                  void* stream) {
    if (!x || !w || !out) return fail(KMSR_E_BADARG, "convk_fwd: null pointer");
    if (int rc = check_desc(d, "convk_fwd")) return rc;
    return convk_forward(d, x, w, bias, out, static_cast<hipStream_t>(stream));
}
