// Small-channel k x k convolution with reflect (or zero) padding and per-unit (or shared) weights -- forward leg
// (semantics: include/kmsr.h; the data / weight gradient legs are in convk_bwd.hip).
// One workgroup = one 16 x 16 (or 16 x 32: two pixels per thread, the 32 -> 32 layers) output tile of one unit; the reflect-padded
// input tile sits in LDS (all input channels, or chunks of 8 for the wide tile), each thread owns its pixel(s) and all COUT output
// channels as v_pk_fma_f32 pairs; weights are wave-uniform ([ci][tap][co] contiguous per unit) so they arrive through the scalar
// cache and feed the FMAs as SGPR operands.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

constexpr int RT = 16;   // output tile edge
constexpr int CK = 64;   // input channels per LDS chunk, one pixel per thread (= all of them: see the note in the kernel)
constexpr int CK2 = 8;   // ... two pixels per thread

__device__ __forceinline__ int reflect_idx(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}

// NPX = output pixels per thread (columns px, px + 16, ...): every scalar-loaded weight pair feeds NPX packed FMAs, which divides
// the scalar-cache traffic per FMA -- the measured bound of this kernel -- by NPX; the tile is then RT rows x 16 NPX columns and the
// input channels go through LDS in chunks of CKN so that the wider tile still fits several workgroups per CU.
template <int KS, int COUT, bool ZERO, int NPX>
__global__ void __launch_bounds__(256) conv_reflect_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                           int H, int W, int tiles_x, int tiles, long long wstride) {
    constexpr int P = KS / 2, IT = RT + KS - 1, ITW = RT * NPX + KS - 1, CKN = NPX > 1 ? CK2 : CK;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [min(Cin, CKN)][IT][ITW]: one chunk of input channels
    const int tid = threadIdx.x;
    const int u = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int ty0 = (t / tiles_x) * RT, tx0 = (t % tiles_x) * (RT * NPX);
    const float* __restrict__ xu = x + (long long)u * Cin * H * W;
    const int py = tid / RT, px = tid % RT;
    kmsr_f32x2 acc2[NPX][(COUT + 1) / 2];
    float accr[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        accr[j] = (COUT & 1) && bias ? bias[COUT - 1] : 0.f;
#pragma unroll
        for (int q = 0; q < COUT / 2; ++q) acc2[j][q] = bias ? kmsr_f32x2{bias[2 * q], bias[2 * q + 1]} : kmsr_f32x2{0.f, 0.f};
    }
    const float* __restrict__ wu = w + (long long)u * wstride;                 // wave-uniform
    // Measured (5x5, 32 -> 32, 40 units at 256^2, profiles/r02_n1_kernelgan_trainer_timing.log), one pixel per thread: all channels
    // in LDS (51 KB, 3 workgroups per CU) 2.69 ms; chunks of 8 (13 KB, 7 workgroups per CU) 3.08 ms; tap t + 1's scalar loads and LDS
    // read requested before tap t's FMAs (checked in the .s) 3.51 ms; scalar vs packed FMAs: no difference.  The waves sit at 10 %
    // issue: a unit's weights (102 KB) stream through the 16 KB scalar cache once per wave, and its miss throughput -- not FMA issue,
    // not load latency -- is the bound.
    for (int c0 = 0; c0 < Cin; c0 += CKN) {
        const int nc = min(CKN, Cin - c0);
        if (c0) __syncthreads();
        // 8 loads in flight per thread before any is written to LDS (a rolled load->store loop pays a round trip per element)
        const int n_el = nc * IT * ITW;
        for (int e0 = 0; e0 < n_el; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + tid + 256 * q, n_el - 1);
                const int c = e / (IT * ITW), rem = e - c * (IT * ITW), r = rem / ITW, cc = rem - r * ITW;
                if constexpr (ZERO) {
                    const int gy = ty0 + r - P, gx = tx0 + cc - P;
                    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    v[q] = in ? xu[((long long)(c0 + c) * H + gy) * W + gx] : 0.f;
                } else {
                    const int gy = reflect_idx(min(ty0 + r - P, H - 1 + P), H), gx = reflect_idx(min(tx0 + cc - P, W - 1 + P), W);
                    v[q] = xu[((long long)(c0 + c) * H + gy) * W + gx];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 256 * q;
                if (e < n_el) tile[e] = v[q];
            }
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            const float* tc = tile + c * IT * ITW + py * ITW + px;
#pragma unroll
            for (int a = 0; a < KS; ++a)
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    const float* __restrict__ wt = wu + ((long long)(c0 + c) * KS * KS + a * KS + b) * COUT;
#pragma unroll
                    for (int j = 0; j < NPX; ++j) fma_row<COUT>(wt, tc[a * ITW + b + RT * j], acc2[j], accr[j]);
                }
        }
    }
    const int y = ty0 + py;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int xx = tx0 + px + RT * j;
        if (y < H && xx < W) {
            float* ou = out + (long long)u * COUT * H * W + (long long)y * W + xx;
#pragma unroll
            for (int o = 0; o < COUT; ++o) ou[(long long)o * H * W] = fma_row_get<COUT>(acc2[j], accr[j], o);
        }
    }
}

template <int KS, int COUT, bool ZERO, int NPX>
int launch_npx(const float* x, const float* w, const float* bias, int U, int Cin, int H, int W, bool shared, float* out,
               hipStream_t st) {
    const int tiles_x = ceil_div(W, RT * NPX), tiles = tiles_x * ceil_div(H, RT);
    const size_t shmem = (size_t)std::min(Cin, NPX > 1 ? CK2 : CK) * (RT + KS - 1) * (RT * NPX + KS - 1) * sizeof(float);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_reflect_kernel<KS, COUT, ZERO, NPX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    conv_reflect_kernel<KS, COUT, ZERO, NPX><<<U * tiles, 256, shmem, st>>>(x, w, bias, out, Cin, H, W, tiles_x, tiles,
                                                                            shared ? 0ll : (long long)Cin * KS * KS * COUT);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

template <int KS, int COUT, bool ZERO>
int launch(const float* x, const float* w, const float* bias, int U, int Cin, int H, int W, bool shared, float* out,
           hipStream_t st) {
    // the 32 -> 32 layers (per-unit weights far beyond the scalar cache) take two pixels per thread
    if constexpr (COUT == 32 && KS > 1) {
        if (Cin >= CK2 && W > RT) return launch_npx<KS, COUT, ZERO, 2>(x, w, bias, U, Cin, H, W, shared, out, st);
    }
    return launch_npx<KS, COUT, ZERO, 1>(x, w, bias, U, Cin, H, W, shared, out, st);
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
