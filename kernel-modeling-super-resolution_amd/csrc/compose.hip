// KernelGAN kernel composition: collapse a bias-free deep-linear conv chain into one K x K kernel per (sample, band).
// Semantics and reference citations: include/kmsr.h.  One workgroup per (sample, band); the running kernel stack
// K_l [c_l][size_l^2] lives in LDS (two ping-pong buffers), stored ZERO-PADDED by k_{l+1} - 1 on every side so that the
// "full" convolution of the next layer needs no bounds.  A 32 -> 32 layer is a GEMM
//     D[o][n] = sum_kk W[o][kk] * im2col(K_l)[kk][n],   kk = (m, tap) -- exactly the memory order of the layer's weights,
// and runs on v_mfma_f32_32x32x2_f32: lane (o = l & 31, h = l >> 5) streams row o of W from L2 in 16-byte pieces at a 4 h byte
// skew (so .x / .z are its kk = 2 s + h operands, no selects), the im2col operand is one ds_read_b32 per MFMA at
// base(n) + off(kk) with off() tabulated in LDS once per layer; the condition scale multiplies the accumulators on the way
// out (the reference scales the weights first: same product, different rounding, inside the 3e-4 parity tolerance).
// Other layers (1 -> 32 first layer, 32 -> 1 last layer, any chain the reference never builds) take the VALU loop.
// 512 patches x 5 bands (config 5): 6.7 ms with the all-VALU kernel of round 1 -> see profiles/r02_config5_kernelgan_timing.log.
#include "kmsr_common.h"

#include <algorithm>
#include <type_traits>

namespace kmsr {
namespace {

constexpr int kMaxLayers = 8;
constexpr int kMaxCh = 64;
constexpr int kMaxK = 31;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
using i32x4 = __attribute__((ext_vector_type(4))) int;

struct ComposeParams {
    int B, C, n_layers;
    int co[kMaxLayers], ci[kMaxLayers], ks[kMaxLayers];
    int woff[kMaxLayers];   // float offset of layer l inside one band's weight block
    int soff[kMaxLayers];   // float offset of layer l inside one (sample, band) scale block
    int mfma[kMaxLayers];   // layer runs on the matrix pipe
    int wband, sband;       // floats per band (weights) / per (sample, band) (scales)
    int K, buf_floats, tab_ints;
};

__global__ void __launch_bounds__(256) compose_kernels_kernel(const float* __restrict__ weights,
                                                              const float* __restrict__ scales,
                                                              float* __restrict__ out, ComposeParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // buf0 | buf1 | offset table | reduction scratch
    float* cur = smem;
    float* nxt = smem + p.buf_floats;
    int* tab = reinterpret_cast<int*>(smem + 2 * p.buf_floats);       // [2][KT / 2]
    float* red = smem + 2 * p.buf_floats + p.tab_ints;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int band = blockIdx.x % p.C, b = blockIdx.x / p.C;
    const float* __restrict__ wb = weights + (long long)band * p.wband;
    const float* __restrict__ sb = scales ? scales + ((long long)b * p.C + band) * p.sband : nullptr;

    // layer 0: K_0[co][k0 x k0] = s_0[co] * W_0[co][0], stored with the padding layer 1 wants
    int size = p.ks[0];
    {
        const int pad = p.n_layers > 1 ? p.ks[1] - 1 : 0, PS = size + 2 * pad, PP = PS * PS;
        for (int e = tid; e < p.co[0] * PP; e += 256) cur[e] = 0.f;
        __syncthreads();
        for (int e = tid; e < p.co[0] * size * size; e += 256) {
            const int o = e / (size * size), rem = e - o * (size * size), y = rem / size, x = rem - y * size;
            cur[o * PP + (y + pad) * PS + x + pad] = wb[p.woff[0] + e] * (sb ? sb[p.soff[0] + o] : 1.f);
        }
    }
    int cmid = p.co[0];
    for (int l = 1; l < p.n_layers; ++l) {
        const int k = p.ks[l], kk2 = k * k, nsize = size + k - 1, np = nsize * nsize, nco = p.co[l];
        const int PS = size + 2 * (k - 1), PP = PS * PS;                               // this layer's (padded) input planes
        const int padn = l + 1 < p.n_layers ? p.ks[l + 1] - 1 : 0, PSn = nsize + 2 * padn, PPn = PSn * PSn;
        const float* __restrict__ wl = wb + p.woff[l];
        const int KT = cmid * kk2;
        for (int e = tid; e < nco * PPn; e += 256) nxt[e] = 0.f;
        if (p.mfma[l]) {
            for (int e = tid; e < KT; e += 256) {                                      // off(kk), split by the parity of kk
                const int m = e / kk2, tap = e - m * kk2, a = tap / k, bb = tap - a * k;
                tab[(e & 1) * (KT >> 1) + (e >> 1)] = m * PP + (k - 1 - a) * PS + (k - 1 - bb);
            }
        }
        __syncthreads();                                                               // input planes, zeroed output, table
        if (p.mfma[l]) {
            const int NT = (np + 31) >> 5;                                             // <= 8 tiles of 32 positions; wave w: tiles w, w + 4
            if (wv < NT) {
                const bool two = wv + 4 < NT;
                int base[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = min((wv + 4 * t) * 32 + l32, np - 1);
                    const int y = n / nsize, x = n - y * nsize;
                    base[t] = y * PS + x;
                }
                f32x16 acc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                const float* __restrict__ arow = wl + (long long)l32 * KT + half;      // row o of W, skewed by h
                const int* trow = tab + half * (KT >> 1);
                const int nblk = KT >> 5;                                              // blocks of 8 chunks = 32 kk (ci = 32: 8 k^2 chunks)
                // One block = 32 kk = 16 MFMA steps per tile.  All operand requests of a block go out before its first MFMA (the
                // offset table, then the 16 x T im2col samples; the next block's weight pieces are already in flight): written
                // step by step hipcc waits for each ds_read_b32 right before its MFMA and the LDS latency is paid per step.
                auto run = [&](auto TT) {
                    constexpr int T = decltype(TT)::value;
                    f32x4 av[2][8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) av[0][c] = *reinterpret_cast<const f32x4_u*>(arow + 4 * c);
                    for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {                               // phase ph computes av[ph] while av[ph ^ 1] loads
                            const int bk = blk + ph;
                            if (bk < nblk) {
                                if (bk + 1 < nblk) {
#pragma unroll
                                    for (int c = 0; c < 8; ++c)
                                        av[ph ^ 1][c] = *reinterpret_cast<const f32x4_u*>(arow + 32 * (bk + 1) + 4 * c);
                                }
                                i32x4 o4[4];
#pragma unroll
                                for (int c4 = 0; c4 < 4; ++c4) o4[c4] = *reinterpret_cast<const i32x4*>(trow + 16 * bk + 4 * c4);
                                float bs[T][16];
#pragma unroll
                                for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
#pragma unroll
                                        for (int t = 0; t < T; ++t) bs[t][4 * c4 + q] = cur[base[t] + o4[c4][q]];
#pragma unroll
                                for (int t = 0; t < T; ++t)
#pragma unroll
                                    for (int e = 0; e < 16; e += 4) asm volatile("" : "+v"(bs[t][e]), "+v"(bs[t][e + 1]), "+v"(bs[t][e + 2]), "+v"(bs[t][e + 3]));
#pragma unroll
                                for (int c4 = 0; c4 < 4; ++c4) {
                                    const float as[4] = {av[ph][2 * c4].x, av[ph][2 * c4].z, av[ph][2 * c4 + 1].x, av[ph][2 * c4 + 1].z};
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
#pragma unroll
                                        for (int t = 0; t < T; ++t)
                                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[q], bs[t][4 * c4 + q], acc[t], 0, 0, 0);
                                }
                            }
                        }
                    }
                };
                if (two) run(std::integral_constant<int, 2>{});
                else run(std::integral_constant<int, 1>{});
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = (wv + 4 * t) * 32 + l32;
                    if ((t == 0 || two) && n < np) {
                        const int y = n / nsize, x = n - y * nsize;
                        float* o = nxt + (y + padn) * PSn + x + padn;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int oc = 8 * g + 4 * half + r;
                                o[oc * PPn] = acc[t][4 * g + r] * (sb ? sb[p.soff[l] + oc] : 1.f);
                            }
                    }
                }
            }
        } else {
            for (int e = tid; e < nco * np; e += 256) {
                const int o = e / np, rem = e - o * np, y = rem / nsize, x = rem - y * nsize;
                const float sc = sb ? sb[p.soff[l] + o] : 1.f;
                float acc = 0.f;
                for (int m = 0; m < cmid; ++m) {
                    const float* __restrict__ A = wl + ((long long)o * cmid + m) * kk2;
                    const float* Bm = cur + m * PP + y * PS + x;
                    float part = 0.f;   // one conv_kernel(W[co,cm], K_cur[cm]) (networks.py:181-183), then acc += it (:184)
                    for (int a = 0; a < k; ++a)
                        for (int bb = 0; bb < k; ++bb) part = fmaf(A[a * k + bb] * sc, Bm[(k - 1 - a) * PS + (k - 1 - bb)], part);
                    acc += part;
                }
                nxt[o * PPn + (y + padn) * PSn + x + padn] = acc;
            }
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
    int woff = 0, soff = 0, K = 1, maxf = 0, maxtab = 0, size = 1;
    for (int l = 0; l < n_layers; ++l) {
        if (co[l] < 1 || co[l] > kMaxCh || ks[l] < 1 || !(ks[l] & 1)) return fail(KMSR_E_UNSUPPORTED, "compose: layer %d shape", l);
        if (l > 0 && ci[l] != co[l - 1]) return fail(KMSR_E_SHAPE, "compose: layer %d has %d inputs, previous has %d outputs", l, ci[l], co[l - 1]);
        p.co[l] = co[l]; p.ci[l] = ci[l]; p.ks[l] = ks[l];
        p.woff[l] = woff; p.soff[l] = soff;
        woff += co[l] * ci[l] * ks[l] * ks[l];
        soff += co[l];
        K += ks[l] - 1;
        size = K;                                                      // this layer's output size
        const int padn = l + 1 < n_layers ? ks[l + 1] - 1 : 0;         // stored with the next layer's padding
        maxf = std::max(maxf, co[l] * (size + 2 * padn) * (size + 2 * padn));
        // matrix-pipe layers: 32 -> 32, at most 8 tiles of 32 output positions, and not the last layer (the operand stream
        // reads 4 bytes past a weight row, which must still be inside the band's block)
        p.mfma[l] = l > 0 && l + 1 < n_layers && co[l] == 32 && ci[l] == 32 && size * size <= 256;
        if (p.mfma[l]) maxtab = std::max(maxtab, ci[l] * ks[l] * ks[l]);
    }
    if (K > kMaxK) return fail(KMSR_E_UNSUPPORTED, "compose: composed kernel %d > %d", K, kMaxK);
    p.wband = woff; p.sband = soff; p.K = K;
    p.buf_floats = round_up(maxf, 4);
    p.tab_ints = round_up(maxtab, 4);
    const size_t shmem = (2 * (size_t)p.buf_floats + p.tab_ints + 256) * sizeof(float);
    if (shmem > 150 * 1024) return fail(KMSR_E_UNSUPPORTED, "compose: %zu B of LDS needed", shmem);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(compose_kernels_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    compose_kernels_kernel<<<B * C, 256, shmem, st>>>(weights, scales, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
