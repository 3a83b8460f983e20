// Gradient legs of the small-channel k x k convolution family (semantics: include/kmsr.h, kmsr_convk_*).
//   dgrad: dx = adjoint of [pad -> conv] applied to dy.  Zero padding: the same convolution with flipped, transposed taps.
//          Reflect padding: additionally every padded position folds back onto the interior pixel it mirrors, so a pixel
//          within k/2 of a border sums up to 3 x 3 mirrored read positions per tap (workgroup-uniform fast path elsewhere).
//   wgrad: dw[ci,tap,co] = sum_pixels dy[co,p] * xpad[ci,p+tap].  Per (unit, strip) workgroup: x tile (padded) and the dy
//          tile ([pixel][co], co fastest) in LDS; a thread owns NU (ci,tap) pairs x CB consecutive co and walks the 256
//          pixels of a tile with 1 b32 + CB/4 b128 LDS reads per CB FMAs.  Strip partials are summed in a fixed order.
//          With scratch memory the 32-channel reflect layers run as a zero-padded pass on the padded domain + convk_fold_kernel.
// Thread-per-pixel / thread-per-output VALU code (weights as SGPR operands) for the 1..5- and 64-channel layers of the KernelGAN
// generator chains and the patch critic; the 32 -> 32 layers' weight gradient is a per-unit GEMM on the matrix pipe
// (convk_wgrad_mfma_kernel).  Numbers: profiles/r02_n1_kernelgan_trainer_timing.log.
#include <algorithm>
#include <type_traits>

#include "kmsr_common.h"

namespace kmsr {
namespace {

constexpr int RT = 16;
constexpr int DCK = 8;   // gradient channels per LDS chunk of the data-gradient kernel

__device__ __forceinline__ int reflect_idx(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad.  dy [U,Cdy,H,W]; wT [U or 1][Cdy][k*k][CDX] (forward weight w[cdx][tap][cdy] transposed, taps NOT flipped);
// dx [U,CDX,H,W].  LDS tile dy[Cdy][IT][IT] with origin (ty0-P, tx0-P), zero outside the image.
// dx[y,x] = sum_{vy in S(y), vx in S(x)} sum_{c,a,b} wT[c][a*k+b][:] * dy0[c][vy+P-a][vx+P-b]
// S(y) = {y} (zero) or {y, -y if 1<=y<=P, 2(H-1)-y if H-1-P<=y<=H-2} (reflect).
// NPX output pixels per thread (columns px, px + 16, ...) as in conv_reflect.hip: the scalar-cache traffic per FMA -- the bound of
// these kernels -- goes down by NPX.
template <int KS, int CDX, bool REFLECT, int NPX>
__global__ void __launch_bounds__(256) convk_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wT,
                                                          float* __restrict__ dx, int Cdy, int H, int W, int tiles_x,
                                                          int tiles, long long wstride, int Hin, int Win, int off) {
    // output domain H x W; dy is Hin x Win and sits at (off, off) of it (off = 0, Hin = H: the plain call; off = P, Hin = H - 2 P:
    // the padded-domain pass of the reflect data gradient, see convk_fold_kernel)
    constexpr int P = KS / 2, IT = RT + KS - 1, ITW = RT * NPX + KS - 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [min(Cdy, DCK)][IT][ITW]: one chunk of gradient channels
    const int tid = threadIdx.x;
    const int u = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int ty0 = (t / tiles_x) * RT, tx0 = (t % tiles_x) * (RT * NPX);
    const float* __restrict__ du = dy + (long long)u * Cdy * Hin * Win;
    const int py = tid / RT, px = tid % RT;
    const int y = ty0 + py;
    kmsr_f32x2 acc2[NPX][(CDX + 1) / 2];
    float accr[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        accr[j] = 0.f;
#pragma unroll
        for (int q = 0; q < (CDX + 1) / 2; ++q) acc2[j][q] = kmsr_f32x2{0.f, 0.f};
    }
    const float* __restrict__ wu = wT + (long long)u * wstride;
    // rows / cols of the LDS tile that tap (a,b) reads for source v:  (v - t0) + 2P - a
    bool border = false;
    if constexpr (REFLECT) border = ty0 <= P || ty0 + RT >= H - 1 - P || tx0 <= P || tx0 + RT * NPX >= W - 1 - P;   // block-uniform
    int ry[3], rx[NPX][3];                         // border tiles: tile row / col of each source before the tap shift; < -64 = absent
    ry[0] = py + 2 * P;
    ry[1] = (y >= 1 && y <= P) ? -y - ty0 + 2 * P : -1000;
    ry[2] = (y >= H - 1 - P && y <= H - 2) ? 2 * (H - 1) - y - ty0 + 2 * P : -1000;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int xx = tx0 + px + RT * j;
        rx[j][0] = px + RT * j + 2 * P;
        rx[j][1] = (xx >= 1 && xx <= P) ? -xx - tx0 + 2 * P : -1000;
        rx[j][2] = (xx >= W - 1 - P && xx <= W - 2) ? 2 * (W - 1) - xx - tx0 + 2 * P : -1000;
    }
    for (int c0 = 0; c0 < Cdy; c0 += DCK) {
        const int nc = min(DCK, Cdy - c0);
        if (c0) __syncthreads();
        const int n_el = nc * IT * ITW;
        for (int e0 = 0; e0 < n_el; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + tid + 256 * q, n_el - 1);
                const int c = e / (IT * ITW), rem = e - c * (IT * ITW), r = rem / ITW, cc = rem - r * ITW;
                const int gy = ty0 + r - P - off, gx = tx0 + cc - P - off;
                const bool in = gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
                v[q] = in ? du[((long long)(c0 + c) * Hin + gy) * Win + gx] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 256 * q;
                if (e < n_el) tile[e] = v[q];
            }
        }
        __syncthreads();
        if (!border) {
            for (int c = 0; c < nc; ++c) {
                const float* tc = tile + c * IT * ITW + (py + 2 * P) * ITW + (px + 2 * P);
#pragma unroll
                for (int a = 0; a < KS; ++a)
#pragma unroll
                    for (int b = 0; b < KS; ++b) {
                        const float* __restrict__ wt = wu + ((long long)(c0 + c) * KS * KS + a * KS + b) * CDX;
#pragma unroll
                        for (int j = 0; j < NPX; ++j) fma_row<CDX>(wt, tc[-a * ITW - b + RT * j], acc2[j], accr[j]);
                    }
            }
        } else {
            for (int c = 0; c < nc; ++c) {
                const float* tc = tile + c * IT * ITW;
                for (int a = 0; a < KS; ++a)
                    for (int b = 0; b < KS; ++b) {
                        float v[NPX];
#pragma unroll
                        for (int j = 0; j < NPX; ++j) {
                            v[j] = 0.f;                                // unrolled 3 x 3 with sentinels: a compacted source list with
#pragma unroll                                                         // run-time trip counts measured slower (4.45 vs 3.58 ms)
                            for (int i = 0; i < 3; ++i)
#pragma unroll
                                for (int jj = 0; jj < 3; ++jj) {
                                    const int r = ry[i] - a, cc = rx[j][jj] - b;
                                    if (r >= 0 && r < IT && cc >= 0 && cc < ITW) v[j] += tc[r * ITW + cc];
                                }
                        }
                        const float* __restrict__ wt = wu + ((long long)(c0 + c) * KS * KS + a * KS + b) * CDX;
#pragma unroll
                        for (int j = 0; j < NPX; ++j) fma_row<CDX>(wt, v[j], acc2[j], accr[j]);
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int xx = tx0 + px + RT * j;
        if (y < H && xx < W) {
            float* ou = dx + (long long)u * CDX * H * W + (long long)y * W + xx;
#pragma unroll
            for (int o = 0; o < CDX; ++o) ou[(long long)o * H * W] = fma_row_get<CDX>(acc2[j], accr[j], o);
        }
    }
}

template <int KS, int CDX, bool REFLECT, int NPX>
int launch_dgrad_npx(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, hipStream_t st, int Ho, int Wo, int off) {
    const int tiles_x = ceil_div(Wo, RT * NPX), tiles = tiles_x * ceil_div(Ho, RT);
    const size_t shmem = (size_t)std::min(d->Cout, DCK) * (RT + KS - 1) * (RT * NPX + KS - 1) * sizeof(float);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(convk_dgrad_kernel<KS, CDX, REFLECT, NPX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    convk_dgrad_kernel<KS, CDX, REFLECT, NPX><<<d->U * tiles, 256, shmem, st>>>(
        dy, wT, dx, d->Cout, Ho, Wo, tiles_x, tiles, d->shared_weights ? 0ll : (long long)d->Cout * KS * KS * CDX, d->H, d->W, off);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

// Reflect data gradient in two passes (workspace given):  F = zero-padded data gradient on the PADDED domain (H + 2P) x (W + 2P)
// (every tile takes the border-free fast path, two pixels per thread), then dx[z] = sum of F over the padded positions that the
// reflection maps onto z (<= 3 per axis).  One pass with mirrored sources per tap: 3.6 ms at 5x5, 40 units; see the trainer log.
__global__ void __launch_bounds__(256) convk_fold_kernel(const float* __restrict__ F, float* __restrict__ dx, long long planes,
                                                         int H, int W, int P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long pl = i / ((long long)H * W);
    const int Wp = W + 2 * P;
    const float* f = F + pl * (long long)(H + 2 * P) * Wp;
    int ys[3] = {y, -1000, -1000}, xs[3] = {x, -1000, -1000};
    if (y >= 1 && y <= P) ys[1] = -y;
    if (y >= H - 1 - P && y <= H - 2) ys[2] = 2 * (H - 1) - y;
    if (x >= 1 && x <= P) xs[1] = -x;
    if (x >= W - 1 - P && x <= W - 2) xs[2] = 2 * (W - 1) - x;
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (ys[q] > -500 && xs[r] > -500) a += f[(long long)(ys[q] + P) * Wp + xs[r] + P];
    dx[i] = a;
}

template <int KS, int CDX, bool REFLECT>
int launch_dgrad(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, float* ws, hipStream_t st) {
    constexpr int P = KS / 2;
    if constexpr (CDX == 32 && KS > 1) {             // the 32 -> 32 layers: per-unit weights far beyond the scalar cache
        if constexpr (!REFLECT) {
            if (d->Cout >= DCK && d->W > RT) return launch_dgrad_npx<KS, CDX, false, 2>(d, dy, wT, dx, st, d->H, d->W, 0);
        } else if (ws && d->Cout >= DCK) {
            if (int rc = launch_dgrad_npx<KS, CDX, false, 2>(d, dy, wT, ws, st, d->H + 2 * P, d->W + 2 * P, P)) return rc;
            const long long n = (long long)d->U * CDX * d->H * d->W;
            convk_fold_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ws, dx, (long long)d->U * CDX, d->H, d->W, P);
            KMSR_HIP_CHECK(hipGetLastError());
            return KMSR_OK;
        }
    }
    // one pass, reflect padding: 2 pixels per thread measured SLOWER (5.67 vs 3.58 ms at 5x5, 40 units): a third of the wider tiles
    // take the mirrored-source path, which pays per pixel
    return launch_dgrad_npx<KS, CDX, REFLECT, 1>(d, dy, wT, dx, st, d->H, d->W, 0);
}

size_t dgrad_ws_bytes(const kmsr_convk_desc* d) {
    const int P = d->k / 2;
    if (d->pad_mode != KMSR_CONVK_REFLECT || d->Cin != 32 || d->k < 3 || d->Cout < DCK) return 0;
    return (size_t)d->U * d->Cin * (d->H + 2 * P) * (d->W + 2 * P) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad
template <int CB, int NU, bool REFLECT>
__global__ void __launch_bounds__(256) convk_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ part, int Cin, int Cout, int H, int W, int k,
                                                          int tiles_x, int tiles, int NS) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = k / 2, IT = RT + k - 1, KK = k * k;
    float* xt = lds;                                      // [Cin][IT][IT]
    float* dt = lds + ((Cin * IT * IT + 3) & ~3);     // [256 px][Cout]
    const int tid = threadIdx.x;
    const int u = blockIdx.x / NS, s = blockIdx.x % NS;
    const int NCB = Cout / CB, n_pairs = Cin * KK, n_units = n_pairs * NCB;
    int xbase[NU], dbase[NU];
    bool valid[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int idx = tid + 256 * i;
        valid[i] = idx < n_units;
        const int pair = min(idx, n_units - 1) / NCB, cb = min(idx, n_units - 1) % NCB;
        const int ci = pair / KK, tap = pair % KK;
        xbase[i] = ci * IT * IT + (tap / k) * IT + tap % k;
        dbase[i] = cb * CB;
    }
    float acc[NU][CB];
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[i][c] = 0.f;
    const float* __restrict__ xu = x + (long long)u * Cin * H * W;
    const float* __restrict__ du = dy + (long long)u * Cout * H * W;
    const int n_x = Cin * IT * IT, n_d = Cout * RT * RT;
    for (int t = s; t < tiles; t += NS) {
        const int ty0 = (t / tiles_x) * RT, tx0 = (t % tiles_x) * RT;
        __syncthreads();                                  // previous tile fully consumed
        for (int e0 = 0; e0 < n_x; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + tid + 256 * q, n_x - 1);
                const int c = e / (IT * IT), rem = e - c * (IT * IT), r = rem / IT, cc = rem - r * IT;
                if constexpr (REFLECT) {
                    const int gy = reflect_idx(min(ty0 + r - P, H - 1 + P), H), gx = reflect_idx(min(tx0 + cc - P, W - 1 + P), W);
                    v[q] = xu[((long long)c * H + gy) * W + gx];
                } else {
                    const int gy = ty0 + r - P, gx = tx0 + cc - P;
                    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    v[q] = in ? xu[((long long)c * H + gy) * W + gx] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 256 * q;
                if (e < n_x) xt[e] = v[q];
            }
        }
        for (int e0 = 0; e0 < n_d; e0 += 256 * 8) {       // e = co*256 + p (global-coalesced along x), LDS at p*Cout + co
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + tid + 256 * q, n_d - 1);
                const int co = e >> 8, p = e & 255, gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
                v[q] = (gy < H && gx < W) ? du[((long long)co * H + gy) * W + gx] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 256 * q;
                if (e < n_d) dt[(e & 255) * Cout + (e >> 8)] = v[q];
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int p = 0; p < RT * RT; ++p) {
            const int po = (p >> 4) * IT + (p & 15);
            const float* dp = dt + p * Cout;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const float xv = xt[xbase[i] + po];
                if constexpr (CB % 4 == 0) {
#pragma unroll
                    for (int c = 0; c < CB; c += 4) {
                        const float4 g = *reinterpret_cast<const float4*>(dp + dbase[i] + c);
                        acc[i][c] = fmaf(xv, g.x, acc[i][c]);
                        acc[i][c + 1] = fmaf(xv, g.y, acc[i][c + 1]);
                        acc[i][c + 2] = fmaf(xv, g.z, acc[i][c + 2]);
                        acc[i][c + 3] = fmaf(xv, g.w, acc[i][c + 3]);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CB; ++c) acc[i][c] = fmaf(xv, dp[dbase[i] + c], acc[i][c]);
                }
            }
        }
    }
    float* pu = part + ((long long)u * NS + s) * n_pairs * Cout;
#pragma unroll
    for (int i = 0; i < NU; ++i)
        if (valid[i]) {
            const int idx = tid + 256 * i;
            float* po = pu + (long long)(idx / NCB) * Cout + dbase[i];
#pragma unroll
            for (int c = 0; c < CB; ++c) po[c] = acc[i][c];
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad of the 32 -> 32 layers on the matrix pipe: per unit  dw[co][(ci,tap)] = sum_px dy[co][px] * xpad[ci][px + tap]  is a GEMM with
// M = 32 (co), N = 32 k^2 columns (ci, tap), K = pixels.  v_mfma_f32_32x32x2_f32: A = dy[co = l & 31][px = 2 s + (l >> 5)] from an LDS
// tile [co][256 + 1] (two lanes per bank: the minimum for 64 lanes), B = x tile read at colbase(column) + pixel offset (im2col by
// addressing, one ds_read_b32 per MFMA).  8 waves; wave w owns the column tiles w, w + 8, ... (<= 4 of the k^2 <= 25) and keeps their
// 32 x 32 sums in registers over all the 16 x 16 pixel tiles of its strip; partial layout and the fixed-order reduction are those of
// the VALU kernel above.  5x5, 40 units at 256^2: 6.6 ms (VALU, LDS-bound: 1 + 4 reads per 16 FMAs) -> see the trainer timing log.
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4w = __attribute__((ext_vector_type(4))) float;
constexpr int WM_DP = RT * RT + 1;         // dy tile pitch
constexpr int WM_MAXQ = 4;                 // column tiles per wave

template <bool REFLECT>
__global__ void __launch_bounds__(512) convk_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ part, int H, int W, int k, int tiles_x,
                                                               int tiles, int NS) {
    constexpr int C = 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = k / 2, IT = RT + k - 1, KK = k * k;
    float* xt = lds;                                      // [32][IT][IT]
    float* dt = lds + ((C * IT * IT + 3) & ~3);           // [32][WM_DP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int u = blockIdx.x / NS, s = blockIdx.x % NS;
    int colbase[WM_MAXQ];
#pragma unroll
    for (int q = 0; q < WM_MAXQ; ++q) {
        const int J = min(wv + 8 * q, KK - 1) * 32 + l32;            // column (ci, tap) of this lane in column tile q
        const int ci = J / KK, tap = J - ci * KK;
        colbase[q] = ci * IT * IT + (tap / k) * IT + tap % k + half;
    }
    const int nq = wv < KK ? (KK - 1 - wv) / 8 + 1 : 0;              // column tiles this wave owns (wave-uniform)
    f32x16 acc[WM_MAXQ];
#pragma unroll
    for (int q = 0; q < WM_MAXQ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float* __restrict__ xu = x + (long long)u * C * H * W;
    const float* __restrict__ du = dy + (long long)u * C * H * W;
    const int n_x = C * IT * IT, n_d = C * RT * RT;
    const float* arow = dt + l32 * WM_DP + half;
    for (int t = s; t < tiles; t += NS) {
        const int ty0 = (t / tiles_x) * RT, tx0 = (t % tiles_x) * RT;
        __syncthreads();                                  // previous tile fully consumed
        for (int e0 = 0; e0 < n_x; e0 += 512 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + tid + 512 * q, n_x - 1);
                const int c = e / (IT * IT), rem = e - c * (IT * IT), r = rem / IT, cc = rem - r * IT;
                if constexpr (REFLECT) {
                    const int gy = reflect_idx(min(ty0 + r - P, H - 1 + P), H), gx = reflect_idx(min(tx0 + cc - P, W - 1 + P), W);
                    v[q] = xu[((long long)c * H + gy) * W + gx];
                } else {
                    const int gy = ty0 + r - P, gx = tx0 + cc - P;
                    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    v[q] = in ? xu[((long long)c * H + gy) * W + gx] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 512 * q;
                if (e < n_x) xt[e] = v[q];
            }
        }
        for (int e0 = 0; e0 < n_d; e0 += 512 * 8) {       // e = co*256 + p: global-coalesced along x, LDS at co*WM_DP + p; 0 outside the image
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 512 * q;
                const int co = e >> 8, p = e & 255, gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
                v[q] = (gy < H && gx < W) ? du[((long long)co * H + gy) * W + gx] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + 512 * q;
                dt[(e >> 8) * WM_DP + (e & 255)] = v[q];
            }
        }
        __syncthreads();
        // all operand requests of a pixel row (8 steps: 8 A + 8 NQ B values) go out before its first MFMA -- written step by step hipcc
        // waits for each ds_read_b32 right before its MFMA (seen in the .s) and the LDS latency is paid per step
        auto run = [&](auto NQT) {
            constexpr int NQ = decltype(NQT)::value;
            for (int row = 0; row < RT; ++row) {
                const float* ar = arow + row * RT;
                const int xo = row * IT;
                float av[RT / 2], bv[NQ][RT / 2];
#pragma unroll
                for (int t8 = 0; t8 < RT / 2; ++t8) {
                    av[t8] = ar[2 * t8];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) bv[q][t8] = xt[colbase[q] + xo + 2 * t8];
                }
                asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]));
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    asm volatile("" : "+v"(bv[q][0]), "+v"(bv[q][1]), "+v"(bv[q][2]), "+v"(bv[q][3]), "+v"(bv[q][4]), "+v"(bv[q][5]), "+v"(bv[q][6]),
                                 "+v"(bv[q][7]));
#pragma unroll
                for (int t8 = 0; t8 < RT / 2; ++t8)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t8], bv[q][t8], acc[q], 0, 0, 0);
            }
        };
        if (nq == 4) run(std::integral_constant<int, 4>{});
        else if (nq == 3) run(std::integral_constant<int, 3>{});
        else if (nq == 2) run(std::integral_constant<int, 2>{});
        else if (nq == 1) run(std::integral_constant<int, 1>{});
    }
    float* pu = part + ((long long)u * NS + s) * (C * KK) * C;
#pragma unroll
    for (int q = 0; q < WM_MAXQ; ++q)
        if (q < nq) {
            const int J = (wv + 8 * q) * 32 + l32;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4w*>(pu + (long long)J * C + 8 * g + 4 * half) =
                    f32x4w{acc[q][4 * g + 0], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]};
        }
}

// dw[g][o] = sum_{r < R} part[g*R + r][o]  (fixed order -> run-to-run identical)
__global__ void __launch_bounds__(256) convk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int R,
                                                           int n_out, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long g = i / n_out;
    const int o = (int)(i - g * n_out);
    const float* p = part + g * R * n_out + o;
    float s0 = 0.f, s1 = 0.f;
    int r = 0;
    for (; r + 1 < R; r += 2) {
        s0 += p[(long long)r * n_out];
        s1 += p[(long long)(r + 1) * n_out];
    }
    if (r < R) s0 += p[(long long)r * n_out];
    dw[i] = s0 + s1;
}

struct WgradPlan {
    int CB, NU, NS;
    bool mfma;
    size_t shmem, ws_bytes;
};

int cu_count_bwd() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

int plan_wgrad(const kmsr_convk_desc* d, WgradPlan* pl) {
    const int IT = RT + d->k - 1, KK = d->k * d->k;
    pl->CB = d->Cout % 16 == 0 ? 16 : d->Cout % 4 == 0 ? 4 : 1;
    const int units = d->Cin * KK * (d->Cout / pl->CB);
    const int nu = ceil_div(units, 256);
    pl->NU = nu <= 1 ? 1 : nu <= 2 ? 2 : nu <= 4 ? 4 : 8;
    if (nu > 8) return fail(KMSR_E_UNSUPPORTED, "convk_wgrad: Cin*k*k*Cout = %d outputs exceed one workgroup's registers", d->Cin * KK * d->Cout);
    const int tiles = ceil_div(d->W, RT) * ceil_div(d->H, RT);
    pl->mfma = d->Cin == 32 && d->Cout == 32 && KK <= 8 * WM_MAXQ;
    if (pl->mfma) {                                      // one 8-wave workgroup per CU: about four rounds of workgroups
        pl->NS = std::max(1, std::min(tiles, 4 * cu_count_bwd() / d->U));
        pl->shmem = ((size_t)round_up(32 * IT * IT, 4) + (size_t)32 * WM_DP) * sizeof(float);
        pl->ws_bytes = (size_t)d->U * pl->NS * d->Cin * KK * d->Cout * sizeof(float);
        return KMSR_OK;
    }
    pl->NS = std::max(1, std::min(tiles, 1024 / d->U));
    pl->shmem = ((size_t)round_up(d->Cin * IT * IT, 4) + (size_t)RT * RT * d->Cout) * sizeof(float);
    if (pl->shmem > 160 * 1024) return fail(KMSR_E_UNSUPPORTED, "convk_wgrad: tiles need %zu bytes of LDS", pl->shmem);
    pl->ws_bytes = (size_t)d->U * pl->NS * d->Cin * KK * d->Cout * sizeof(float);
    return KMSR_OK;
}

template <int CB, int NU>
int launch_wgrad(const kmsr_convk_desc* d, const WgradPlan& pl, const float* x, const float* dy, float* part, hipStream_t st) {
    const int tiles_x = ceil_div(d->W, RT), tiles = tiles_x * ceil_div(d->H, RT);
    const bool refl = d->pad_mode == KMSR_CONVK_REFLECT;
    const void* fn = refl ? reinterpret_cast<const void*>(convk_wgrad_kernel<CB, NU, true>)
                          : reinterpret_cast<const void*>(convk_wgrad_kernel<CB, NU, false>);
    if (pl.shmem > 64 * 1024) KMSR_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.shmem));
    if (refl)
        convk_wgrad_kernel<CB, NU, true><<<d->U * pl.NS, 256, pl.shmem, st>>>(x, dy, part, d->Cin, d->Cout, d->H, d->W, d->k,
                                                                              tiles_x, tiles, pl.NS);
    else
        convk_wgrad_kernel<CB, NU, false><<<d->U * pl.NS, 256, pl.shmem, st>>>(x, dy, part, d->Cin, d->Cout, d->H, d->W, d->k,
                                                                               tiles_x, tiles, pl.NS);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
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

extern "C" size_t kmsr_convk_dgrad_workspace_bytes(const kmsr_convk_desc* d) {
    if (check_desc(d, "convk_dgrad_workspace_bytes")) return 0;
    return dgrad_ws_bytes(d);
}

extern "C" int kmsr_convk_dgrad_ws_f32(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!dy || !wT || !dx) return fail(KMSR_E_BADARG, "convk_dgrad: null pointer");
    if (int rc = check_desc(d, "convk_dgrad")) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool refl = d->pad_mode == KMSR_CONVK_REFLECT;
    float* ws = nullptr;
    if (workspace && dgrad_ws_bytes(d)) {
        if (workspace_bytes < dgrad_ws_bytes(d))
            return fail(KMSR_E_NOMEM, "convk_dgrad: workspace %zu < %zu bytes", workspace_bytes, dgrad_ws_bytes(d));
        ws = static_cast<float*>(workspace);
    }
#define KMSR_DG(K_, C_)                                                                              \
    if (d->k == K_ && d->Cin == C_)                                                                  \
        return refl ? launch_dgrad<K_, C_, true>(d, dy, wT, dx, ws, st) : launch_dgrad<K_, C_, false>(d, dy, wT, dx, ws, st);
    KMSR_DG(1, 1) KMSR_DG(1, 32) KMSR_DG(1, 64) KMSR_DG(3, 1) KMSR_DG(3, 32) KMSR_DG(5, 1) KMSR_DG(5, 32) KMSR_DG(7, 1)
    KMSR_DG(7, 5) KMSR_DG(7, 32)
#undef KMSR_DG
    return fail(KMSR_E_UNSUPPORTED, "convk_dgrad: k=%d Cin=%d not instantiated (k in 1,3,5,7; Cin in 1,5,32,64)", d->k, d->Cin);
}

extern "C" int kmsr_convk_dgrad_f32(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, void* stream) {
    return kmsr_convk_dgrad_ws_f32(d, dy, wT, dx, nullptr, 0, stream);
}

extern "C" size_t kmsr_convk_wgrad_workspace_bytes(const kmsr_convk_desc* d) {
    WgradPlan pl;
    if (check_desc(d, "convk_wgrad_workspace_bytes") || plan_wgrad(d, &pl)) return 0;
    return pl.ws_bytes;
}

extern "C" int kmsr_convk_wgrad_f32(const kmsr_convk_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return fail(KMSR_E_BADARG, "convk_wgrad: null pointer");
    if (int rc = check_desc(d, "convk_wgrad")) return rc;
    WgradPlan pl;
    if (int rc = plan_wgrad(d, &pl)) return rc;
    if (workspace_bytes < pl.ws_bytes) return fail(KMSR_E_NOMEM, "convk_wgrad: workspace %zu < %zu bytes", workspace_bytes, pl.ws_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    int rc = KMSR_E_UNSUPPORTED;
    if (pl.mfma) {
        const int tiles_x = ceil_div(d->W, RT), tiles = tiles_x * ceil_div(d->H, RT);
        const bool refl = d->pad_mode == KMSR_CONVK_REFLECT;
        const void* fn = refl ? reinterpret_cast<const void*>(convk_wgrad_mfma_kernel<true>)
                              : reinterpret_cast<const void*>(convk_wgrad_mfma_kernel<false>);
        if (pl.shmem > 64 * 1024) KMSR_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.shmem));
        if (refl) convk_wgrad_mfma_kernel<true><<<d->U * pl.NS, 512, pl.shmem, st>>>(x, dy, part, d->H, d->W, d->k, tiles_x, tiles, pl.NS);
        else convk_wgrad_mfma_kernel<false><<<d->U * pl.NS, 512, pl.shmem, st>>>(x, dy, part, d->H, d->W, d->k, tiles_x, tiles, pl.NS);
        KMSR_HIP_CHECK(hipGetLastError());
        rc = KMSR_OK;
    }
#define KMSR_WG(CB_, NU_) \
    if (!pl.mfma && pl.CB == CB_ && pl.NU == NU_) rc = launch_wgrad<CB_, NU_>(d, pl, x, dy, part, st);
    KMSR_WG(1, 1) KMSR_WG(1, 2) KMSR_WG(1, 4) KMSR_WG(1, 8) KMSR_WG(4, 1) KMSR_WG(4, 2) KMSR_WG(4, 4) KMSR_WG(4, 8)
    KMSR_WG(16, 1) KMSR_WG(16, 2) KMSR_WG(16, 4) KMSR_WG(16, 8)
#undef KMSR_WG
    if (rc) return rc;
    const int n_out = d->Cin * d->k * d->k * d->Cout;
    const int R = d->shared_weights ? d->U * pl.NS : pl.NS;          // partial rows summed into one dw row
    const long long total = (long long)(d->shared_weights ? 1 : d->U) * n_out;
    convk_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, dw, R, n_out, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
