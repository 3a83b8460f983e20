// Gradient legs of the small-channel k x k convolution family (semantics: include/kmsr.h, kmsr_convk_*).
//   dgrad: dx = adjoint of [pad -> conv] applied to dy.  Zero padding: the same convolution with flipped, transposed taps.
//          Reflect padding: additionally every padded position folds back onto the interior pixel it mirrors, so a pixel
//          within k/2 of a border sums up to 3 x 3 mirrored read positions per tap (workgroup-uniform fast path elsewhere).
//   wgrad: dw[ci,tap,co] = sum_pixels dy[co,p] * xpad[ci,p+tap].  Per (unit, strip) workgroup: x tile (padded) and the dy
//          tile ([pixel][co], co fastest) in LDS; a thread owns NU (ci,tap) pairs x CB consecutive co and walks the 256
//          pixels of a tile with 1 b32 + CB/4 b128 LDS reads per CB FMAs.  Strip partials are summed in a fixed order.
// Thread-per-pixel / thread-per-output VALU code: these layers have 1..64 channels (KernelGAN generator chains and the
// patch critic), far below an MFMA tile.
#include <algorithm>

#include "kmsr_common.h"

namespace kmsr {
namespace {

constexpr int RT = 16;

__device__ __forceinline__ int reflect_idx(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad.  dy [U,Cdy,H,W]; wT [U or 1][Cdy][k*k][CDX] (forward weight w[cdx][tap][cdy] transposed, taps NOT flipped);
// dx [U,CDX,H,W].  LDS tile dy[Cdy][IT][IT] with origin (ty0-P, tx0-P), zero outside the image.
// dx[y,x] = sum_{vy in S(y), vx in S(x)} sum_{c,a,b} wT[c][a*k+b][:] * dy0[c][vy+P-a][vx+P-b]
// S(y) = {y} (zero) or {y, -y if 1<=y<=P, 2(H-1)-y if H-1-P<=y<=H-2} (reflect).
template <int KS, int CDX, bool REFLECT>
__global__ void __launch_bounds__(256) convk_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wT,
                                                          float* __restrict__ dx, int Cdy, int H, int W, int tiles_x,
                                                          int tiles, long long wstride) {
    constexpr int P = KS / 2, IT = RT + KS - 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [Cdy][IT][IT]
    const int tid = threadIdx.x;
    const int u = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int ty0 = (t / tiles_x) * RT, tx0 = (t % tiles_x) * RT;
    const float* __restrict__ du = dy + (long long)u * Cdy * H * W;
    const int n_el = Cdy * IT * IT;
    for (int e0 = 0; e0 < n_el; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = min(e0 + tid + 256 * q, n_el - 1);
            const int c = e / (IT * IT), rem = e - c * (IT * IT), r = rem / IT, cc = rem - r * IT;
            const int gy = ty0 + r - P, gx = tx0 + cc - P;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[q] = in ? du[((long long)c * H + gy) * W + gx] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + tid + 256 * q;
            if (e < n_el) tile[e] = v[q];
        }
    }
    __syncthreads();
    const int py = tid / RT, px = tid % RT;
    const int y = ty0 + py, xx = tx0 + px;
    kmsr_f32x2 acc2[(CDX + 1) / 2];
    float accr = 0.f;
#pragma unroll
    for (int q = 0; q < (CDX + 1) / 2; ++q) acc2[q] = kmsr_f32x2{0.f, 0.f};
    const float* __restrict__ wu = wT + (long long)u * wstride;
    // rows / cols of the LDS tile that tap (a,b) reads for source v:  (v - t0) + 2P - a
    bool border = false;
    if constexpr (REFLECT) border = ty0 <= P || ty0 + RT >= H - 1 - P || tx0 <= P || tx0 + RT >= W - 1 - P;   // block-uniform
    if (!border) {
        for (int c = 0; c < Cdy; ++c) {
            const float* tc = tile + c * IT * IT + (py + 2 * P) * IT + (px + 2 * P);
#pragma unroll
            for (int a = 0; a < KS; ++a)
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    const float v = tc[-a * IT - b];
                    const float* __restrict__ wt = wu + ((long long)c * KS * KS + a * KS + b) * CDX;
fma_row<CDX>(wt, v, acc2, accr);
                }
        }
    } else {
        int ry[3], rx[3];                              // tile row / col of each source before the tap shift; < -64 = absent
        ry[0] = py + 2 * P;
        ry[1] = (y >= 1 && y <= P) ? -y - ty0 + 2 * P : -1000;
        ry[2] = (y >= H - 1 - P && y <= H - 2) ? 2 * (H - 1) - y - ty0 + 2 * P : -1000;
        rx[0] = px + 2 * P;
        rx[1] = (xx >= 1 && xx <= P) ? -xx - tx0 + 2 * P : -1000;
        rx[2] = (xx >= W - 1 - P && xx <= W - 2) ? 2 * (W - 1) - xx - tx0 + 2 * P : -1000;
        for (int c = 0; c < Cdy; ++c) {
            const float* tc = tile + c * IT * IT;
            for (int a = 0; a < KS; ++a)
                for (int b = 0; b < KS; ++b) {
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int r = ry[i] - a, cc = rx[j] - b;
                            if (r >= 0 && r < IT && cc >= 0 && cc < IT) v += tc[r * IT + cc];
                        }
                    const float* __restrict__ wt = wu + ((long long)c * KS * KS + a * KS + b) * CDX;
fma_row<CDX>(wt, v, acc2, accr);
                }
        }
    }
    if (y < H && xx < W) {
        float* ou = dx + (long long)u * CDX * H * W + (long long)y * W + xx;
#pragma unroll
        for (int o = 0; o < CDX; ++o) ou[(long long)o * H * W] = fma_row_get<CDX>(acc2, accr, o);
    }
}

template <int KS, int CDX, bool REFLECT>
int launch_dgrad(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, hipStream_t st) {
    const int tiles_x = ceil_div(d->W, RT), tiles = tiles_x * ceil_div(d->H, RT);
    const size_t shmem = (size_t)d->Cout * (RT + KS - 1) * (RT + KS - 1) * sizeof(float);
    if (shmem > 64 * 1024)
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(convk_dgrad_kernel<KS, CDX, REFLECT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    convk_dgrad_kernel<KS, CDX, REFLECT><<<d->U * tiles, 256, shmem, st>>>(
        dy, wT, dx, d->Cout, d->H, d->W, tiles_x, tiles, d->shared_weights ? 0ll : (long long)d->Cout * KS * KS * CDX);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
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
    size_t shmem, ws_bytes;
};

int plan_wgrad(const kmsr_convk_desc* d, WgradPlan* pl) {
    const int IT = RT + d->k - 1, KK = d->k * d->k;
    pl->CB = d->Cout % 16 == 0 ? 16 : d->Cout % 4 == 0 ? 4 : 1;
    const int units = d->Cin * KK * (d->Cout / pl->CB);
    const int nu = ceil_div(units, 256);
    pl->NU = nu <= 1 ? 1 : nu <= 2 ? 2 : nu <= 4 ? 4 : 8;
    if (nu > 8) return fail(KMSR_E_UNSUPPORTED, "convk_wgrad: Cin*k*k*Cout = %d outputs exceed one workgroup's registers", d->Cin * KK * d->Cout);
    const int tiles = ceil_div(d->W, RT) * ceil_div(d->H, RT);
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

extern "C" int kmsr_convk_dgrad_f32(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, void* stream) {
    if (!dy || !wT || !dx) return fail(KMSR_E_BADARG, "convk_dgrad: null pointer");
    if (int rc = check_desc(d, "convk_dgrad")) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool refl = d->pad_mode == KMSR_CONVK_REFLECT;
#define KMSR_DG(K_, C_)                                                                              \
    if (d->k == K_ && d->Cin == C_)                                                                  \
        return refl ? launch_dgrad<K_, C_, true>(d, dy, wT, dx, st) : launch_dgrad<K_, C_, false>(d, dy, wT, dx, st);
    KMSR_DG(1, 1) KMSR_DG(1, 32) KMSR_DG(1, 64) KMSR_DG(3, 1) KMSR_DG(3, 32) KMSR_DG(5, 1) KMSR_DG(5, 32) KMSR_DG(7, 1)
    KMSR_DG(7, 5) KMSR_DG(7, 32)
#undef KMSR_DG
    return fail(KMSR_E_UNSUPPORTED, "convk_dgrad: k=%d Cin=%d not instantiated (k in 1,3,5,7; Cin in 1,5,32,64)", d->k, d->Cin);
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
#define KMSR_WG(CB_, NU_) \
    if (pl.CB == CB_ && pl.NU == NU_) rc = launch_wgrad<CB_, NU_>(d, pl, x, dy, part, st);
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
