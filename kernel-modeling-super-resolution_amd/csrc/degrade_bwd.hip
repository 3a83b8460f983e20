// Backward of the fused degradation (V-B configuration: strided sampling, kernels applied as given) with respect to
// the blur kernels and the noise sigmas -- what the MoE / KernelGAN trainers differentiate
// (ContentAdaptiveDegradation.forward, muti_kernel/train_gemini.py:117-138; loss_G.backward() :219-222).
//
//   out[b,c,i,j] = sum_{u,v} K[b,c,u,v] * x0[b,c,F*i+u-ph,F*j+v-pw] + sigma[b,c] * eps[b,c,i,j]
//   dK[b,c,u,v]  = sum_{i,j} g[b,c,i,j] * x0[b,c,F*i+u-ph,F*j+v-pw]
//   dsigma[b,c]  = sum_{i,j} g[b,c,i,j] * eps[b,c,i,j]          (eps regenerated from the forward's Philox stream)
//
// Each (sample, band) plane has its own image, so this is a batched matrix-vector product: no operand reuse across
// planes, HBM-bound in principle (the HR batch is read once, like the forward).  Workgroup = one 16x16 tile of LR pixels
// of one plane: the HR halo tile ((16-1)F+kh rows) and the 256 upstream gradients go to LDS, threads accumulate taps
// over the tile (generic kernel: one tap per thread; V-B kernel below: 4 taps per thread, register blocked); per-tile
// partial sums go to the workspace and a second kernel adds them in a fixed order (bit-reproducible, no float atomics).
// Measured at the config-3 batch (HR [32,5,1024,1024]): 0.46 ms = 1.5 TB/s algorithmic (0.19 of HBM peak), 0.55 ms with
// dsigma -- bounded by the tile fill's index arithmetic and the per-tile barriers, not by HBM; a strip-persistent
// variant with register prefetch was slower (0.59 ms: 128+ VGPRs, 2 workgroups per CU).
#include "kmsr_common.h"
#include "kmsr_noise.h"

#include <algorithm>

namespace kmsr {
namespace {

constexpr int BT = 16;          // LR tile edge
constexpr int BTHREADS = 512;

struct BwdParams {
    const float* hr;
    long long sN, sC, sH;
    const float* g;
    float* part;           // [planes][tiles][ntaps + 1]
    int C, H, W, h, w, F, kh, kw, ph, pw, pad_zero;
    int tiles_x, tiles, rows, cols, pitch, S;
    unsigned long long seed;
    unsigned offset;
    int want_sigma;
};

using f32x4 = __attribute__((ext_vector_type(4))) float;

// TF/TK > 0: factor and (square) kernel size are compile-time constants (V-B: 4, 13), so the index arithmetic of the
// tile fill and of the tap loop needs no runtime divisions; TF = TK = 0: generic.
template <int TF, int TK>
__global__ void __launch_bounds__(BTHREADS) degrade_bwd_kernel(BwdParams pr) {
    struct Geo {
        const BwdParams& q;
        __device__ int F() const { return TF ? TF : q.F; }
        __device__ int kh() const { return TK ? TK : q.kh; }
        __device__ int kw() const { return TK ? TK : q.kw; }
        __device__ int rows() const { return TK ? (BT - 1) * TF + TK : q.rows; }
        __device__ int cols() const { return TK ? (BT - 1) * TF + TK : q.cols; }
        __device__ int pitch() const { return TK ? (((BT - 1) * TF + TK) | 1) : q.pitch; }
        __device__ int S() const { return TK ? (BTHREADS / (TK * TK) < BT ? BTHREADS / (TK * TK) : BT) : q.S; }
    };
    const BwdParams& p = pr;
    const Geo G{pr};
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                   // [rows][pitch]
    float* gs = smem + G.rows() * G.pitch();                // [BT][BT]
    float* red = gs + BT * BT;                          // [S][ntaps] + sigma scratch [64]
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % p.tiles, plane = blockIdx.x / p.tiles;
    const int n = plane / p.C, c = plane % p.C;
    const int i0 = (tile / p.tiles_x) * BT, j0 = (tile % p.tiles_x) * BT;
    const float* __restrict__ x = p.hr + n * p.sN + c * p.sC;
    const int ntaps = G.kh() * G.kw();

    // ---- stage the HR halo tile (zero or replicate padding) and the gradient tile (0 outside the LR image)
    const int y0 = i0 * G.F() - G.kh() / 2, x0 = j0 * G.F() - G.kw() / 2;
    auto fetch = [&](int e) -> float {
        const int r = e / G.cols(), q = e - r * G.cols();
        int gy = y0 + r, gx = x0 + q;
        if (p.pad_zero) {
            const bool ok = e < G.rows() * G.cols() && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float t = x[ok ? gy * p.sH + gx : 0];      // safe address + select: loads stay back to back
            return ok ? t : 0.f;
        }
        gy = min(max(gy, 0), p.H - 1);
        gx = min(max(gx, 0), p.W - 1);
        return x[gy * p.sH + gx];
    };
    auto put = [&](int e, float v) {
        const int r = e / G.cols(), q = e - r * G.cols();
        if (e < G.rows() * G.cols()) xs[r * G.pitch() + q] = v;
    };
    if constexpr (TK > 0) {
        constexpr int NE = ((BT - 1) * TF + TK) * ((BT - 1) * TF + TK), NT = (NE + BTHREADS - 1) / BTHREADS;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = fetch(tid + t * BTHREADS);      // all loads in flight before the first store
#pragma unroll
        for (int t = 0; t < NT; ++t) put(tid + t * BTHREADS, v[t]);
    } else {
        for (int e = tid; e < G.rows() * G.cols(); e += BTHREADS) put(e, fetch(e));
    }
    const float* __restrict__ gp = p.g + (long long)plane * p.h * p.w;
    if (tid < BT * BT) {
        const int i = i0 + tid / BT, j = j0 + tid % BT;
        const bool ok = i < p.h && j < p.w;
        const float t = gp[ok ? (long long)i * p.w + j : 0];
        gs[tid] = ok ? t : 0.f;
    }
    __syncthreads();

    // ---- thread (tap, s): acc = sum over rows i = s (mod S), all 16 columns
    const int s = tid / ntaps, tap = tid - s * ntaps;
    float acc = 0.f;
    if (s < G.S()) {
        const int u = tap / G.kw(), v = tap - u * G.kw();
#pragma unroll 2
        for (int i = s; i < BT; i += G.S()) {
            const float* xr = xs + (i * G.F() + u) * G.pitch() + v;
            const f32x4* g4 = reinterpret_cast<const f32x4*>(gs + i * BT);
#pragma unroll
            for (int q = 0; q < BT / 4; ++q) {
                const f32x4 gv = g4[q];
                acc = fmaf(gv.x, xr[(4 * q + 0) * G.F()], acc);
                acc = fmaf(gv.y, xr[(4 * q + 1) * G.F()], acc);
                acc = fmaf(gv.z, xr[(4 * q + 2) * G.F()], acc);
                acc = fmaf(gv.w, xr[(4 * q + 3) * G.F()], acc);
            }
        }
        red[s * ntaps + tap] = acc;
    }
    // ---- dsigma partial: 64 threads, each 4 rows of one column (one Philox call yields the 4 normals of rows 4*iblk..+3)
    float* sred = red + G.S() * ntaps;
    if (p.want_sigma && tid < 64) {
        const int ib = tid / BT, jj = tid % BT;
        const int i = i0 + 4 * ib, j = j0 + jj;
        float z[4], sg = 0.f;
        eps4(p.seed, p.offset, (unsigned)plane, (unsigned)(i >> 2), (unsigned)j, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) sg = fmaf(gs[(4 * ib + k) * BT + jj], z[k], sg);   // gs is 0 outside the image
        sred[tid] = sg;
    }
    __syncthreads();
    float* out = p.part + ((long long)plane * p.tiles + tile) * (ntaps + 1);
    for (int t = tid; t < ntaps; t += BTHREADS) {
        float a = red[t];
        for (int k = 1; k < G.S(); ++k) a += red[k * ntaps + t];
        out[t] = a;
    }
    if (tid == 0) {
        float a = 0.f;
        if (p.want_sigma)
            for (int k = 0; k < 64; ++k) a += sred[k];
        out[ntaps] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// V-B specialisation (factor 4, 13x13 taps): register-blocked tap loop.  Taps v, v+4, v+8, v+12 of one kernel row read
// the SAME HR samples at neighbouring LR columns (x[4j + v + 4t] = x[4(j+t) + v]), so thread (s, u, v mod 4) loads one
// strided run of 19 samples and 16 gradients per LR row and feeds 4 accumulators: 23 LDS reads per 64 FMAs instead of
// 80 per 64.  s = 8 row splits (rows s, s+8); 8 x 13 x 4 = 416 of 512 threads take part in the tap loop; the last
// wave regenerates the noise for the sigma gradient meanwhile.
// ------------------------------------------------------------------------------------------------
constexpr int VB_F = 4, VB_K = 13, VB_ROWS = (BT - 1) * VB_F + VB_K, VB_PITCH = VB_ROWS;   // 73 (odd)
constexpr int VB_S = 8, VB_TAPS = VB_K * VB_K;
constexpr int VB_XS = VB_ROWS * VB_PITCH + 8;      // + slack: the 19th sample of the v mod 4 > 0 threads is read, not used
// ALIGNED tile fill (W % 4 == 0, 16-byte aligned rows, zero padding): the 73-column window starts at HR column 4 j0 - 6, so the
// aligned window [4 j0 - 8, 4 j0 + 72) = 20 float4 per row covers it with every float4 wholly inside or outside the image:
// 3 global_load_dwordx4 + 3 ds_write_b128 per thread instead of 11 dword loads / stores with per-element index arithmetic
// (the fill, not HBM, bounded the scalar version: 0.19 of the HBM peak).  LDS pitch 84 floats (16-byte rows; 84 = 20 mod 32 keeps
// the 13 tap rows a wave touches on different banks except u, u + 8).
constexpr int VB_AQ = 20, VB_APITCH = 84, VB_AOFF = 2, VB_AXS = VB_ROWS * VB_APITCH;

template <bool ALIGNED>
__global__ void __launch_bounds__(BTHREADS) degrade_bwd_vb_kernel(BwdParams p) {
    __shared__ __attribute__((aligned(16))) float xs[ALIGNED ? VB_AXS : VB_XS];
    __shared__ __attribute__((aligned(16))) float gs[BT * BT];
    __shared__ float red[VB_S * VB_TAPS + 64];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % p.tiles, plane = blockIdx.x / p.tiles;
    const int n = plane / p.C, c = plane % p.C;
    const int i0 = (tile / p.tiles_x) * BT, j0 = (tile % p.tiles_x) * BT;
    const float* __restrict__ x = p.hr + n * p.sN + c * p.sC;
    const int y0 = i0 * VB_F - VB_K / 2, x0 = j0 * VB_F - VB_K / 2;

    const float* __restrict__ gp = p.g + (long long)plane * p.h * p.w;
    float gv = 0.f;
    if (tid < BT * BT) {
        const int i = i0 + tid / BT, j = j0 + tid % BT;
        const bool ok = i < p.h && j < p.w;
        const float t = gp[ok ? (long long)i * p.w + j : 0];
        gv = ok ? t : 0.f;
    }
    if constexpr (ALIGNED) {
        constexpr int NE4 = VB_ROWS * VB_AQ, NT4 = (NE4 + BTHREADS - 1) / BTHREADS;
        f32x4 v4[NT4];
#pragma unroll
        for (int t = 0; t < NT4; ++t) {         // all loads in flight before the first LDS store
            const int e = tid + t * BTHREADS;
            const int r = e / VB_AQ, q = e - r * VB_AQ;
            const int gy = y0 + r, gx = x0 - VB_AOFF + 4 * q;
            const bool ok = e < NE4 && gy >= 0 && gy < p.H && gx >= 0 && gx + 3 < p.W;
            const f32x4 tv = *reinterpret_cast<const f32x4*>(x + (ok ? gy * p.sH + gx : 0));
            v4[t] = ok ? tv : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < NT4; ++t) {
            const int e = tid + t * BTHREADS;
            const int r = e / VB_AQ, q = e - r * VB_AQ;
            if (e < NE4) *reinterpret_cast<f32x4*>(xs + r * VB_APITCH + 4 * q) = v4[t];
        }
    } else {
        constexpr int NE = VB_ROWS * VB_ROWS, NT = (NE + BTHREADS - 1) / BTHREADS;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {              // all loads in flight before the first LDS store
            const int e = tid + t * BTHREADS;
            const int r = e / VB_ROWS, q = e - r * VB_ROWS;
            int gy = y0 + r, gx = x0 + q;
            if (p.pad_zero) {
                const bool ok = e < NE && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const float tv = x[ok ? gy * p.sH + gx : 0];
                v[t] = ok ? tv : 0.f;
            } else {
                gy = min(max(gy, 0), p.H - 1);
                gx = min(max(gx, 0), p.W - 1);
                v[t] = x[gy * p.sH + gx];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int e = tid + t * BTHREADS;
            if (e < NE) xs[e] = v[t];               // pitch == row length: the tile is stored densely
        }
        if (tid < 8) xs[NE + tid] = 0.f;
    }
    if (tid < BT * BT) gs[tid] = gv;
    __syncthreads();

    const int vm = tid & 3, u = (tid >> 2) % VB_K, s = tid / (4 * VB_K);
    if (s < VB_S) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < BT / VB_S; ++rr) {
            const int i = s + rr * VB_S;
            const float* xr = xs + (i * VB_F + u) * (ALIGNED ? VB_APITCH : VB_PITCH) + (ALIGNED ? VB_AOFF : 0) + vm;
            float xv[BT + 3];
#pragma unroll
            for (int j = 0; j < BT + 3; ++j) xv[j] = xr[VB_F * j];
            const f32x4* g4 = reinterpret_cast<const f32x4*>(gs + i * BT);
#pragma unroll
            for (int q = 0; q < BT / 4; ++q) {
                const f32x4 gq = g4[q];
                const float gj[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = fmaf(gj[k], xv[4 * q + k + t], acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int vv = vm + 4 * t;
            if (vv < VB_K) red[s * VB_TAPS + u * VB_K + vv] = acc[t];
        }
    }
    float* sred = red + VB_S * VB_TAPS;
    if (p.want_sigma && tid >= 448) {           // the last wave is idle in the tap loop: it owns the sigma partial
        const int l = tid - 448, ib = l / BT, jj = l % BT;
        const int i = i0 + 4 * ib, j = j0 + jj;
        float z[4], sg = 0.f;
        eps4(p.seed, p.offset, (unsigned)plane, (unsigned)(i >> 2), (unsigned)j, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) sg = fmaf(gs[(4 * ib + k) * BT + jj], z[k], sg);   // gs is 0 outside the image
        sred[l] = sg;
    }
    __syncthreads();
    float* out = p.part + ((long long)plane * p.tiles + tile) * (VB_TAPS + 1);
    if (tid < VB_TAPS) {
        float a = red[tid];
#pragma unroll
        for (int k = 1; k < VB_S; ++k) a += red[k * VB_TAPS + tid];
        out[tid] = a;
    }
    if (tid == 256) {
        float a = 0.f;
        if (p.want_sigma)
            for (int k = 0; k < 64; ++k) a += sred[k];
        out[VB_TAPS] = a;
    }
}

// dk[plane][tap] = sum_tiles part[plane][tile][tap]; dsigma[plane] likewise.  Fixed summation order (8 interleaved
// running sums per output, combined pairwise) so results are bit-reproducible; the 8 loads of a round are independent,
// which keeps the tile loop from serialising on memory latency.
__global__ void degrade_bwd_reduce_kernel(const float* __restrict__ part, int tiles, int ntaps, float* __restrict__ dk,
                                          float* __restrict__ dsigma) {
    const int plane = blockIdx.x;
    const long long pitch = ntaps + 1;
    const float* pp = part + (long long)plane * tiles * pitch;
    for (int t = threadIdx.x; t <= ntaps; t += blockDim.x) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= tiles; k += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += pp[(k + q) * pitch + t];
        }
        for (int q = 0; k < tiles; ++k, ++q) a[q] += pp[k * pitch + t];
        const float r = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        if (t < ntaps) {
            if (dk) dk[(long long)plane * ntaps + t] = r;
        } else if (dsigma) {
            dsigma[plane] = r;
        }
    }
}

int check_desc(const kmsr_degrade_desc* d) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad HR shape");
    if (d->kh <= 0 || d->kw <= 0 || d->kh > 31 || d->kw > 31 || !(d->kh & 1) || !(d->kw & 1))
        return fail(KMSR_E_BADARG, "kernel size must be odd and <= 31 (got %dx%d)", d->kh, d->kw);
    if (d->down_mode != KMSR_DOWN_STRIDE || d->normalize)
        return fail(KMSR_E_UNSUPPORTED, "degrade backward is implemented for down_mode STRIDE without normalisation (V-B)");
    if (d->factor < 1 || d->factor > 8) return fail(KMSR_E_UNSUPPORTED, "degrade backward: factor %d not in [1, 8]", d->factor);
    if (d->pad_mode != KMSR_PAD_ZERO && d->pad_mode != KMSR_PAD_REPLICATE) return fail(KMSR_E_BADARG, "bad pad_mode");
    return KMSR_OK;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_degrade_bwd_workspace_bytes(const kmsr_degrade_desc* d) {
    if (check_desc(d) != KMSR_OK) return 0;
    const int h = ceil_div(d->H, d->factor), w = ceil_div(d->W, d->factor);
    const size_t tiles = (size_t)ceil_div(h, BT) * ceil_div(w, BT);
    return (size_t)d->N * d->C * tiles * (d->kh * d->kw + 1) * sizeof(float);
}

extern "C" int kmsr_degrade_bwd_f32(const kmsr_degrade_desc* d, const float* hr, const float* grad_lr, float* dkernels,
                                    float* dsigma, void* workspace, size_t workspace_bytes, void* stream_) {
    const int rc = check_desc(d);
    if (rc != KMSR_OK) return rc;
    if (!hr || !grad_lr || !workspace || (!dkernels && !dsigma)) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (dsigma && d->noise_mode != KMSR_NOISE_GAUSSIAN)
        return fail(KMSR_E_BADARG, "dsigma requested but the forward used no Gaussian noise");
    if (workspace_bytes < kmsr_degrade_bwd_workspace_bytes(d))
        return fail(KMSR_E_NOMEM, "degrade backward workspace too small: %zu < %zu", workspace_bytes,
                    kmsr_degrade_bwd_workspace_bytes(d));
    if (d->hr_stride_c * d->C >= (1ll << 40)) return fail(KMSR_E_UNSUPPORTED, "HR strides too large");
    BwdParams p{};
    p.hr = hr; p.sN = d->hr_stride_n; p.sC = d->hr_stride_c; p.sH = d->hr_stride_h;
    p.g = grad_lr;
    p.part = static_cast<float*>(workspace);
    p.C = d->C; p.H = d->H; p.W = d->W; p.F = d->factor; p.kh = d->kh; p.kw = d->kw; p.ph = d->kh / 2; p.pw = d->kw / 2;
    p.h = ceil_div(d->H, d->factor);
    p.w = ceil_div(d->W, d->factor);
    p.pad_zero = d->pad_mode == KMSR_PAD_ZERO;
    p.tiles_x = ceil_div(p.w, BT);
    p.tiles = p.tiles_x * ceil_div(p.h, BT);
    p.rows = (BT - 1) * p.F + p.kh;
    p.cols = (BT - 1) * p.F + p.kw;
    p.pitch = p.cols | 1;            // odd pitch: the 13 rows of taps a wave touches fall on different banks
    const int ntaps = p.kh * p.kw;
    p.S = std::max(1, std::min(BT, BTHREADS / ntaps));
    if (ntaps > BTHREADS) return fail(KMSR_E_UNSUPPORTED, "degrade backward: kernels above 22x22 taps are not implemented");
    p.seed = d->seed; p.offset = d->offset;
    p.want_sigma = dsigma != nullptr;
    const size_t lds = ((size_t)p.rows * p.pitch + BT * BT + (size_t)p.S * ntaps + 64) * sizeof(float);
    if (lds > 160 * 1024) return fail(KMSR_E_UNSUPPORTED, "degrade backward: tile does not fit LDS");
    const long long planes = (long long)d->N * d->C;
    const long long nblocks = planes * p.tiles;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "grid too large");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (p.F == VB_F && p.kh == VB_K && p.kw == VB_K) {
        const bool aligned = p.pad_zero && d->W % 4 == 0 && p.sH % 4 == 0 && p.sC % 4 == 0 && p.sN % 4 == 0 &&
                             reinterpret_cast<uintptr_t>(hr) % 16 == 0;
        if (aligned) degrade_bwd_vb_kernel<true><<<(int)nblocks, BTHREADS, 0, stream>>>(p);
        else degrade_bwd_vb_kernel<false><<<(int)nblocks, BTHREADS, 0, stream>>>(p);
    } else {
        if (lds > 64 * 1024)
            KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(degrade_bwd_kernel<0, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        degrade_bwd_kernel<0, 0><<<(int)nblocks, BTHREADS, lds, stream>>>(p);
    }
    KMSR_HIP_CHECK(hipGetLastError());
    degrade_bwd_reduce_kernel<<<(int)planes, 256, 0, stream>>>(p.part, p.tiles, ntaps, dkernels, dsigma);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
