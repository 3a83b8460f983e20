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
// Three kernels: the strip-streaming V-B kernel (factor 4, 13x13, W % 16 == 0, zero padding -- the trainer's shapes; 0.184 ms at the
// config-3 batch HR [32,5,1024,1024] = 0.48 of the HBM peak, profiles/r02_degrade_bwd_strip.log), the per-tile V-B kernel for the
// other alignments / replicate padding (0.257 ms), and the generic per-tile kernel for every other factor / kernel size.
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
    unsigned nz_base, nz_step;  // noise plane of local (n, c) = nz_base + n * nz_step + c  (same addressing as the forward)
    int want_sigma;
};

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

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
        eps4(p.seed, p.offset, p.nz_base + (unsigned)n * p.nz_step + (unsigned)c, (unsigned)(i >> 2), (unsigned)j, z);
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
        eps4(p.seed, p.offset, p.nz_base + (unsigned)n * p.nz_step + (unsigned)c, (unsigned)(i >> 2), (unsigned)j, z);
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

// ------------------------------------------------------------------------------------------------
// Strip-streaming form of the V-B kernel (W % 16 == 0, zero padding): the default for the trainer's shapes.
// The per-tile chain  load -> barrier -> taps -> barrier -> reduce  of the kernel above leaves a workgroup without bytes in flight
// for most of its life (0.35 of the HBM peak; timing-only ablations, tools/diag/make_ablate_dbwd.py: loads alone 0.165 ms, taps
// alone 0.137 ms, together 0.255 ms).  Here a workgroup owns a COLUMN STRIP (16 LR columns = 80 aligned HR columns) of a plane
// and walks it top to bottom in blocks of 64 HR rows (16 LR rows):
//   * a three-slot ring of blocks in LDS, filled by LDS-DMA (no staging registers): while block k is consumed, block k+1 has
//     (mostly) landed and block k+2 is in flight -- every HR row of the strip is fetched exactly once, the 9 halo rows a step needs
//     below its own block are the head of the next ring slot;
//   * counted waits: a wave issues the same 3 DMA instructions per block (lanes outside the image read a zero page, so the count
//     never depends on the data), hence `vmcnt(3)` = "everything but my share of the newest block has landed";
//   * the 169 tap sums (and the sigma sum) stay in registers for the whole strip: one reduction / store per strip instead of
//     one per tile, and 16x fewer partials for the second kernel.
// LDS-DMA writes 64 lanes x 16 bytes contiguously, so a ring row is 21 float4 (20 data + 1 filler, never read): pitch 84 floats
// as above, a block = 64 rows = exactly 21 instructions (waves 0..6 x 3); wave 7 fetches the block's 256 gradients (one
// instruction) and owns the sigma sum.  Workgroups of one XCD take neighbouring strips, so the 16 shared columns hit its L2.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
constexpr int VS_BR = BT * VB_F;                     // 64 HR rows per block
constexpr int VS_Q = VB_APITCH / 4;                  // 21 float4 per ring row
constexpr int VS_BF = VS_BR * VB_APITCH;             // 5376 floats per ring slot
constexpr int VS_NI = 3;                             // DMA instructions per wave (0..6) and block
constexpr int VS_SLOTS = 3;
constexpr int VS_UP = 7;                             // tap-row pairs (u, u + 7); u + 7 = 13 is computed and dropped
static_assert(7 * VS_NI * 64 * 4 == VS_BF, "a block is 21 whole DMA instructions");
__device__ __attribute__((aligned(16))) float g_zero_page[4];      // zero-initialised; what lanes outside the image fetch

struct StripCursor {          // block `kb` of this workgroup's job `m`
    int m, kb, plane, strip;
    bool live;
};

__global__ void __launch_bounds__(BTHREADS) degrade_bwd_vb_strip_kernel(BwdParams p, int jobs) {
    // ONE LDS object: with a second __shared__ array beside the LDS-DMA target hipcc (ROCm 7.2) waits vmcnt(0) before the first
    // ds_read of every step, i.e. for the block that was just issued (checked in the .s)
    __shared__ __attribute__((aligned(16))) float lds[VS_SLOTS * VS_BF + VS_SLOTS * BT * BT + BT * VB_TAPS + 64];
    float (*xs)[VS_BF] = reinterpret_cast<float (*)[VS_BF]>(lds);
    float (*gs)[BT * BT] = reinterpret_cast<float (*)[BT * BT]>(lds + VS_SLOTS * VS_BF);
    float* red = lds + VS_SLOTS * VS_BF + VS_SLOTS * BT * BT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    // workgroups are dealt to the 8 XCDs round-robin: give the workgroups of one XCD consecutive jobs
    const int v = G % 8 == 0 ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (v >= jobs) return;                           // block-uniform
    const int strips = p.tiles_x, nsteps = p.tiles / p.tiles_x, NB = nsteps + 1;
    const int njobs = (jobs - v + G - 1) / G;        // jobs v, v + G, v + 2 G, ...
    const int TB = njobs * NB;

    // this thread's three float4 of every block (waves 0..6), or its float4 of the gradients (wave 7)
    int roff[VS_NI], rr[VS_NI], cq[VS_NI];
#pragma unroll
    for (int k = 0; k < VS_NI; ++k) {
        const int e = ((wv % 7) * VS_NI + k) * 64 + lane;
        rr[k] = e / VS_Q;
        cq[k] = 4 * (e - rr[k] * VS_Q);
        roff[k] = rr[k] * (int)p.sH + cq[k];
    }
    const int gi = lane >> 2, gj = 4 * (lane & 3);
    const float* zp = g_zero_page;

    auto cursor = [&](int m, int kb) {
        StripCursor c{m, kb, 0, 0, m < njobs};
        const int job = v + G * (c.live ? m : 0);
        c.plane = job / strips;
        c.strip = job - c.plane * strips;
        return c;
    };
    auto advance = [&](StripCursor c) { return c.kb + 1 < NB ? StripCursor{c.m, c.kb + 1, c.plane, c.strip, c.live} : cursor(c.m + 1, 0); };
    auto issue = [&](const StripCursor& c, int slot) {          // always the same number of vector-memory instructions per wave
        if (wv < 7) {
            const int n = c.plane / p.C, ch = c.plane - n * p.C;
            const float* __restrict__ x = p.hr + n * p.sN + ch * p.sC;
            const int y0 = c.kb * VS_BR - VB_K / 2, xa = c.strip * (BT * VB_F) - VB_K / 2 - VB_AOFF;
            const float* __restrict__ xo = x + (long long)y0 * p.sH + xa;
#pragma unroll
            for (int k = 0; k < VS_NI; ++k) {
                const int gy = y0 + rr[k], gx = xa + cq[k];
                const bool ok = c.live && cq[k] < 4 * VB_AQ && gy >= 0 && gy < p.H && gx >= 0 && gx + 3 < p.W;
                float* dst = xs[slot] + (wv * VS_NI + k) * 256;         // wave-uniform; the lane's slot is dst + 4 * lane
                __builtin_amdgcn_global_load_lds((glb_void_t*)(ok ? xo + roff[k] : zp), (lds_void_t*)dst, 16, 0, 0);
            }
        } else {
            const int i = c.kb * BT + gi, j = c.strip * BT + gj;
            const bool ok = c.live && i < p.h && j < p.w;               // w % 4 == 0: a float4 is wholly inside or outside
            const float* __restrict__ gp = p.g + (long long)c.plane * p.h * p.w + (long long)i * p.w + j;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(ok ? gp : zp), (lds_void_t*)gs[slot], 16, 0, 0);
        }
    };

    // tap loop: thread (LR row i of the step, tap rows u and u + 7, v mod 4) -- 16 x 7 x 4 = 448 threads = waves 0..6.  The two
    // tap rows share the gradient, so one v_pk_fma_f32 (gradient broadcast to both halves) does two of the 128 FMAs per thread and
    // step; taps v, v + 4, v + 8, v + 12 read the same samples at neighbouring LR columns (19 samples feed 16 x 4 products).
    const int vm = tid & 3, up = (tid >> 2) % VS_UP, li = tid / (4 * VS_UP);
    const int row0 = li * VB_F + up, row1 = row0 + VS_UP;                // ring rows of the two tap rows (row1 <= 73: inside slot t + 1)
    f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float sg = 0.f;

    StripCursor cur = cursor(0, 0), nxt = advance(cur), nn = advance(nxt);
    issue(cur, 0);
    issue(nxt, 1);
    int sl = 0;                                                          // ring slot of the current block
    for (int t = 0; t < TB; ++t) {
        // landed before this step may run: blocks <= t, and the first 10 rows of block t + 1 (inside wave 0's share of it)
        asm volatile("" ::: "memory");
        if (wv == 0) __builtin_amdgcn_s_waitcnt(0x0070 | 0);            // vmcnt(0), lgkmcnt(0)
        else if (wv < 7) __builtin_amdgcn_s_waitcnt(0x0070 | VS_NI);    // vmcnt(3): all but my share of block t + 1
        else __builtin_amdgcn_s_waitcnt(0x0070 | 1);                    // vmcnt(1): all but the gradients of block t + 1
        __builtin_amdgcn_s_barrier();                // NOT __syncthreads(): its fence would drain vmcnt to 0, i.e. wait for block t + 1
        asm volatile("" ::: "memory");
        const int sl1 = sl == VS_SLOTS - 1 ? 0 : sl + 1, sl2 = sl1 == VS_SLOTS - 1 ? 0 : sl1 + 1;
        issue(nn, sl2);                              // slot of block t - 1: its last readers finished before this barrier
        if (cur.kb < nsteps) {
            const float* gb = gs[sl];
            if (wv < 7) {
                const float* xr0 = (row0 < VS_BR ? xs[sl] + row0 * VB_APITCH : xs[sl1] + (row0 - VS_BR) * VB_APITCH) + VB_AOFF + vm;
                const float* xr1 = (row1 < VS_BR ? xs[sl] + row1 * VB_APITCH : xs[sl1] + (row1 - VS_BR) * VB_APITCH) + VB_AOFF + vm;
                f32x2 xv[BT + 3];
#pragma unroll
                for (int j = 0; j < BT + 3; ++j) xv[j] = f32x2{xr0[VB_F * j], xr1[VB_F * j]};
                const f32x4* g4 = reinterpret_cast<const f32x4*>(gb + li * BT);
#pragma unroll
                for (int q = 0; q < BT / 4; ++q) {
                    const f32x4 gq = g4[q];
                    const float gk[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_elementwise_fma(f32x2{gk[k], gk[k]}, xv[4 * q + k + tt], acc[tt]);
                }
            } else if (p.want_sigma) {               // the last wave is idle in the tap loop: it owns the sigma sum
                const int ib = lane / BT, jj = lane % BT;
                const int i = cur.kb * BT + 4 * ib, j = cur.strip * BT + jj;
                float z[4];
                const unsigned sn = (unsigned)(cur.plane / p.C);
                eps4(p.seed, p.offset, p.nz_base + sn * p.nz_step + ((unsigned)cur.plane - sn * (unsigned)p.C), (unsigned)(i >> 2), (unsigned)j, z);
#pragma unroll
                for (int k = 0; k < 4; ++k) sg = fmaf(gb[(4 * ib + k) * BT + jj], z[k], sg);   // gradients are 0 outside the image
            }
        } else {                                     // the strip's last (halo-only) block: hand the sums over
            if (wv < 7) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int vv = vm + 4 * tt;
                    if (vv < VB_K) {
                        red[li * VB_TAPS + up * VB_K + vv] = acc[tt].x;
                        if (up + VS_UP < VB_K) red[li * VB_TAPS + (up + VS_UP) * VB_K + vv] = acc[tt].y;
                    }
                    acc[tt] = f32x2{0.f, 0.f};
                }
            } else {
                red[BT * VB_TAPS + lane] = sg;
                sg = 0.f;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only: the blocks in flight stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float* out = p.part + ((long long)cur.plane * strips + cur.strip) * (VB_TAPS + 1);
            if (tid < VB_TAPS) {
                float a = red[tid];
#pragma unroll
                for (int k = 1; k < BT; ++k) a += red[k * VB_TAPS + tid];
                out[tid] = a;
            }
            if (tid == 256) {
                float a = 0.f;
                if (p.want_sigma)
                    for (int k = 0; k < 64; ++k) a += red[BT * VB_TAPS + k];
                out[VB_TAPS] = a;
            }
        }
        cur = nxt;
        nxt = nn;
        nn = advance(nn);
        sl = sl1;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070 | 0);          // the two look-ahead fetches past the end (zero page) must not outlive the LDS
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

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
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
    p.nz_step = (d->noise_sample_step ? d->noise_sample_step : 1u) * (unsigned)d->C;
    p.nz_base = d->noise_sample_base * (unsigned)d->C;
    p.want_sigma = dsigma != nullptr;
    const size_t lds = ((size_t)p.rows * p.pitch + BT * BT + (size_t)p.S * ntaps + 64) * sizeof(float);
    if (lds > 160 * 1024) return fail(KMSR_E_UNSUPPORTED, "degrade backward: tile does not fit LDS");
    const long long planes = (long long)d->N * d->C;
    const long long nblocks = planes * p.tiles;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "grid too large");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int reduce_tiles = p.tiles;
    if (p.F == VB_F && p.kh == VB_K && p.kw == VB_K) {
        const bool aligned = p.pad_zero && d->W % 4 == 0 && p.sH % 4 == 0 && p.sC % 4 == 0 && p.sN % 4 == 0 &&
                             reinterpret_cast<uintptr_t>(hr) % 16 == 0;
        const bool strip = aligned && d->W % 16 == 0 && reinterpret_cast<uintptr_t>(grad_lr) % 16 == 0 &&
                           (long long)d->H * p.sH < (1ll << 31);
        if (strip) {
            const int jobs = (int)(planes * p.tiles_x);
            const int wgs = std::min(jobs, 2 * cu_count());              // 78 KB of LDS each: two per CU
            degrade_bwd_vb_strip_kernel<<<wgs, BTHREADS, 0, stream>>>(p, jobs);
            reduce_tiles = p.tiles_x;                                    // one partial per column strip
        } else if (aligned) degrade_bwd_vb_kernel<true><<<(int)nblocks, BTHREADS, 0, stream>>>(p);
        else degrade_bwd_vb_kernel<false><<<(int)nblocks, BTHREADS, 0, stream>>>(p);
    } else {
        if (lds > 64 * 1024)
            KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(degrade_bwd_kernel<0, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        degrade_bwd_kernel<0, 0><<<(int)nblocks, BTHREADS, lds, stream>>>(p);
    }
    KMSR_HIP_CHECK(hipGetLastError());
    degrade_bwd_reduce_kernel<<<(int)planes, 256, 0, stream>>>(p.part, reduce_tiles, ntaps, dkernels, dsigma);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
