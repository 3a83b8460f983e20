// Weight gradient of the 3x3 convolution in the Winograd F(2x2,3x3) domain on the fp32 matrix cores
// (semantics: include/kmsr.h, kmsr_conv3x3_wgrad_f32; this file serves even H, W % 4 == 0 with Cout > 32).
//
//   forward per 2x2 output tile:  Y = A^T [ (G g G^T) .* (B^T d B) ] A
//   =>  dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G          16 products per (co, ci, tile) instead of 36
//
// The sum over tiles (and images) is the K dimension of 16 independent [Cout x Cin] GEMMs, one per transform point xi:
//   M_xi[co][ci] += Ahat_xi[co][tile] * Vhat_xi[ci][tile]     on v_mfma_f32_16x16x4_f32  (M = 16 co, N = 16 ci, K = 4 tiles).
// A lane computes BOTH operands it feeds from raw data in LDS -- Vhat (32 adds) of its own (ci, tile) 4x4 patch and Ahat
// (12 adds) of its own (co, tile) 2x2 gradient block -- so transformed data never travels through LDS.  The minus signs of
// A are left out of Ahat and applied once to M in the final transform (s = [1,1,1,-1] per axis).
//
// Workgroup = 8 waves, persistent over a strided list of stages (stage = one tile row x 16 tiles = 2 x 32 gradient pixels);
// it keeps the whole 16 x 64co x 64ci slab in registers: wave w owns ci block (w & 3) and co blocks 2 (w >> 2) + {0,1} for
// all 16 points = 32 accumulator tiles (128 VGPRs).  Per 4 tiles a wave issues 32 MFMAs against 32 + 2 x 12 transform adds.
// Stages are double-buffered in LDS ([channel][row][col]; channel pitches 140 / 68 floats = 4 x odd mod 64 banks: a half-wave's
// b64 operand reads -- 16 channels x 2 tiles x 2 floats -- then hit all 64 banks once); the next stage is fetched with buffer
// loads (range check = zero padding)
// during the products.  Slabs are reduced over workgroups in a fixed order, then transformed to [Cout,Cin,3,3].
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int GT = 16;                 // tiles per stage
constexpr int GC = 2 * GT;             // gradient columns per stage (32)
constexpr int GXC = GC + 2;            // x columns incl. halo (34)
constexpr int GXP = 4 * GXC + 4;       // floats per x channel: 4 rows x 34 = 136, padded to 140 (= 4 x odd mod 64, see below)
constexpr int GDP = 68;                // floats per dy channel: 2 rows x 32 = 64, padded to 68 (= 4 x odd mod 64)
constexpr int GXF = 64 * GXP;          // 8704
constexpr int GDF = 64 * GDP;          // 4608
constexpr int GSTAGE = GXF + GDF;      // 13312 floats = 52 KB
constexpr int GSLAB = 16 * 64 * 64;    // floats per workgroup slab

struct WgWinoParams {
    int N, Cin, H, W, Cout;
    int cblocks, trows, n_stages;      // 32-column blocks per row, tile rows per image, stages per image
    int cogs, cigs, nblk;
};

__global__ void __launch_bounds__(512) conv3x3_wgrad_wino_kernel(const float* __restrict__ xg, const float* __restrict__ dyg,
                                                                 float* __restrict__ partial, WgWinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 2 stages
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, tl = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cib = wv & 3, cobp = wv >> 2;
    const int grp = blockIdx.y, cog = grp / p.cigs, cig = grp % p.cigs;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    const unsigned HWb = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;
    const int total = p.N * p.n_stages;

    // ---- staging maps (fixed per thread): 16-byte pieces (W % 4 == 0: a piece is inside or outside the image as a whole)
    //   x main : e = tid + 512 j, j < 4 -> piece e & 7 (columns 4 piece .. +3), row (e >> 3) & 3, channel e >> 5
    //   x halo : e = tid -> channel e >> 3, row (e >> 1) & 3, side e & 1 (column -1 / +32), one float
    //   dy     : e = tid + 512 j, j < 2 -> piece e & 7, row (e >> 3) & 1, channel e >> 4
    // The 7 loads of the next stage are issued two per MFMA block over the first 4 of the stage's 8 blocks (issue_part): all at
    // the top they sit in front of every wave's first products while the texture path takes them in (0.16 ms of 0.83 ms with 25
    // dword loads per thread back to back); spread to the last block they have not landed when the stage ends.
    const int xq = tid & 7, xrow = (tid >> 3) & 3, xch = tid >> 5;
    const int hch = tid >> 3, hrow = (tid >> 1) & 3, hside = tid & 1;
    const int dq = tid & 7, drow = (tid >> 3) & 1, dch = tid >> 4;
    f32x4 xr[4], dr[2];
    float xh;
    __amdgpu_buffer_rsrc_t rx, rdy;
    unsigned vox = 0, voh = 0, vod = 0;
    auto stage_setup = [&](int s) {
        const int n = s / p.n_stages, t = s % p.n_stages;
        const int y0 = (t / p.cblocks) * 2, x0 = (t % p.cblocks) * GC;
        const int co_n = min(p.Cout - cog * 64, 64), ci_n = min(p.Cin - cig * 64, 64);
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + ((long long)n * p.Cin + cig * 64) * HW), 0,
                                               (unsigned)ci_n * HWb, 0x00020000);
        rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyg + ((long long)n * p.Cout + cog * 64) * HW), 0,
                                                (unsigned)co_n * HWb, 0x00020000);
        {
            const int y = y0 - 1 + xrow, x = x0 + 4 * xq;
            vox = ((y >= 0 && y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB) + (unsigned)xch * HWb;
        }
        {
            const int y = y0 - 1 + hrow, x = hside ? x0 + GC : x0 - 1;
            voh = ((y >= 0 && y < H && x >= 0 && x < W) ? (unsigned)(y * W + x) * 4u : OOB) + (unsigned)hch * HWb;
        }
        {
            const int y = y0 + drow, x = x0 + 4 * dq;
            vod = ((y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB) + (unsigned)dch * HWb;
        }
    };
    auto issue_part = [&](int k) {      // k = 0 .. 7 (compile-time after unrolling)
        if (k < 4) xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vox + (unsigned)(16 * k) * HWb, 0, 0));
        else if (k < 6) dr[k - 4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, vod + (unsigned)(32 * (k - 4)) * HWb, 0, 0));
        else if (k == 6) xh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voh, 0, 0));
    };
    auto store_stage = [&](int b) {
        float* xl = lds + b * GSTAGE;
        float* dl = xl + GXF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* o = xl + (xch + 16 * j) * GXP + xrow * GXC + 1 + 4 * xq;      // column 1 + 4 piece: 4-byte aligned only
            o[0] = xr[j].x; o[1] = xr[j].y; o[2] = xr[j].z; o[3] = xr[j].w;
        }
        xl[hch * GXP + hrow * GXC + (hside ? GXC - 1 : 0)] = xh;
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(dl + (dch + 32 * j) * GDP + drow * GC + 4 * dq) = dr[j];
    };

    f32x4 acc[2][16];
#pragma unroll
    for (int pq = 0; pq < 2; ++pq)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[pq][xi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand read bases (floats): this lane's channel and first tile column pair
    const int xbase = (16 * cib + c16) * GXP + 2 * tl;
    const int dbase = GXF + (16 * (2 * cobp) + c16) * GDP + 2 * tl;

    int s = blockIdx.x, b = 0;
    if (s < total) {
        stage_setup(s);
#pragma unroll
        for (int k = 0; k < 7; ++k) issue_part(k);
        store_stage(0);
    }
    __syncthreads();
    for (; s < total; s += p.nblk) {
        const bool more = s + p.nblk < total;
        if (more) stage_setup(s + p.nblk);
        const float* st = lds + b * GSTAGE;
        // The matrix pipe and the VALU do not overlap on this chip, and every MFMA <-> VALU switch costs ~12 idle cycles, so the
        // transform adds are issued as two clusters per 4 tiles (Vhat + Ahat of co block 0 | Ahat of co block 1), each followed by
        // 16 back-to-back MFMAs; the raw reads of the next 4 tiles are issued in front of the second MFMA block.
        f32x2 d0[4], d1[4], g[2][2];
        auto read_raw = [&](int ks) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d0[r] = *reinterpret_cast<const f32x2*>(st + xbase + r * GXC + 8 * ks);
                d1[r] = *reinterpret_cast<const f32x2*>(st + xbase + r * GXC + 8 * ks + 2);
            }
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {
                g[pq][0] = *reinterpret_cast<const f32x2*>(st + dbase + pq * 16 * GDP + 8 * ks);
                g[pq][1] = *reinterpret_cast<const f32x2*>(st + dbase + pq * 16 * GDP + GC + 8 * ks);
            }
        };
        read_raw(0);
#pragma unroll
        for (int ks = 0; ks < GT / 4; ++ks) {
            float V[16], A[16];
            __builtin_amdgcn_sched_barrier(0);
            {   // ---- Vhat = B^T d B of this lane's (ci, tile) patch, on packed fp32 pairs (v_pk_add_f32: two adds per issue slot).
                //      d0[r] / d1[r] are the column pairs (0,1) / (2,3) of patch row r.  Rows first:
                const f32x2 Ra[4] = {d0[0] - d0[2], d0[1] + d0[2], d0[2] - d0[1], d0[1] - d0[3]};
                const f32x2 Rb[4] = {d1[0] - d1[2], d1[1] + d1[2], d1[2] - d1[1], d1[1] - d1[3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // then columns: (R0 - R2, R1 + R2) and (R2 - R1, R1 - R3), halves picked by op_sel
                    f32x2 v01, v23;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(Ra[i]), "v"(Rb[i]));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(v23) : "v"(Ra[i]), "v"(Rb[i]));
                    V[4 * i + 0] = v01.x; V[4 * i + 1] = v01.y; V[4 * i + 2] = v23.x; V[4 * i + 3] = v23.y;
                }
            }
            auto ahat = [&](int pq) {   // sign-free A dY A^T of this lane's (co, tile) 2x2 block [[a,b],[c,d]]; rows (p,q) packed
                const f32x2 T[4] = {g[pq][0], g[pq][0] + g[pq][1], g[pq][0] - g[pq][1], g[pq][1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x2 pm;                                                             // (p + q, p - q)
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(pm) : "v"(T[i]), "v"(T[i]));
                    A[4 * i + 0] = T[i].x; A[4 * i + 1] = pm.x; A[4 * i + 2] = pm.y; A[4 * i + 3] = T[i].y;
                }
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) asm volatile("" : "+v"(A[xi]));
            };
            ahat(0);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) asm volatile("" : "+v"(V[xi]));
            if (more && ks < 2) { issue_part(4 * ks); issue_part(4 * ks + 1); }     // block-uniform; all 7 loads are out by the
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[0][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xi], V[xi], acc[0][xi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ahat(1);
            if (ks + 1 < GT / 4) read_raw(ks + 1);              // raw registers are dead: refill them under the next 16 MFMAs
            if (more && ks < 2) { issue_part(4 * ks + 2); issue_part(4 * ks + 3); } // 4th of 8 MFMA blocks: >= 2 us to land
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[1][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xi], V[xi], acc[1][xi], 0, 0, 0);
        }
        if (more) store_stage(b ^ 1);                        // the buffer every wave finished reading before the last barrier
        __syncthreads();
        b ^= 1;
    }
    // slab[(grp * nblk + blockIdx.x)][xi][co 64][ci 64]; D lane holds rows co = 4 tl + r, column ci = c16 of its 16 x 16 tile
    float* out = partial + ((long long)grp * p.nblk + blockIdx.x) * GSLAB;
#pragma unroll
    for (int pq = 0; pq < 2; ++pq)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(xi * 64 + 16 * (2 * cobp + pq) + 4 * tl + r) * 64 + 16 * cib + c16] = acc[pq][xi][r];
}

// msum[grp][e] = sum_blk slab[grp][blk][e]   (e over 16*64*64; fixed order, 8 interleaved running sums)
__global__ void __launch_bounds__(256) wgrad_wino_sum_kernel(const float* __restrict__ partial, float* __restrict__ msum, int nblk) {
    const int e = blockIdx.x * 256 + threadIdx.x;            // < GSLAB
    const int grp = blockIdx.y;
    const float* src = partial + (long long)grp * nblk * GSLAB + e;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] += src[(long long)(b + q) * GSLAB];
    }
    for (int q = 0; b < nblk; ++b, ++q) a8[q] += src[(long long)b * GSLAB];
    msum[(long long)grp * GSLAB + e] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

// dw[co][ci][u][v] = sum_{i,j} G[i][u] s_i M[i][j] s_j G[j][v],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], s = [1,1,1,-1]
__global__ void __launch_bounds__(256) wgrad_wino_finish_kernel(const float* __restrict__ msum, float* __restrict__ dw, int Cin,
                                                                int Cout, int cigs) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // over Cout * Cin, ci fastest
    if (idx >= Cout * Cin) return;
    const int ci = idx % Cin, co = idx / Cin;
    const float* m = msum + (long long)((co / 64) * cigs + ci / 64) * GSLAB + (co % 64) * 64 + ci % 64;
    float M[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = m[(4 * i + j) * 4096];
            M[i][j] = ((i == 3) != (j == 3)) ? -v : v;
        }
    float T[3][4];                                           // T = G^T M
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float h1 = 0.5f * M[1][j], h2 = 0.5f * M[2][j];
        T[0][j] = M[0][j] + h1 + h2;
        T[1][j] = h1 - h2;
        T[2][j] = h1 + h2 + M[3][j];
    }
    float* o = dw + ((long long)co * Cin + ci) * 9;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const float h1 = 0.5f * T[u][1], h2 = 0.5f * T[u][2];
        o[u * 3 + 0] = T[u][0] + h1 + h2;
        o[u * 3 + 1] = h1 - h2;
        o[u * 3 + 2] = h1 + h2 + T[u][3];
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

}  // namespace

// ---- used by kmsr_conv3x3_wgrad_f32 (conv3x3_wgrad.hip)
bool wgrad_wino_eligible(const kmsr_conv3x3_desc* d) {
    return d->H % 2 == 0 && d->W % 4 == 0 && d->Cout > 32 && !(d->flags & KMSR_CONV_FLAG_WGRAD_DIRECT);
}

static void wgrad_wino_plan(const kmsr_conv3x3_desc* d, WgWinoParams* p) {
    p->N = d->N; p->Cin = d->Cin; p->H = d->H; p->W = d->W; p->Cout = d->Cout;
    p->cblocks = ceil_div(d->W, GC); p->trows = d->H / 2; p->n_stages = p->cblocks * p->trows;
    p->cogs = ceil_div(d->Cout, 64); p->cigs = ceil_div(d->Cin, 64);
    const long long stages = (long long)d->N * p->n_stages;
    p->nblk = (int)std::max<long long>(1, std::min<long long>(stages, cu_count() / (p->cogs * p->cigs)));   // one 8-wave workgroup per CU
}

size_t wgrad_wino_workspace_bytes(const kmsr_conv3x3_desc* d) {
    WgWinoParams p{};
    wgrad_wino_plan(d, &p);
    return (size_t)p.cogs * p.cigs * ((size_t)p.nblk + 1) * GSLAB * sizeof(float);
}

int wgrad_wino_launch(const kmsr_conv3x3_desc* d, const float* x, const float* dy, float* dw, float* ws, hipStream_t stream) {
    WgWinoParams p{};
    wgrad_wino_plan(d, &p);
    const int groups = p.cogs * p.cigs;
    float* msum = ws + (size_t)groups * p.nblk * GSLAB;
    const int shmem = 2 * GSTAGE * (int)sizeof(float);
    KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_wino_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
    conv3x3_wgrad_wino_kernel<<<dim3(p.nblk, groups), 512, shmem, stream>>>(x, dy, ws, p);
    KMSR_HIP_CHECK(hipGetLastError());
    wgrad_wino_sum_kernel<<<dim3(GSLAB / 256, groups), 256, 0, stream>>>(ws, msum, p.nblk);
    KMSR_HIP_CHECK(hipGetLastError());
    wgrad_wino_finish_kernel<<<ceil_div(d->Cout * d->Cin, 256), 256, 0, stream>>>(msum, dw, d->Cin, d->Cout, p.cigs);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

}  // namespace kmsr
