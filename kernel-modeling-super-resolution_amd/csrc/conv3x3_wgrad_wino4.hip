// Weight gradient of the 3x3 convolution in the Winograd F(4x4,3x3) domain on the fp32 matrix cores
// (semantics: include/kmsr.h, kmsr_conv3x3_wgrad_f32; this file serves W % 4 == 0 with Cout > 32).
//
//   forward per 4x4 output tile:  Y = A^T [ (G g G^T) .* (B^T d B) ] A          (A 6x4, G 6x3, B 6x6; kmsr_wino4_xf.h)
//   =>  dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G      36 products per (co, ci, 4x4 tile): 144 direct, 64 for F(2x2,3x3)
//
// The sum over tiles (and images) is the K dimension of 36 independent [Cout x Cin] GEMMs, one per transform point xi = (i, j):
//   M_xi[co][ci] += Ahat_xi[co][tile] * Vhat_xi[ci][tile]     on v_mfma_f32_16x16x4_f32  (M = 16 co, N = 16 ci, K = 4 tiles).
// As in the forward kernel a lane computes BOTH operands it feeds from raw data in LDS -- Vhat = B^T d B of its own (ci, tile)
// 6x6 patch and Ahat = A dY A^T of its own (co, tile) 4x4 gradient block, on packed-fp32 register pairs -- so transformed data
// never travels through LDS.
//
// Register budget decides the shape.  36 x 64 x 64 accumulators are 576 KB -- more than a CU's register file -- so a workgroup owns
// 64 co x 32 ci (288 KB of accumulators) and its FOUR waves (one 512-register wave per SIMD) split that slab as
//   (transform-point half ps: rows i = 3 ps .. 3 ps + 2 of the 6x6 point grid) x (co half coh):  18 points x 2 co blocks x 2 ci blocks
//   = 72 accumulator tiles = 288 registers (256 AGPRs + 8 tiles in ArchVGPRs through the VGPR form of the instruction).
// Splitting the POINTS by rows keeps the transforms proportional: a wave computes only "its" three rows of B^T d (18 packed VALU)
// and their row transforms (3 x 7), and only its three rows of A dY (8) and their row transforms (3 x 4): per 4 tiles
// 2 x 39 + 2 x 20 = 118 packed VALU against 72 MFMAs (the F(2x2) kernel: 56 per 32; per pixel both counts fall ~1.8x).
//
// Data path: a stage = one row of 8 tiles (4 gradient rows x 32 columns).  Its raw x rows (32 ci x 6 rows x 40 columns) and dy rows
// (64 co x 4 rows x 32 columns) arrive by BUFFER LDS-DMA (16 bytes per lane, 64 instructions per stage = 16 per wave, issued
// between the MFMAs of the previous stage) into a double-buffered 2 x 65 KB LDS image.  One instruction = one x plane (6 rows x 10
// quads) or one pair of dy planes (2 x 4 rows x 8 quads), placed 65 quads apart: the pitch is odd, so the 16 channels of an operand
// read hit 16 different bank quads.  The lane -> (row, quad) map is the same for every instruction of a tensor: the stage origin
// sits in the descriptor base, the plane in the instruction's scalar offset, and ONE voffset register per tensor carries the lane
// part -- out of range (the hardware then writes zeros) where the stage's halo leaves the image; channel tails and requests past
// the end of the stream are record counts of 0.  No masks in the transforms, ~15 VALU of addressing per stage.  One barrier per stage.
// Workgroups walk contiguous ranges of the column-major stage order (down an image column): the two halo rows a stage shares with
// the previous one are L2 hits, and the two ci-half workgroups of a range run on the same XCD (dy is fetched from HBM once).
// Per-workgroup slabs are summed in a fixed order (bit-reproducible, no float atomics), then transformed to [Cout,Cin,3,3].
#include "kmsr_common.h"
#include "kmsr_wino4_xf.h"

#include <algorithm>

namespace kmsr {
namespace {

using namespace wino4xf;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int HT = 8;                       // tiles per stage
constexpr int HXQ = HT + 2;                 // x quads per row: columns x0 - 4 .. x0 + 35
constexpr int HXROW = 4 * HXQ;              // 40 floats
constexpr int HPL = 4 * 65;                 // floats between the LDS images of two DMA instructions: 64 quads + 1 (odd pitch)
constexpr int HCI = 32, HCO = 64;           // channels per workgroup
constexpr int HXINSTR = HCI;                // one DMA instruction per x plane (60 live lanes: 6 rows x 10 quads)
constexpr int HXF = HXINSTR * HPL;          // 8320 floats
constexpr int HDROW = 4 * HT;               // 32 floats
constexpr int HDCH = 4 * HDROW;             // 128 floats: one dy plane (4 rows x 32 columns)
constexpr int HDINSTR = HCO / 2;            // one DMA instruction per PAIR of dy planes (lanes 0-31 / 32-63)
constexpr int HINSTR = HXINSTR + HDINSTR;   // 64 per stage
constexpr int HWI = HINSTR / 4;             // 16 per wave: instruction I = 4 i + wv, x planes for i < 8, dy pairs for i >= 8
constexpr int HSTAGE = HINSTR * HPL;        // 16640 floats = 65 KB
constexpr int HPW = 18;                     // transform points per wave
constexpr int HSLAB = 36 * HCO * HCI;       // floats per workgroup slab
constexpr int HFLUSH = 128;                 // stages (1024 tiles) between two flushes of the accumulators into the workgroup's slab
constexpr int HVT = 4;                      // points (of 18) whose 4 tiles accumulate in ArchVGPRs (64 registers; 56 tiles = 224 AGPRs)
static_assert(HXINSTR % 4 == 0 && HINSTR % 4 == 0, "DMA instruction split");

struct Wg4Params {
    int N, Cin, H, W, Cout;
    int cblocks, trows, n_stages;           // 32-column blocks per row, tile rows per image, stages per image
    int cogs, cigs, nblk;
};

constexpr int h_wait_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }
#define H_SB() __builtin_amdgcn_sched_barrier(0)

// ---- the three rows i = 3 PS + {0,1,2} of B^T d for a pair of columns (6 packed VALU); d[] = raw rows PS .. PS + 4
template <int PS>
__device__ __forceinline__ void bt_rows3(const f32x2 (&d)[5], f32x2& o0, f32x2& o1, f32x2& o2) {
    if constexpr (PS == 0) {                // rows 0, 1, 2 from d0 .. d4
        const f32x2 a = fma2(splat2(-4.f), d[2], d[4]), b = fma2(splat2(-4.f), d[1], d[3]);
        o0 = fma2(splat2(4.f), d[0], fma2(splat2(-5.f), d[2], d[4]));
        o1 = a + b;
        o2 = a - b;
    } else {                                // rows 3, 4, 5 from d1 .. d5 (d[k] = raw row k + 1)
        const f32x2 c = d[3] - d[1], e = d[2] - d[0];
        o0 = fma2(splat2(2.f), e, c);
        o1 = fma2(splat2(-2.f), e, c);
        o2 = fma2(splat2(4.f), d[0], fma2(splat2(-5.f), d[2], d[4]));
    }
}
// ---- the three rows i = 3 PS + {0,1,2} of A dY for a pair of columns (4 packed VALU; one row is a copy); g[] = gradient rows 0 .. 3
//      A = [[1,0,0,0],[1,1,1,1],[1,-1,1,-1],[1,2,4,8],[1,-2,4,-8],[0,0,0,1]]
template <int PS>
__device__ __forceinline__ void a_rows3(const f32x2 (&g)[4], f32x2& o0, f32x2& o1, f32x2& o2) {
    if constexpr (PS == 0) {
        const f32x2 e = g[0] + g[2], o = g[1] + g[3];
        o0 = g[0];
        o1 = e + o;
        o2 = e - o;
    } else {
        const f32x2 e2 = fma2(splat2(4.f), g[2], g[0]), o4 = fma2(splat2(4.f), g[3], g[1]);
        o0 = fma2(splat2(2.f), o4, e2);
        o1 = fma2(splat2(-2.f), o4, e2);
        o2 = g[3];
    }
}
// Where the next stage's 16 DMA instructions of a wave sit: (K-step, group g = 2 a + b, MFMA pair q of 9) -> instruction index or -1.
// One request per ~9 MFMAs over the first six groups of the stage: every request that misses L2 stalls the issuing wave ~140 cycles
// (measured: 66 ns per instruction with the HBM stream, 8 ns with L2-resident data -- the per-CU miss path), and back to back they
// queue behind each other; the last one still leaves two groups (> 0.5 us) before the barrier that publishes the stage.
__device__ constexpr int dma_slot(int ks, int g, int q) {
    const int grp = ks * 4 + g;                      // 0 .. 7
    if (grp >= 6) return -1;
    const int base = grp < 4 ? 3 * grp : 12 + 2 * (grp - 4);      // groups 0-3: 3 each (12), groups 4-5: 2 each (4)
    if (grp < 4) return q == 1 ? base : q == 4 ? base + 1 : q == 7 ? base + 2 : -1;
    return q == 2 ? base : q == 6 ? base + 1 : -1;
}

template <int PS>
__device__ __forceinline__ void wg4_body(float* __restrict__ lds, const float* __restrict__ xg, const float* __restrict__ dyg,
                                         float* __restrict__ partial, const Wg4Params& p, int wv) {
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, tl = lane >> 4;
    const int coh = wv >> 1;
    // ---- block -> (channel group grp, strip stream xr).  The hardware deals workgroups to the 8 XCDs round-robin (id % 8); with
    //      nblk a multiple of 8 the two (or more) channel groups of one stream sit next to each other in dispatch order on ONE XCD
    //      (they read the same dy rows: one HBM fetch, one L2 hit), and the streams of an XCD are consecutive (vertically adjacent
    //      strips at any moment: the two halo rows they share are L2 hits as well).
    const int ngrp = p.cogs * p.cigs;
    int grp, xr;
    if ((p.nblk & 7) == 0) {
        const int id = blockIdx.x, j = id >> 3;
        grp = j % ngrp;
        xr = (id & 7) * (p.nblk >> 3) + j / ngrp;
    } else {
        grp = blockIdx.x % ngrp;
        xr = blockIdx.x / ngrp;
    }
    const int cog = grp / p.cigs, cig = grp % p.cigs;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    const unsigned HWb = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;

    // ---- this workgroup's stages: the strips (image n, tile row tr) u = xr, xr + nblk, ... each walked left to right in 32-column
    //      blocks.  Walking along the rows makes the column halo (8 of 40 columns) a re-read of lines this CU fetched one stage
    //      earlier, and at any moment the nblk streams work on nblk consecutive strips.  (First version: contiguous ranges of the
    //      column-major order -- every 16-byte halo quad cost its own HBM sector and nothing was shared: 2.4 GB fetched for 1.07.)
    const int n_strips = p.N * p.trows;
    const int my_strips = xr < n_strips ? (n_strips - 1 - xr) / p.nblk + 1 : 0;
    const int s_beg = 0, s_end = my_strips * p.cblocks;

    // ---- DMA maps (fixed per lane; ONE voffset per tensor for all 16 instructions of a wave):
    //      x  instruction c      : plane c of the group, lane -> (row = lane / 10, q = lane % 10), lanes 60 .. 63 dead
    //      dy instruction k      : planes 2k, 2k + 1,    lane -> (plane = lane >> 5, row = (lane >> 3) & 3, q = lane & 7)
    //      The stage origin ((y0 - 1, x0 - 4) of x, (y0, x0) of dy) lives in the descriptor BASE, the plane in the scalar offset (x) or
    //      in the base as well (dy); a lane is switched off (zeros) by an out-of-range voffset (0x80000000: beyond any record count
    //      with or without the scalar offset added).
    const unsigned voX0 = lane < 6 * HXQ ? (unsigned)((lane / HXQ) * W + 4 * (lane % HXQ)) * 4u : OOB;
    const unsigned voD0 = (unsigned)(lane >> 5) * HWb + (unsigned)(((lane >> 3) & 3) * W + 4 * (lane & 7)) * 4u;
    const int ci_n = min(p.Cin - cig * HCI, HCI), co_n = min(p.Cout - cog * HCO, HCO);

    // ---- stage cursor of the DMA stream (the stage being REQUESTED)
    int rs = s_beg, ru = xr, rcb = 0;
    int rn = ru / p.trows, rtr = ru - rn * p.trows;
    const float* bX = xg;                    // descriptor bases of the stage being requested (may point in front of the tensor:
    const float* bD = dyg;                   // such lanes are switched off)
    unsigned voX = OOB, voD = OOB;
    auto dma_setup = [&]() {                 // stage (rn, rcb, rtr)
        const int y0 = 4 * rtr, x0 = 4 * HT * rcb;
        bX = xg + ((long long)rn * p.Cin + cig * HCI) * HW + ((y0 - 1) * W + x0 - 4);
        bD = dyg + ((long long)rn * p.Cout + cog * HCO) * HW + (y0 * W + x0);
        int ln = lane;
        asm volatile("" : "+v"(ln));         // re-derived from the lane id here: nothing lane-shaped but voX / voD stays live across a stage
        const int xr = ln / HXQ, xqq = ln - xr * HXQ;
        const int gy = y0 - 1 + xr, gx = x0 - 4 + 4 * xqq;
        voX = (gy >= 0 && gy < H && gx >= 0 && gx + 3 < W) ? voX0 : OOB;
        const int dy_ = y0 + ((ln >> 3) & 3), dx_ = x0 + 4 * (ln & 7);
        voD = (dy_ < H && dx_ + 3 < W) ? voD0 : OOB;
    };
    // Channel tails are CLAMPED, not zeroed: instruction c >= ci_n (co_n) re-reads the group's last real plane.  Those planes only feed
    // rows / columns of M that belong to padding channels, which the final transform never reads.  (A record count per instruction --
    // 16 compare masks -- cost 170 spilled scalars.)  The same holds past the end of the stream: the last stage re-requests its own
    // rows into the idle buffer.  Only the odd tail of a dy pair needs the range check: its second plane may lie beyond the tensor.
    auto dma_issue = [&](int i, float* buf) {                // instruction I = 4 i + wv
        lds_void_t* dst = (lds_void_t*)(buf + (4 * i + wv) * HPL);
        if (i < HXINSTR / 4) {
            const int c = min(4 * i + wv, ci_n - 1);
            const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bX), 0, 0x7FFFFFF0u, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voX, (unsigned)c * HWb, 0, 0);
        } else {
            const int c = min(2 * (4 * (i - HXINSTR / 4) + wv), co_n - 1);
            // the pair's first plane sits in the BASE (not in the scalar offset): measured on gfx950, the scalar offset IS part of the
            // range check (plane in soffset + num_records = one plane returned zeros for the whole instruction)
            const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bD + (long long)c * HW), 0, (unsigned)(co_n - c) * HWb, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voD, 0, 0, 0);
        }
    };
    auto dma_advance = [&]() {
        ++rs;
        if (++rcb == p.cblocks) {
            rcb = 0;
            ru += p.nblk;
            rn = ru / p.trows;
            rtr = ru - rn * p.trows;
        }
    };

    // ---- accumulators: [co block a][ci block b][point]; D lane holds rows co = 4 tl + r, column ci = c16

    // ---- operand reads (floats inside a stage image): x patch rows PS .. PS + 4, columns 4 tile + 3 .. + 8; dy rows 0 .. 3
    //      x: channel 16 b + c16 of the group;  dy: channel 32 coh + 2 c16 + a (block a = the parity: the two planes of a DMA pair sit
    //      32 quads apart, so a block's 16 lanes take planes of 16 DIFFERENT pairs -> 16 different bank quads)
    const int xbase = c16 * HPL + PS * HXROW + 4 * tl + 3;
    const int dbase = HXF + (16 * coh + c16) * HPL + 4 * tl;
    struct XRaw { f32x4 mid[5]; f32x2 e[5]; };
    struct DRaw { f32x4 g[4]; };
    // ONE LDS instruction each: k = 2 r (ds_read_b128: columns 1-4 of patch row r) / 2 r + 1 (ds_read2_b32: columns 0 and 5); dy row r
    auto rx1 = [&](XRaw& P, const float* st, int b, int ks, int k) {
        const float* rp = st + xbase + b * 16 * HPL + 16 * ks + (k >> 1) * HXROW;
        if (k & 1) P.e[k >> 1] = f32x2{rp[0], rp[5]};
        else P.mid[k >> 1] = *reinterpret_cast<const f32x4*>(rp + 1);
    };
    auto rd1 = [&](DRaw& D, const float* st, int a, int ks, int r) {
        D.g[r] = *reinterpret_cast<const f32x4*>(st + dbase + a * HDCH + 16 * ks + r * HDROW);
    };
    f32x2 K41 = f32x2{4.f, 1.f}, K54 = f32x2{-5.f, 4.f}, K15 = f32x2{1.f, -5.f};
    asm volatile("" : "+v"(K41), "+v"(K54), "+v"(K15));
    // Vhat rows i = 3 PS .. 3 PS + 2 of one (ci block, 4 tiles): column passes on the pairs (0,5), (1,2), (3,4) (3 x 6 packed VALU), then the
    // line pass of every row (3 x 7)
    auto vhat = [&](const XRaw& P, float (&v)[HPW]) {
        f32x2 p0[3], p1[3], p2[3];                          // per row i: (t0,t5), (t1,t2), (t3,t4)
        {
            const f32x2 d[5] = {P.e[0], P.e[1], P.e[2], P.e[3], P.e[4]};
            bt_rows3<PS>(d, p0[0], p0[1], p0[2]);
        }
        {
            const f32x2 d[5] = {f32x2{P.mid[0].x, P.mid[0].y}, f32x2{P.mid[1].x, P.mid[1].y}, f32x2{P.mid[2].x, P.mid[2].y},
                                f32x2{P.mid[3].x, P.mid[3].y}, f32x2{P.mid[4].x, P.mid[4].y}};
            bt_rows3<PS>(d, p1[0], p1[1], p1[2]);
        }
        {
            const f32x2 d[5] = {f32x2{P.mid[0].z, P.mid[0].w}, f32x2{P.mid[1].z, P.mid[1].w}, f32x2{P.mid[2].z, P.mid[2].w},
                                f32x2{P.mid[3].z, P.mid[3].w}, f32x2{P.mid[4].z, P.mid[4].w}};
            bt_rows3<PS>(d, p2[0], p2[1], p2[2]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f32x2 O05, O12, O34;
            bt6row(p0[i], p1[i], p2[i], K41, K54, K15, O05, O12, O34);
            v[6 * i + 0] = O05.x; v[6 * i + 5] = O05.y;
            v[6 * i + 1] = O12.x; v[6 * i + 2] = O12.y;
            v[6 * i + 3] = O34.x; v[6 * i + 4] = O34.y;
        }
#pragma unroll
        for (int k = 0; k < HPW; ++k) asm volatile("" : "+v"(v[k]));
    };
    // Ahat rows of one (co block, 4 tiles): 2 x 4 + 3 x 4 = 20 packed VALU
    auto ahat = [&](const DRaw& D, float (&o)[HPW]) {
        f32x2 lo[3], hi[3];                                 // per row i: (t0,t1), (t2,t3)
        {
            const f32x2 g[4] = {f32x2{D.g[0].x, D.g[0].y}, f32x2{D.g[1].x, D.g[1].y}, f32x2{D.g[2].x, D.g[2].y}, f32x2{D.g[3].x, D.g[3].y}};
            a_rows3<PS>(g, lo[0], lo[1], lo[2]);
        }
        {
            const f32x2 g[4] = {f32x2{D.g[0].z, D.g[0].w}, f32x2{D.g[1].z, D.g[1].w}, f32x2{D.g[2].z, D.g[2].w}, f32x2{D.g[3].z, D.g[3].w}};
            a_rows3<PS>(g, hi[0], hi[1], hi[2]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x2 O12 = pk_lo_pm_hi(lo[i] + hi[i]);                     // (e + o, e - o), e = t0 + t2, o = t1 + t3
            const f32x2 O34 = pk_lo_pm_2hi(fma2(splat2(4.f), hi[i], lo[i]));  // (e2 + 2 o4, e2 - 2 o4), e2 = t0 + 4 t2, o4 = t1 + 4 t3
            o[6 * i + 0] = lo[i].x; o[6 * i + 1] = O12.x; o[6 * i + 2] = O12.y; o[6 * i + 3] = O34.x; o[6 * i + 4] = O34.y; o[6 * i + 5] = hi[i].y;
        }
#pragma unroll
        for (int k = 0; k < HPW; ++k) asm volatile("" : "+v"(o[k]));
    };

    // 288 accumulator registers exceed the 256 AGPRs (see conv3x3_wino4.hip): the tiles of the first HVT points use the VGPR form
    // of the instruction, the other 56 the AGPR form -- BOTH through inline asm with the register class in the constraint: left to
    // the builtin, hipcc moved accumulator tiles between the files around every product (1400 v_accvgpr_* in the first build).
    // No hazard needs padding by hand: a tile is written once per 72 MFMAs, operands come from VALU results (interlocked), and the
    // accumulators are only read (by VALU) behind the stream's last barrier.
#define H_MFMA(A_, B_, K_)                                                                                                    \
    {                                                                                                                         \
        if constexpr ((K_) < HVT)                                                                                             \
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[A_][B_][K_]) : "v"(Aop[A_][K_]), "v"(Vop[B_][K_])); \
        else                                                                                                                  \
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[A_][B_][K_]) : "v"(Aop[A_][K_]), "v"(Vop[B_][K_])); \
    }

    // ---- slab of this workgroup in the RAW register image: [wave 4][tile (a*2+b)*18 + k][lane 64][r 4] -- one 16-byte store per lane
    //      and tile (1 KB per instruction).  The accumulators are flushed into it every HFLUSH stages (read-modify-write by the owning
    //      wave: fixed order, bit-reproducible): an fp32 accumulator that runs over n products carries ~eps * sqrt(n) of ITS
    //      magnitude, and the transform-domain sums of the outer points (+-2, +-2) are ~40x the weight gradient they contribute to
    //      (measured vs float64 at batch 32: 9e-6 of max|dw| after 1024 tiles per accumulator, 2.1e-5 after 4096).
    const auto rslab = __builtin_amdgcn_make_buffer_rsrc(partial + ((long long)grp * p.nblk + xr) * HSLAB + (long long)wv * (72 * 256), 0,
                                                         72 * 1024, 0x00020000);
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    auto flush = [&](f32x4 (&acc)[2][2][HPW], bool first) {
        // three phases of 24 tiles: the old values of a phase are requested together (96 registers in flight), then added and stored
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            f32x4 old[24];
            if (!first) {                                    // block-uniform
#pragma unroll
                for (int t = 0; t < 24; ++t)      // one lane offset + a per-tile scalar offset (72 hoisted 64-bit pointers were spilled)
                    old[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rslab, lane * 16, (24 * ph + t) * 1024, 0));
            } else {
#pragma unroll
                for (int t = 0; t < 24; ++t) old[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 24; ++t) {
                const int T = 24 * ph + t;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[T / 36][(T / 18) & 1][T % 18] + old[t]), rslab, lane * 16, T * 1024, 0);
            }
            asm volatile("" ::: "memory");
        }
    };
    bool first_flush = true;

    // ---- prologue: stage s_beg -> buffer 0
    int b = 0;
    if (s_beg < s_end) {
        dma_setup();
#pragma unroll
        for (int i = 0; i < HWI; ++i) dma_issue(i, lds);
        dma_advance();
    }
    __builtin_amdgcn_s_waitcnt(h_wait_imm(0, 15));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // chunks of HFLUSH stages: the accumulators live (in their fixed register classes) from the zeroing at the top of a chunk to the
    // flush at its end -- a flush INSIDE the stage loop made hipcc spill 670 registers around the loop-carried tiles
    for (int s0 = s_beg; s0 < s_end; s0 += HFLUSH) {
    const int s1 = min(s0 + HFLUSH, s_end);
    f32x4 acc[2][2][HPW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bq = 0; bq < 2; ++bq)
#pragma unroll
            for (int k = 0; k < HPW; ++k) acc[a][bq][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; ++s) {
        const bool req_on = rs < s_end;                      // block-uniform: a next stage exists -> request it into the other buffer
        const float* st = lds + b * HSTAGE;
        float* nb = lds + (b ^ 1) * HSTAGE;
        if (req_on) dma_setup();
        XRaw X0, X1;
        DRaw D0, D1;
#pragma unroll
        for (int k = 0; k < 10; ++k) rx1(X0, st, 0, 0, k);   // the stage's first operands: the only LDS reads nothing covers
#pragma unroll
        for (int r = 0; r < 4; ++r) rd1(D0, st, 0, 0, r);
#pragma unroll
        for (int ks = 0; ks < HT / 4; ++ks) {
            // One K-step (4 tiles): 118 packed VALU + 72 MFMAs in four (co block, ci block) groups of 18 points (9 MFMA pairs each).  A
            // group's operands are transformed right before it; everything else sits BETWEEN its pairs one or two instructions at a
            // time: the LDS reads of the raw data needed one group (or more) later, and in K-step 0 the next stage's 16 DMA requests.
            // (Four waves posting their 10 .. 28 reads in a burst keep the LDS pipe -- and in order behind it every wave's next MFMAs --
            // busy for ~1200 cycles per K-step: cycle stamps, 3000 of 9900 cycles per stage.)
            constexpr bool kFalse = false;
            const bool first = ks == 0, nxt = ks + 1 < HT / 4;
            float Aop[2][HPW], Vop[2][HPW];
            H_SB();
            vhat(X0, Vop[0]);
            ahat(D0, Aop[0]);
            H_SB();
#define H_G(A_, B_, Q_, K0, K1, EXTRA)                                        \
    H_MFMA(A_, B_, K0) H_MFMA(A_, B_, K1)                                     \
    {                                                                         \
        const int di = dma_slot(ks, 2 * (A_) + (B_), Q_);                     \
        if (di >= 0) dma_issue(di, nb);                                       \
    }                                                                         \
    EXTRA H_SB();
#define H_X1(Q_) if (first) { rx1(X1, st, 1, ks, 2 * (Q_)); rx1(X1, st, 1, ks, 2 * (Q_) + 1); }
#define H_D1(Q_) rd1(D1, st, 1, ks, Q_);
#define H_X0N(Q_) if (nxt) rx1(X0, st, 0, ks + 1, Q_);
#define H_D0N(Q_) if (nxt) rd1(D0, st, 0, ks + 1, Q_);
#define H_X1N(Q_) if (nxt) rx1(X1, st, 1, ks + 1, Q_);
            // group (0,0): K-step 0 requests X1 here (its Vhat follows this group), later K-steps got it during the previous (1,1)
            H_G(0, 0, 0, 4, 5, H_X1(0)) H_G(0, 0, 1, 6, 7, H_X1(1)) H_G(0, 0, 2, 8, 9, H_X1(2)) H_G(0, 0, 3, 10, 11, H_X1(3)) H_G(0, 0, 4, 0, 1, H_X1(4))
            H_G(0, 0, 5, 2, 3, H_D1(0)) H_G(0, 0, 6, 12, 13, H_D1(1)) H_G(0, 0, 7, 14, 15, H_D1(2)) H_G(0, 0, 8, 16, 17, H_D1(3))
            vhat(X1, Vop[1]);
            H_SB();
            H_G(0, 1, 0, 4, 5, H_X0N(0)) H_G(0, 1, 1, 6, 7, H_X0N(1)) H_G(0, 1, 2, 8, 9, H_X0N(2)) H_G(0, 1, 3, 10, 11, H_X0N(3))
            H_G(0, 1, 4, 0, 1, H_X0N(4)) H_G(0, 1, 5, 2, 3, H_X0N(5)) H_G(0, 1, 6, 12, 13, H_X0N(6)) H_G(0, 1, 7, 14, 15, H_X0N(7))
            H_G(0, 1, 8, 16, 17, H_X0N(8) H_X0N(9))
            ahat(D1, Aop[1]);
            H_SB();
            H_G(1, 0, 0, 4, 5, H_D0N(0)) H_G(1, 0, 1, 6, 7, H_D0N(1)) H_G(1, 0, 2, 8, 9, H_D0N(2)) H_G(1, 0, 3, 10, 11, H_D0N(3))
            H_G(1, 0, 4, 0, 1, ) H_G(1, 0, 5, 2, 3, ) H_G(1, 0, 6, 12, 13, ) H_G(1, 0, 7, 14, 15, ) H_G(1, 0, 8, 16, 17, )
            H_G(1, 1, 0, 4, 5, H_X1N(0) H_X1N(1)) H_G(1, 1, 1, 6, 7, H_X1N(2) H_X1N(3)) H_G(1, 1, 2, 8, 9, H_X1N(4) H_X1N(5)) H_G(1, 1, 3, 10, 11, H_X1N(6) H_X1N(7))
            H_G(1, 1, 4, 0, 1, H_X1N(8) H_X1N(9)) H_G(1, 1, 5, 2, 3, ) H_G(1, 1, 6, 12, 13, ) H_G(1, 1, 7, 14, 15, ) H_G(1, 1, 8, 16, 17, )
            (void)kFalse;
#undef H_G
#undef H_X1
#undef H_D1
#undef H_X0N
#undef H_D0N
#undef H_X1N
        }
        if (req_on) dma_advance();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(h_wait_imm(0, 0));       // next stage landed (requested a whole stage ago); my reads of this one are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        b ^= 1;
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // the last MFMAs' results (inline asm: invisible to the hazard recognizer)
    flush(acc, first_flush);
    first_flush = false;
    }
    if (s_beg >= s_end) {                                    // an idle workgroup still owns a slab the reduction reads
#pragma unroll 4
        for (int t = 0; t < 72; ++t) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rslab, lane * 16, t * 1024, 0);
    }
#undef H_MFMA

}

__global__ void __launch_bounds__(256, 1) conv3x3_wgrad_wino4_kernel(const float* __restrict__ xg, const float* __restrict__ dyg,
                                                                     float* __restrict__ partial, Wg4Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 2 stage images
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv & 1) wg4_body<1>(lds, xg, dyg, partial, p, wv);               // wave-uniform: the point halves run different transform code
    else wg4_body<0>(lds, xg, dyg, partial, p, wv);
}
#undef H_SB

// msum[grp][e] = sum_blk slab[grp][blk][e]   (fixed order, 8 interleaved running sums)
__global__ void __launch_bounds__(256) wgrad_wino4_sum_kernel(const float* __restrict__ partial, float* __restrict__ msum, int nblk) {
    const int e = blockIdx.x * 256 + threadIdx.x;            // < HSLAB
    const int grp = blockIdx.y;
    const float* src = partial + (long long)grp * nblk * HSLAB + e;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] += src[(long long)(b + q) * HSLAB];
    }
    for (int q = 0; b < nblk; ++b, ++q) a8[q] += src[(long long)b * HSLAB];
    msum[(long long)grp * HSLAB + e] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

// dw[co][ci][u][v] = sum_{i,j} G[i][u] M[i][j] G[j][v]
__global__ void __launch_bounds__(256) wgrad_wino4_finish_kernel(const float* __restrict__ msum, float* __restrict__ dw, int Cin,
                                                                 int Cout, int cigs) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // over Cout * Cin, ci fastest
    if (idx >= Cout * Cin) return;
    const int ci = idx % Cin, co = idx / Cin;
    // raw register image (see the kernel's flush): co -> (half coh, block a = parity, MFMA row), ci -> (block b, column)
    const int col = co % HCO, cil = ci % HCI;
    const int coh = col >> 5, a = col & 1, mrow = (col & 31) >> 1, bb = cil >> 4, lane = (mrow >> 2) * 16 + (cil & 15);
    const float* m = msum + (long long)((co / HCO) * cigs + ci / HCI) * HSLAB + (long long)((a * 2 + bb) * HPW * 64 + lane) * 4 + (mrow & 3);
    auto at = [&](int i, int j) {            // point (i, j): wave 2 coh + (i >= 3), tile index + (6 (i % 3) + j)
        return m[(long long)(((2 * coh + (i >= 3 ? 1 : 0)) * 72 + 6 * (i % 3) + j) * 64) * 4];
    };
    // T = G^T M (3 x 6), row by row of M
    float T[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float M[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) M[i] = at(i, j);
        const float s12 = M[1] + M[2], d12 = M[2] - M[1], s34 = M[3] + M[4], d34 = M[3] - M[4];
        T[0][j] = 0.25f * M[0] - (1.f / 6) * s12 + (1.f / 24) * s34;
        T[1][j] = (1.f / 6) * d12 + (1.f / 12) * d34;
        T[2][j] = -(1.f / 6) * s12 + (1.f / 6) * s34 + M[5];
    }
    float* o = dw + ((long long)co * Cin + ci) * 9;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const float s12 = T[u][1] + T[u][2], d12 = T[u][2] - T[u][1], s34 = T[u][3] + T[u][4], d34 = T[u][3] - T[u][4];
        o[u * 3 + 0] = 0.25f * T[u][0] - (1.f / 6) * s12 + (1.f / 24) * s34;
        o[u * 3 + 1] = (1.f / 6) * d12 + (1.f / 12) * d34;
        o[u * 3 + 2] = -(1.f / 6) * s12 + (1.f / 6) * s34 + T[u][5];
    }
}

int cu_count4() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

void wgrad_wino4_plan(const kmsr_conv3x3_desc* d, Wg4Params* p) {
    p->N = d->N; p->Cin = d->Cin; p->H = d->H; p->W = d->W; p->Cout = d->Cout;
    p->cblocks = ceil_div(d->W, 4 * HT); p->trows = ceil_div(d->H, 4); p->n_stages = p->cblocks * p->trows;
    p->cogs = ceil_div(d->Cout, HCO); p->cigs = ceil_div(d->Cin, HCI);
    const long long stages = (long long)d->N * p->n_stages;
    int nb = (int)std::max<long long>(1, std::min<long long>(stages, cu_count4() / (p->cogs * p->cigs)));   // one workgroup per CU
    if (nb > 8) nb &= ~7;       // a multiple of 8: block (x, group) and (x, group') land on the same XCD (linear id mod 8)
    p->nblk = nb;
}

}  // namespace

// ---- used by kmsr_conv3x3_wgrad_f32 (conv3x3_wgrad.hip)
bool wgrad_wino4_eligible(const kmsr_conv3x3_desc* d) {
    return d->W % 4 == 0 && d->Cout > 32 && d->H >= 4 && !(d->flags & (KMSR_CONV_FLAG_WGRAD_DIRECT | KMSR_CONV_FLAG_WGRAD_WINO2));
}

size_t wgrad_wino4_workspace_bytes(const kmsr_conv3x3_desc* d) {
    Wg4Params p{};
    wgrad_wino4_plan(d, &p);
    return (size_t)p.cogs * p.cigs * ((size_t)p.nblk + 1) * HSLAB * sizeof(float);
}

int wgrad_wino4_launch(const kmsr_conv3x3_desc* d, const float* x, const float* dy, float* dw, float* ws, hipStream_t stream) {
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(dy) % 16)
        return fail(KMSR_E_BADARG, "wgrad (winograd 4x4): x / dy must be 16-byte aligned");
    Wg4Params p{};
    wgrad_wino4_plan(d, &p);
    const int groups = p.cogs * p.cigs;
    float* msum = ws + (size_t)groups * p.nblk * HSLAB;
    const int shmem = 2 * HSTAGE * (int)sizeof(float);
    KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_wino4_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
    conv3x3_wgrad_wino4_kernel<<<p.nblk * groups, 256, shmem, stream>>>(x, dy, ws, p);
    KMSR_HIP_CHECK(hipGetLastError());
    wgrad_wino4_sum_kernel<<<dim3(HSLAB / 256, groups), 256, 0, stream>>>(ws, msum, p.nblk);
    KMSR_HIP_CHECK(hipGetLastError());
    wgrad_wino4_finish_kernel<<<ceil_div(d->Cout * d->Cin, 256), 256, 0, stream>>>(msum, dw, d->Cin, d->Cout, p.cigs);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

}  // namespace kmsr
