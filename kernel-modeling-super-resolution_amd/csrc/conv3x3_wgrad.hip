// Weight gradient of the 3x3 convolution on the fp32 matrix cores (semantics: include/kmsr.h).
//
//   per tap t=(u,v):  dW_t[co, ci] += A[co, px] * B_t[px, ci],   A = dy,  B_t[px, ci] = x[ci, px shifted by (u-1, v-1)]
//   one v_mfma_f32_32x32x2_f32 per 32 co x 32 ci x 2 pixels.  Lane l holds A[co = l&31][px = l>>5] and
//   B[px = l>>5][ci = l&31], so both operands want channel-contiguous LDS images: tiles are staged TRANSPOSED
//   ([pixel][channel], pitch 65 floats so that the transposing ds_writes are bank-conflict free).
//
// Persistent grid: every workgroup walks a strided list of pixel stages (2 rows x 32 cols of one image) and keeps a
// whole 64 co x 64 ci x 9 tap slab in registers: wave w owns (co block w&1, ci block w>>1) = 9 accumulator tiles
// (144 VGPRs).  The slab is written once to the workspace; reduce_partials_kernel sums the per-workgroup slabs in a
// fixed order (deterministic, no float atomics).
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int kWgradSmallCo = 7;     // Cout <= 7: conv3x3_wgrad_smallco_kernel
constexpr int WR = 2, WC = 32;            // pixel stage: rows x cols
constexpr int WPX = WR * WC;              // 64 pixels per stage
constexpr int XR = WR + 2, XC = WC + 2;   // halo tile of x
constexpr int XPOS = XR * XC;             // 136 positions
constexpr int PCH = 65;                   // LDS pitch (floats) of one [pixel][64 channels] row
constexpr int DYF = WPX * PCH;            // 4160
constexpr int XF = XPOS * PCH;            // 8840
constexpr int STAGE_F = DYF + XF;         // 13000 floats = 52 KB
constexpr int NDY = WPX * 64 / 256;       // 16 loads / thread

struct WgradParams {
    int N, Cin, H, W, Cout;
    int tiles_x, tiles_y, n_stages;       // stages per (image): tiles_y * tiles_x ; total = N * that
    int cogs, cigs;                       // 64-channel groups
    int nblk;                             // persistent workgroups per (co group, ci group)
};

template <bool SMALL_M>     // SMALL_M: Cout <= 32 (one co block), the co-block wave pair splits the pixels instead
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_kernel(const float* __restrict__ xg, const float* __restrict__ dyg,
                                                               float* __restrict__ partial, WgradParams p) {
    __shared__ __attribute__((aligned(16))) float lds[STAGE_F];   // ONE stage (52 KB): two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wv & 1, ib = wv >> 1;
    const int grp = blockIdx.y, cog = grp / p.cigs, cig = grp % p.cigs;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    const int total = p.N * p.n_stages;

    // ---- staging maps, computed once per thread (pixel / halo position fixed, channels stride per load):
    //   dy : thread = (px = tid & 63, co_sub = tid >> 6)          -> 16 loads, co = 4q + co_sub
    //   x  : thread = (pos = tid & 127, ci_sub = tid >> 7)        -> 32 loads, ci = 2q + ci_sub   (positions 0..127)
    //        + 2 loads for the last 8 halo positions 128..135:       e = tid + 256 j -> pos = 128 + (e & 7), ci = e >> 3
    // dcs / xcs are wave-uniform (tid >> 6 is the wave id): told to the compiler so the channel part of every address is scalar
    const int dpx = tid & 63, dcs = __builtin_amdgcn_readfirstlane(tid >> 6), dr = dpx / WC, dc = dpx % WC;
    const int xpos = tid & 127, xcs = __builtin_amdgcn_readfirstlane(tid >> 7), xrr = xpos / XC, xcc = xpos % XC;
    int tpos[2], tci[2], trr[2], tcc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + 256 * j;
        tpos[j] = 128 + (e & 7); tci[j] = e >> 3; trr[j] = tpos[j] / XC; tcc[j] = tpos[j] % XC;
    }
    float dyr[NDY], xr[32], xt[2];
    // Staging loads are buffer loads: one descriptor per (image, 64-channel group) whose record count ends at the group's last
    // real channel, a per-thread byte offset of the pixel (or an out-of-range sentinel for halo positions outside the image)
    // plus a wave-uniform channel stride.  The range check returns 0 for padding pixels AND for channels past Cin / Cout, so
    // a load costs one v_add instead of a predicated 64-bit address (this kernel's matrix pipe shares its issue slots with VALU).
    const unsigned HWb = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;
    auto load_stage = [&](int s) {
        const int n = s / p.n_stages, t = s % p.n_stages;
        const int y0 = (t / p.tiles_x) * WR, x0 = (t % p.tiles_x) * WC;
        const int co_n = min(p.Cout - cog * 64, 64), ci_n = min(p.Cin - cig * 64, 64);
        const auto rdy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(dyg + ((long long)n * p.Cout + cog * 64) * HW), 0, (unsigned)co_n * HWb, 0x00020000);
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xg + ((long long)n * p.Cin + cig * 64) * HW), 0, (unsigned)ci_n * HWb, 0x00020000);
        {
            const int y = y0 + dr, x = x0 + dc;
            const unsigned voff = (y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB;
#pragma unroll
            for (int q = 0; q < NDY; ++q)
                dyr[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, voff + (unsigned)(4 * q + dcs) * HWb, 0, 0));
        }
        {
            const int y = y0 - 1 + xrr, x = x0 - 1 + xcc;
            const unsigned voff = (y >= 0 && y < H && x >= 0 && x < W) ? (unsigned)(y * W + x) * 4u : OOB;
#pragma unroll
            for (int q = 0; q < 32; ++q)
                xr[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff + (unsigned)(2 * q + xcs) * HWb, 0, 0));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int yy = y0 - 1 + trr[j], xx = x0 - 1 + tcc[j];
                const unsigned vo = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)(yy * W + xx) * 4u : OOB;
                xt[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo + (unsigned)tci[j] * HWb, 0, 0));
            }
        }
    };
    auto store_stage = [&]() {
        float* dl = lds;
        float* xl = dl + DYF;
#pragma unroll
        for (int q = 0; q < NDY; ++q) dl[dpx * PCH + 4 * q + dcs] = dyr[q];      // [px][co]
#pragma unroll
        for (int q = 0; q < 32; ++q) xl[xpos * PCH + 2 * q + xcs] = xr[q];       // [pos][ci]
#pragma unroll
        for (int j = 0; j < 2; ++j) xl[tpos[j] * PCH + tci[j]] = xt[j];
    };

    // <= 32 output channels in this group (the SR tail: 3..7): the second 32-row co block would multiply zeros, so the two
    // waves of a (co block) pair instead split the stage's pixel pairs (even / odd) over the SAME 32 channels; wave cb = 1
    // still writes its slab into rows 32..63 and reduce_partials_kernel adds the two halves.
    constexpr bool small_m = SMALL_M;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // A lone wave cannot keep a SIMD's fp32 matrix pipe full (measured: ~65 % busy with 9 independent accumulators), so the
    // kernel runs TWO workgroups per CU (one wave of each per SIMD).  That leaves one 52 KB stage buffer per workgroup: the
    // next stage is fetched into registers during the products and written after a barrier; the other workgroup's waves own
    // the matrix pipes meanwhile.
    int s = blockIdx.x;
    if (s < total) {
        load_stage(s);
        store_stage();
    }
    __syncthreads();
    for (; s < total; s += p.nblk) {
        const bool more = s + p.nblk < total;
        if (more) load_stage(s + p.nblk);                                       // in flight under the 288 MFMAs below
        const float* dl = lds;
        const float* ap = dl + half * PCH + (small_m ? 0 : cb * 32) + l32;      // + px*PCH
        const float* bp = dl + DYF + half * PCH + ib * 32 + l32;                // + pos*PCH
#pragma unroll
        for (int r = 0; r < WR; ++r) {
#pragma unroll 4
            for (int c = 0; c < WC / 2; ++c) {
                if constexpr (SMALL_M) {
                    if ((c & 1) != cb) continue;                                // <= 32 output channels: the co-block pair splits the pixels
                }
                const float a = ap[(r * WC + 2 * c) * PCH];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float bv = bp[((r + t / 3) * XC + 2 * c + t % 3) * PCH];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                        // every wave is done reading the stage
        if (more) store_stage();
        __syncthreads();
    }
    // partial[(grp*nblk + blockIdx.x)][tap][co 64][ci 64]
    float* out = partial + ((long long)grp * p.nblk + blockIdx.x) * (9 * 64 * 64);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[(t * 64 + co) * 64 + ib * 32 + l32] = acc[t][r];
        }
}

// dw[co][ci][tap] = sum_blk partial[grp][blk][tap][co%64][ci%64]   (fixed order: deterministic)
__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cin, int Cout,
                                       int cigs, int nblk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over Cout*Cin*9, ci fastest for coalesced partial reads
    if (idx >= Cout * Cin * 9) return;
    const int ci = idx % Cin, tap = (idx / Cin) % 9, co = idx / (Cin * 9);
    const int grp = (co / 64) * cigs + ci / 64;
    const float* src = partial + (long long)grp * nblk * (9 * 64 * 64) + (tap * 64 + co % 64) * 64 + ci % 64;
    const bool small_m = Cout <= 32;                         // pixel-split slabs: rows co and co + 32 hold the two halves
    // fixed summation order (8 interleaved running sums, combined pairwise): bit-reproducible, and the 8 loads of a round are
    // independent so the loop does not serialise on memory latency
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] += src[(long long)(b + q) * (9 * 64 * 64)];
    }
    for (int q = 0; b < nblk; ++b, ++q) a8[q] += src[(long long)b * (9 * 64 * 64)];
    float s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    if (small_m) {
        const float* src2 = src + 32 * 64;
        float c8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b2 = 0;
        for (; b2 + 8 <= nblk; b2 += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) c8[q] += src2[(long long)(b2 + q) * (9 * 64 * 64)];
        }
        for (int q = 0; b2 < nblk; ++b2, ++q) c8[q] += src2[(long long)b2 * (9 * 64 * 64)];
        s += ((c8[0] + c8[1]) + (c8[2] + c8[3])) + ((c8[4] + c8[5]) + (c8[6] + c8[7]));
    }
    dw[((long long)co * Cin + ci) * 9 + tap] = s;
}

// ------------------------------------------------------------------------------------------------
// Few output channels (Cout <= 7: the SR tail F -> bands).  The 64 x 64 x 9 slab above would be 90 % zero padding, so the
// GEMM is turned around: the (co, tap) pairs -- at most 63 -- become the N dimension and the SHIFT moves from x to dy:
//   dW[(co,u,v), ci] = sum_q x[ci, q] * dy[co, q - (u-1, v-1)]        (q runs over the x pixels; dy is 0 outside the image)
//   A = x  (lane l: A[m = ci = l&31 (+32 mb)][k = q pair half]),  B = shifted dy (lane l: B[k][n = (co,tap) = l&31 (+32 nb)])
// Workgroup = 4 waves = (ci block mb, n block nb): ONE accumulator tile per wave, one MFMA per wave and pixel pair; the x
// stage (64 pixels x 64 channels, transposed, no halo) streams through LDS double buffered -- the kernel is bound by reading
// x once (HBM), not by the matrix pipe.  Slab per workgroup: [n 64][ci 64]; reduced in a fixed order like the big kernel.
// ------------------------------------------------------------------------------------------------
constexpr int SD_ROWS = WR + 2, SD_COLS = WC + 2, SD_POS = SD_ROWS * SD_COLS;   // dy halo tile 4 x 34
constexpr int SD_P = 9;                                                          // pitch of one dy position (co <= 7), odd
constexpr int SX_F = WPX * PCH;                                                  // x stage [px][ci], pitch 65
constexpr int SS_F = SX_F + SD_POS * SD_P;

// N48 (round 4, Cout <= 5: at most 45 (co, tap) columns): v_mfma_f32_16x16x4_f32 tiles instead -- 4 ci blocks x 3 column blocks of 16, wave = one ci
// block with its three accumulator tiles, K = 4 consecutive pixels of a row: 48 MFMAs of 32 cycles per stage and wave where the 32 x 32 tiling
// (2 x 2 blocks, the second column block 13/32 used) spends 32 of 64 cycles; the kernel is bound by those cycles, not by reading x.
template <bool N48>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_smallco_kernel(const float* __restrict__ xg, const float* __restrict__ dyg,
                                                                       float* __restrict__ partial, WgradParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SS_F];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wv & 1, nb = wv >> 1;
    const int cig = blockIdx.y;
    const int H = p.H, W = p.W, Cout = p.Cout;
    const long long HW = (long long)H * W;
    const int total = p.N * p.n_stages;
    const int ci_left = p.Cin - cig * 64;

    // staging maps (buffer loads: the range check supplies the zero padding and the channel tail):
    //   x : 16-byte pieces when W % 4 == 0: e = tid + 256 j, j < 4 -> piece e & 7 (columns 4 piece .. +3), row (e >> 3) & 1, channel e >> 4;
    //       dword loads otherwise: thread = (px = tid & 63, sub = wave id) -> channels 4 q + sub, q < 16
    //   dy halo: element e = tid + 256 j -> (pos = e / Cout, co = e % Cout), e < 136 * Cout (<= 952 -> 4 loads)
    const int dpx = tid & 63, sub = wv, dr = dpx / WC, dc = dpx % WC;
    const int xq = tid & 7, xrow = (tid >> 3) & 1, xch = tid >> 4;
    constexpr int NDYL = (SD_POS * 7 + 255) / 256;
    const unsigned HWb = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;
    const bool wide = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(xg) & 15) == 0;       // block-uniform
    // the dy halo map of this thread is the same for every stage: (row, column) relative to the stage origin, channel byte offset, LDS slot
    // (the divisions by the run-time Cout cost ~160 VALU per stage next to its 48 MFMAs when they sat inside the stage loop)
    int dmap_y[NDYL], dmap_x[NDYL], dmap_l[NDYL];
    unsigned dmap_c[NDYL];
#pragma unroll
    for (int j = 0; j < NDYL; ++j) {
        const int e = tid + 256 * j;
        const int pos = e / Cout, co = e - pos * Cout;
        dmap_y[j] = pos / SD_COLS - 1;
        dmap_x[j] = pos % SD_COLS - 1;
        dmap_c[j] = (unsigned)co * HWb;
        dmap_l[j] = e < SD_POS * Cout ? pos * SD_P + co : -1;
    }
    float xrA[16], dyrA[NDYL];
    auto load_stage_to = [&](int s, float (&xr)[16], float (&dyr)[NDYL]) {
        const int n = s / p.n_stages, t = s % p.n_stages;
        const int y0 = (t / p.tiles_x) * WR, x0 = (t % p.tiles_x) * WC;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + ((long long)n * p.Cin + cig * 64) * HW), 0,
                                                          (unsigned)min(ci_left, 64) * HWb, 0x00020000);
        const auto rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyg + (long long)n * Cout * HW), 0, (unsigned)Cout * HWb, 0x00020000);
        if (wide) {
            const int y = y0 + xrow, x = x0 + 4 * xq;
            const unsigned vo = ((y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB) + (unsigned)xch * HWb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + (unsigned)(16 * j) * HWb, 0, 0));
                xr[4 * j + 0] = v.x; xr[4 * j + 1] = v.y; xr[4 * j + 2] = v.z; xr[4 * j + 3] = v.w;
            }
        } else {
            const int y = y0 + dr, x = x0 + dc;
            const unsigned vo = (y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                xr[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo + (unsigned)(4 * q + sub) * HWb, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < NDYL; ++j) {
            const int y = y0 + dmap_y[j], x = x0 + dmap_x[j];
            const bool ok = dmap_l[j] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            dyr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, ok ? (unsigned)(y * W + x) * 4u + dmap_c[j] : OOB, 0, 0));
        }
    };
    auto store_stage_from = [&](int b, const float (&xr)[16], const float (&dyr)[NDYL]) {
        float* xl = lds + b * SS_F;
        float* dl = xl + SX_F;
        if (wide) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) xl[(xrow * WC + 4 * xq + k) * PCH + xch + 16 * j] = xr[4 * j + k];     // [px][ci]: banks (px + ci) mod 64, all distinct
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) xl[dpx * PCH + 4 * q + sub] = xr[q];    // [px][ci]
        }
#pragma unroll
        for (int j = 0; j < NDYL; ++j)
            if (dmap_l[j] >= 0) dl[dmap_l[j]] = dyr[j];                           // [pos][co]
    };

    auto load_stage = [&](int s) { load_stage_to(s, xrA, dyrA); };
    auto store_stage = [&](int b) { store_stage_from(b, xrA, dyrA); };
    if constexpr (N48) {
        using f32x4a = __attribute__((ext_vector_type(4))) float;
        const int l16 = lane & 15, kq = lane >> 4;
        int boffs3[3];
#pragma unroll
        for (int nbk = 0; nbk < 3; ++nbk) {
            const int ncol = 16 * nbk + l16;
            const bool nok = ncol < Cout * 9;
            const int nco = nok ? ncol / 9 : 0, ntap = nok ? ncol % 9 : 0;
            boffs3[nbk] = ((2 - ntap / 3) * SD_COLS + (2 - ntap % 3)) * SD_P + nco + kq * SD_P;   // + (r * SD_COLS + 4 c4) * SD_P; k = pixel of the quad
        }
        f32x4a acc3[3];
#pragma unroll
        for (int nbk = 0; nbk < 3; ++nbk) acc3[nbk] = f32x4a{0.f, 0.f, 0.f, 0.f};
        // two register staging sets: stage s + 2 nblk is requested while stage s is multiplied and written to LDS during stage s + nblk -- one
        // stage of MFMAs (48 x 32 cycles = 0.7 us) does not cover an HBM round trip
        float xrB[16], dyrB[NDYL];
        auto products = [&](int b) {
            const float* xl = lds + b * SS_F;
            const float* ap = xl + kq * PCH + 16 * wv + l16;                      // A[m = ci of block wv][k = pixel of the quad]; + px * PCH
            const float* bq = xl + SX_F;
#pragma unroll
            for (int r = 0; r < WR; ++r) {
#pragma unroll
                for (int c4 = 0; c4 < WC / 4; ++c4) {
                    const float a = ap[(r * WC + 4 * c4) * PCH];
#pragma unroll
                    for (int nbk = 0; nbk < 3; ++nbk) {
                        // columns past Cout * 9 read column 0's values: a column of D depends on its own B column only, and the reduction
                        // never reads those columns -- no select on the port the MFMA shares
                        const float bv = bq[boffs3[nbk] + (r * SD_COLS + 4 * c4) * SD_P];
                        acc3[nbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc3[nbk], 0, 0, 0);
                    }
                }
            }
        };
        int s = blockIdx.x;
        const int step = p.nblk;
        if (s < total) {
            load_stage_to(s, xrA, dyrA);
            if (s + step < total) load_stage_to(s + step, xrB, dyrB);
            store_stage_from(0, xrA, dyrA);
        }
        __syncthreads();
        for (; s < total; s += 2 * step) {
            {   // even stage of the pair: LDS buffer 0; request s + 2 step into set A, write s + step from set B
                if (s + 2 * step < total) load_stage_to(s + 2 * step, xrA, dyrA);
                products(0);
                if (s + step < total) store_stage_from(1, xrB, dyrB);
                __syncthreads();
            }
            if (s + step < total) {
                if (s + 3 * step < total) load_stage_to(s + 3 * step, xrB, dyrB);
                products(1);
                if (s + 2 * step < total) store_stage_from(0, xrA, dyrA);
                __syncthreads();
            }
        }
        // D lane l: column n = l & 15 of block nbk, rows ci = 16 wv + 4 (l >> 4) + reg
        float* out = partial + ((long long)cig * p.nblk + blockIdx.x) * (64 * 64);
#pragma unroll
        for (int nbk = 0; nbk < 3; ++nbk)
            *reinterpret_cast<f32x4a*>(out + (16 * nbk + l16) * 64 + 16 * wv + 4 * kq) = acc3[nbk];
        return;
    }
    // this lane's B column: n = 32 nb + l32 = co * 9 + tap; the dy position of x pixel (r, c) is (r + 2 - u, c + 2 - v) in the halo tile
    const int ncol = 32 * nb + l32;
    const bool nok = ncol < Cout * 9;
    const int nco = nok ? ncol / 9 : 0, ntap = nok ? ncol % 9 : 0;
    const int boffs = ((2 - ntap / 3) * SD_COLS + (2 - ntap % 3)) * SD_P + nco;   // + (r * SD_COLS + c) * SD_P

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    int s = blockIdx.x;
    if (s < total) {
        load_stage(s);
        store_stage(0);
    }
    __syncthreads();
    int b = 0;
    for (; s < total; s += p.nblk) {
        const bool more = s + p.nblk < total;
        if (more) load_stage(s + p.nblk);
        const float* xl = lds + b * SS_F;
        const float* ap = xl + half * PCH + mb * 32 + l32;                       // + px * PCH
        const float* bp = xl + SX_F + boffs + half * SD_P;                       // + (r * SD_COLS + c) * SD_P
#pragma unroll
        for (int r = 0; r < WR; ++r) {
#pragma unroll
            for (int c = 0; c < WC / 2; ++c) {
                const float a = ap[(r * WC + 2 * c) * PCH];
                const float bv = nok ? bp[(r * SD_COLS + 2 * c) * SD_P] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
            }
        }
        if (more) store_stage(b ^ 1);
        __syncthreads();
        b ^= 1;
    }
    // partial[(cig * nblk + blockIdx.x)][n 64][ci 64]; D lane l holds column n = l&31 and rows ci = (reg&3) + 8 (reg>>2) + 4 (l>>5)
    float* out = partial + ((long long)cig * p.nblk + blockIdx.x) * (64 * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[(32 * nb + l32) * 64 + ci] = acc[r];
    }
}

// dw[co][ci][tap] = sum_blk partial[cig][blk][co*9+tap][ci%64]   (fixed order, 8 interleaved running sums)
__global__ void reduce_partials_smallco_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cin, int Cout,
                                               int nblk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over Cout*9*Cin, ci fastest
    if (idx >= Cout * 9 * Cin) return;
    const int ci = idx % Cin, n = idx / Cin, co = n / 9, tap = n % 9;
    const float* src = partial + (long long)(ci / 64) * nblk * (64 * 64) + n * 64 + ci % 64;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] += src[(long long)(b + q) * (64 * 64)];
    }
    for (int q = 0; b < nblk; ++b, ++q) a8[q] += src[(long long)b * (64 * 64)];
    dw[((long long)co * Cin + ci) * 9 + tap] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}


int wgrad_blocks(const kmsr_conv3x3_desc* d) {
    const long long stages = (long long)d->N * ceil_div(d->H, WR) * ceil_div(d->W, WC);
    // two persistent workgroups per CU (52 KB of LDS each, 2 waves per SIMD); the small-Cout kernel (43 KB, 104 VGPRs) runs three
    return (int)std::min<long long>(stages, d->Cout <= kWgradSmallCo ? 768 : 512);
}

}  // namespace

// Winograd-domain path (conv3x3_wgrad_wino.hip): even H, W and Cout > 32
bool wgrad_wino_eligible(const kmsr_conv3x3_desc* d);
size_t wgrad_wino_workspace_bytes(const kmsr_conv3x3_desc* d);
int wgrad_wino_launch(const kmsr_conv3x3_desc* d, const float* x, const float* dy, float* dw, float* ws, hipStream_t stream);
// Winograd F(4x4,3x3)-domain path (conv3x3_wgrad_wino4.hip): W % 4 == 0, H >= 4 and Cout > 32 -- tried first
bool wgrad_wino4_eligible(const kmsr_conv3x3_desc* d);
size_t wgrad_wino4_workspace_bytes(const kmsr_conv3x3_desc* d);
int wgrad_wino4_launch(const kmsr_conv3x3_desc* d, const float* x, const float* dy, float* dw, float* ws, hipStream_t stream);
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv3x3_wgrad_workspace_bytes(const kmsr_conv3x3_desc* d) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return 0;
    if (wgrad_wino4_eligible(d)) return wgrad_wino4_workspace_bytes(d);
    if (wgrad_wino_eligible(d)) return wgrad_wino_workspace_bytes(d);
    return (size_t)ceil_div(d->Cout, 64) * ceil_div(d->Cin, 64) * wgrad_blocks(d) * 9 * 64 * 64 * sizeof(float);
}

extern "C" int kmsr_conv3x3_wgrad_f32(const kmsr_conv3x3_desc* d, const float* x, const float* dy, float* dw,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "wgrad: stride %d not implemented", d->stride);
    if (!x || !dy || !dw) return fail(KMSR_E_BADARG, "null tensor pointer");
    const size_t need = kmsr_conv3x3_wgrad_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) return fail(KMSR_E_NOMEM, "wgrad workspace %zu B < required %zu B", workspace_bytes, need);
    if ((long long)std::max(d->Cin, d->Cout) * d->H * d->W >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "per-image tensor exceeds 2^31 elements");
    if ((long long)d->H * d->W * 4 * 64 >= (1ll << 31))       // buffer-load byte offsets of one 64-channel group are 32-bit
        return fail(KMSR_E_UNSUPPORTED, "wgrad: H*W = %lld exceeds 2^23 pixels per plane", (long long)d->H * d->W);
    if (d->flags & ~(uint32_t)(KMSR_CONV_FLAG_WGRAD_DIRECT | KMSR_CONV_FLAG_WGRAD_WINO2)) return fail(KMSR_E_BADARG, "wgrad: unknown flags 0x%x", d->flags);
    if (wgrad_wino4_eligible(d))
        return wgrad_wino4_launch(d, x, dy, dw, static_cast<float*>(workspace), static_cast<hipStream_t>(stream_));
    if (wgrad_wino_eligible(d))
        return wgrad_wino_launch(d, x, dy, dw, static_cast<float*>(workspace), static_cast<hipStream_t>(stream_));
    WgradParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, WC); p.tiles_y = ceil_div(d->H, WR); p.n_stages = p.tiles_x * p.tiles_y;
    p.cogs = ceil_div(d->Cout, 64); p.cigs = ceil_div(d->Cin, 64);
    p.nblk = wgrad_blocks(d);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->Cout <= kWgradSmallCo) {
        if (d->Cout * 9 <= 48)
            conv3x3_wgrad_smallco_kernel<true><<<dim3(p.nblk, p.cigs), 256, 0, stream>>>(x, dy, static_cast<float*>(workspace), p);
        else
            conv3x3_wgrad_smallco_kernel<false><<<dim3(p.nblk, p.cigs), 256, 0, stream>>>(x, dy, static_cast<float*>(workspace), p);
        KMSR_HIP_CHECK(hipGetLastError());
        const int tot = d->Cout * d->Cin * 9;
        reduce_partials_smallco_kernel<<<ceil_div(tot, 256), 256, 0, stream>>>(static_cast<const float*>(workspace), dw, d->Cin,
                                                                               d->Cout, p.nblk);
        KMSR_HIP_CHECK(hipGetLastError());
        return KMSR_OK;
    }
    if (d->Cout <= 32)
        conv3x3_wgrad_kernel<true><<<dim3(p.nblk, p.cogs * p.cigs), 256, 0, stream>>>(x, dy, static_cast<float*>(workspace), p);
    else
        conv3x3_wgrad_kernel<false><<<dim3(p.nblk, p.cogs * p.cigs), 256, 0, stream>>>(x, dy, static_cast<float*>(workspace), p);
    KMSR_HIP_CHECK(hipGetLastError());
    const int total = d->Cout * d->Cin * 9;
    reduce_partials_kernel<<<ceil_div(total, 256), 256, 0, stream>>>(static_cast<const float*>(workspace), dw, d->Cin,
                                                                     d->Cout, p.cigs, p.nblk);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
