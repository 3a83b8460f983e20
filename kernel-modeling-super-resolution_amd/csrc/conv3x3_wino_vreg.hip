// Winograd F(2x2,3x3) forward convolution on the fp32 matrix cores, "V in registers" form (semantics: include/kmsr.h;
// same math, epilogue and tolerances as conv3x3_wino.hip / conv3x3_wino_dma.hip).
//
// What the LDS-DMA kernel (conv3x3_wino_dma.hip) leaves on the table (profiles/r01_wino_dma_experiments.log, matrix pipe busy
// 51 %): every 8-channel chunk pays two workgroup barriers around an LDS round trip of the transformed inputs V, every tile
// pays an exposed prologue (cold DMA) and an epilogue in which the SIMD partners exchange half of the output transform
// through LDS.  Two intermediate designs of this round are logged in profiles/r02_winov_v1_stamps.log (xi-row waves with an
// LDS exchange epilogue; 4-wave workgroups two per CU -- DMA-latency-bound).  This kernel:
//   * MFMA tile v_mfma_f32_16x16x4_f32 (16 channels x 16 tiles x 4 input channels).  A wave owns 32 output channels x 16 tiles
//     for ALL 16 transform points: 2 x 16 accumulator tiles x 4 registers = 128 registers.  Lane (tile t = lane & 15,
//     channel kq = lane >> 4) supplies the B operand V[xi][4 s + kq][t] for every xi -- exactly the 4x4 input transform of ITS
//     patch, which it computes itself from the raw rows in LDS (12 ds_read_b64 + 32 VALU per 32 MFMAs, placed by hand between
//     the MFMAs of the previous step).  V never goes through LDS: no V buffer, no transform barrier;
//   * all 16 transform points of a (channel, tile) pair sit in one lane, so the output transform A^T M A is lane-local:
//     no exchange through LDS, no epilogue barrier -- the epilogue is ~200 VALU and the stores;
//   * the workgroup (8 waves = 2 channel halves x 4 tile quarters = 64 channels x 64 tiles, one per CU) is PERSISTENT: it walks
//     an interleaved list of (image, cout group, tile) items and treats their chunks as ONE stream.  U (pre-transformed weights,
//     32 KB per 8-channel chunk, shared by the 8 waves) and the raw input rows (13.5 KB per chunk) arrive by LDS-DMA in rings
//     of 3 and 4 slots; the DMA for chunk G+2 / G+4 is issued during chunk G whatever item it belongs to, so only the first
//     item of a workgroup sees a cold prologue;
//   * ONE barrier per chunk, placed mid-chunk where the MFMA operands of the following step are already in registers; every wave
//     issues the same 6 DMA instructions per chunk (padding lanes, out-of-image pieces, channel tails and the slots past the end
//     of the stream fetch a 16-byte zero page), so the counted `s_waitcnt vmcnt(2)` before the barrier is exact;
//   * the A operands are packed so that a lane's four consecutive transform points are one ds_read_b128, lane-linear (no bank
//     conflicts), requested one 8-MFMA group ahead.
// LDS: U ring 3 x 32 KB + raw ring 4 x 14 KB + 1 KB dump = 153 KB.  Requires W % 4 == 0 and 16-byte aligned x / packed_w.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int VCK = 8;                 // input channels per chunk (2 MFMA steps of 4)
constexpr int VCO = 64;                // output channels per workgroup
constexpr int TROWS = 2, TCOLS = 32;   // tiles per workgroup: 2 x 32  (output 4 rows x 64 cols)
constexpr int kThreads = 512;
constexpr int VUF = 16 * VCK * VCO;    // 8192 floats = 32 KB of U per chunk: [step 2][cb16 4][m 4][lane 64][jj 4]
constexpr int NU = 3;                  // U ring slots (chunk G+2 is requested during chunk G)
constexpr int UINSTR = VUF / 4 / kThreads;         // 4 U DMA instructions per wave and chunk (every wave issues its share: concentrating
                                                   // the DMA on one wave per SIMD was slower, 439 vs 394 us)
constexpr int RROWS = 2 * TROWS + 2;   // 6 input rows
constexpr int RQ = (2 * TCOLS + 8) / 4;            // 18 float4 per row segment: columns tx0-4 .. tx0+67
constexpr int RCOLS = 4 * RQ;                      // 72
constexpr int RLANES = VCK * RROWS * RQ;           // 864 lane-loads per chunk = 13.5 wave instructions
constexpr int RINSTR = 2;                          // raw DMA instructions per wave and chunk (16 slots, 14 real, 2 to the dump)
constexpr int RREAL = (RLANES + 63) / 64;          // 14
constexpr int RAWF = RREAL * 256;                  // 3584 floats = 14 KB per ring slot
constexpr int NRAW = 4;                            // raw ring slots (chunk G+4 is requested during chunk G)
constexpr int MAX_BIAS = 1024;                     // the whole bias vector sits in LDS (read in the epilogue without touching vmcnt)
constexpr int OFF_U = 0, OFF_R = NU * VUF, OFF_DUMP = OFF_R + NRAW * RAWF, OFF_B = OFF_DUMP + 256, LDS_FLOATS = OFF_B + MAX_BIAS;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget of one CU");

__device__ __attribute__((aligned(16))) float g_zero_page[4];   // zero-initialised: source of every padded 16-byte piece

struct WinoVParams {
    int N, Cin, H, W, Cout;            // stride 1: Ho = H, Wo = W
    int tiles_x, tiles, cgs, nchunks, relu, n_items;
    int res_mask;                      // the skip tensor is a ReLU mask source: out = res > 0 ? conv : 0 (instead of conv + res)
};

// packed[cg][chunk][step][cb16][m][lane][jj] = (G g G^T)[xi = 4m + jj] for
// (co, ci) = (cg*64 + cb16*16 + (lane & 15), chunk*8 + step*4 + (lane >> 4)), zero padded
__global__ void winov_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ packed, int nchunks,
                                  long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const int m = (int)((idx >> 8) & 3);
        const int cb16 = (int)((idx >> 10) & 3);
        const int step = (int)((idx >> 12) & 1);
        const int chunk = (int)((idx >> 13) % nchunks);
        const int cg = (int)((idx >> 13) / nchunks);
        const int cog = cg * VCO + cb16 * 16 + (lane & 15), cig = chunk * VCK + step * 4 + (lane >> 4);
        float v = 0.f;
        if (cog < Cout && cig < Cin) {
            const float* g = w + ((long long)cog * Cin + cig) * 9;
            // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  U = G g G^T
            float row[3];   // (G g)[m][:]
            for (int c = 0; c < 3; ++c) {
                const float g0 = g[0 * 3 + c], g1 = g[1 * 3 + c], g2 = g[2 * 3 + c];
                row[c] = m == 0 ? g0 : (m == 1 ? 0.5f * (g0 + g1 + g2) : (m == 2 ? 0.5f * (g0 - g1 + g2) : g2));
            }
            v = jj == 0 ? row[0] : (jj == 1 ? 0.5f * (row[0] + row[1] + row[2]) : (jj == 2 ? 0.5f * (row[0] - row[1] + row[2]) : row[2]));
        }
        packed[idx] = v;
    }
}

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]).  The builtin, not inline asm:
// hipcc then KNOWS the counter state and does not re-wait conservatively for LDS reads that were already retired.
constexpr int wait_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }
#define KMSR_FENCE() asm volatile("" ::: "memory")
#define KMSR_SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void barrier_lds() {   // LDS hand-over only: does NOT wait for LDS-DMA in flight (counted by the caller)
    KMSR_FENCE();
    __builtin_amdgcn_s_waitcnt(wait_imm(63, 0));
    __builtin_amdgcn_s_barrier();
    KMSR_FENCE();
}
template <int N>
__device__ __forceinline__ void wait_dma() {      // all but the N newest vector-memory operations of this wave have completed
    KMSR_FENCE();
    __builtin_amdgcn_s_waitcnt(wait_imm(N, 15));
    KMSR_FENCE();
}

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
struct PatchPair { f32x2 p[4]; };      // one aligned column pair of the 4 patch rows (4 ds_read_b64)

template <bool PS, bool TRAIN>     // TRAIN: the training-step epilogue extras (ReLU-mask skip tensor, per-channel sums); never with PS
__global__ void __launch_bounds__(kThreads, 2) conv3x3_winov_kernel(const float* __restrict__ xg,
                                                                    const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ res,
                                                                    float* __restrict__ out, float* __restrict__ csum,
                                                                    WinoVParams p) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = wv & 1, tq = wv >> 1, tb = tq >> 1, tc0 = 16 * (tq & 1);   // channel half, tile quarter = (tile row, first tile column)
    const int H = p.H, W = p.W, nch = p.nchunks;
    const long long HW = (long long)H * W;

    // Items of this workgroup: vb, vb + nwg, vb + 2 nwg, ...  (vb = XCD-grouped id: the workgroups of one XCD hold consecutive vb).
    // All workgroups walk their streams in step, so at any moment the chip works on ~nwg CONSECUTIVE items: horizontally and
    // vertically adjacent tiles (shared halo rows / columns) and the common U chunk are touched at the same time by CUs of one
    // XCD and are served from its L2 (contiguous per-workgroup ranges re-read them 4 items apart: 2.7x the algorithmic HBM bytes).
    const int nwg = gridDim.x, bq = nwg >> 3, br = nwg & 7, xcd = blockIdx.x & 7;
    const int vb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int it0 = vb, istep = nwg;
    if (it0 >= p.n_items) return;                            // block-uniform
    const int T = ((p.n_items - it0 + istep - 1) / istep) * nch;   // chunks in this workgroup's stream

    const float* zero_page = g_zero_page;
    asm volatile("" : "+s"(zero_page));                      // resolved once (the GOT load would otherwise be repeated per chunk)

    // ---- DMA cursors.  U: chunk gU of the stream = chunk cU of item itU (only its cout group matters).
    int gU = 0, cU = 0, itU = it0, cgU = (it0 / p.tiles) % p.cgs;
    const int uoff = (UINSTR * wv * 64 + lane) * 4;          // this lane's float offset inside a chunk of U (instruction i: + 256 i)
    auto issue_u = [&](int i0, int i1) {                     // DMA instructions [i0, i1) of chunk gU; the cursor advances after the last one
        float* ub = lds + OFF_U + (gU % NU) * VUF;
        // wave-uniform base + constant lane offset (global_load_lds saddr form: no per-instruction address VALU).  Past the end of
        // the stream the cursor stays on the last chunk: the slot is free, the re-fetched data is never read.
        const float* __restrict__ ubase = wpk + ((long long)cgU * nch + cU) * VUF;
#pragma unroll
        for (int i = i0; i < i1; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(ubase + uoff + i * 256), (lds_void_t*)(ub + (UINSTR * wv + i) * 256), 16, 0, 0);
        if (i1 == UINSTR) {
            ++gU;
            if (gU < T && ++cU == nch) {
                cU = 0;
                itU += istep;
                cgU = (itU / p.tiles) % p.cgs;
            }
        }
    };
    // raw rows: chunk gR = chunk cR of item itR.  Raw tile, lane-linear: L = (2 wv + i) * 64 + lane <-> (segment = L / 18 =
    // channel * 6 + row, q = L % 18 = float4 within the row); LDS offset = 4 L floats; instruction slots 14, 15 go to the dump.
    int gR = 0, cR = 0, itR = it0;
    int roff[RINSTR];                                        // per-lane float offset of its 16-byte piece inside an 8-channel chunk, -1: padding
    const float* xnR = xg;
    auto raw_map = [&]() {                                   // lane map of item itR (when the cursor enters a new item)
        const int tile = itR % p.tiles, n = itR / (p.tiles * p.cgs);
        const int ty0 = (tile / p.tiles_x) * (2 * TROWS), tx0 = (tile % p.tiles_x) * (2 * TCOLS);
        xnR = xg + (long long)n * p.Cin * HW;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                     // recompute HERE (hoisted out of the stream loop the lane constants would be spilled)
#pragma unroll
        for (int i = 0; i < RINSTR; ++i) {
            const int L = (RINSTR * wv + i) * 64 + lane_o;
            const int seg = L / RQ, q = L - seg * RQ, ch = seg / RROWS, row = seg - ch * RROWS;
            const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * q;
            const bool ok = L < RLANES && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
            roff[i] = ok ? (int)(ch * HW + (long long)gy * W + gx) : -1;
        }
    };
    raw_map();
    auto issue_raw = [&]() {
        const float* __restrict__ xc = xnR + (long long)cR * VCK * HW;
        const int cin_left = gR < T ? p.Cin - cR * VCK : 0;  // past the end of the stream: zeros
        float* rb = lds + OFF_R + (gR % NRAW) * RAWF;
        const bool tail = cin_left < VCK;                    // block-uniform: only a channel-tail chunk (or the end of the stream) needs the channel test
#pragma unroll
        for (int i = 0; i < RINSTR; ++i) {
            bool ok = roff[i] >= 0;
            if (tail) ok = ok && ((RINSTR * wv + i) * 64 + lane) / (RQ * RROWS) < cin_left;
            const float* src = ok ? xc + roff[i] : zero_page;
            const int slot = RINSTR * wv + i;                // wave-uniform
            float* dst = slot < RREAL ? rb + slot * 256 : lds + OFF_DUMP;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
        }
        ++gR;
        if (++cR == nch) {
            cR = 0;
            itR += istep;
            if (gR < T) raw_map();
        }
    };

    // ---- input transform: lane (tile t16 of the wave's 16, channel kq) reads its 4x4 patch from the raw tile: patch rows r <->
    //      raw rows 2 tb + r; patch columns <-> raw columns 2 tc + 3 .. + 6 (tc = tc0 + t16), read as the aligned pairs at raw columns
    //      2 tc + 2, + 4, + 6.  V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].
    const int pbase = (kq * RROWS + 2 * tb) * RCOLS + 2 * (tc0 + t16) + 2;          // + step * 4 * RROWS * RCOLS
    auto load_pair = [&](int g, int step, int k) {           // column pair k (raw columns 2 tc + 2 + 2k, +1) of the 4 patch rows
        const float* rp = lds + OFF_R + (g % NRAW) * RAWF + pbase + step * (4 * RROWS * RCOLS) + 2 * k;
        PatchPair q;
#pragma unroll
        for (int r = 0; r < 4; ++r) q.p[r] = *reinterpret_cast<const f32x2*>(rp + r * RCOLS);
        return q;
    };
    // row transform of one patch column c: t[i][c] = (B^T d)[i][c] from the column's four values d0 .. d3 (4 VALU)
    auto rows_col = [&](float (&t)[4][4], int c, float d0, float d1, float d2, float d3) {
        t[0][c] = d0 - d2;
        t[1][c] = d1 + d2;
        t[2][c] = d2 - d1;
        t[3][c] = d1 - d3;
        asm volatile("" : "+v"(t[0][c]), "+v"(t[1][c]), "+v"(t[2][c]), "+v"(t[3][c]));   // keep the VALU work where it is written (LLVM would sink it)
    };
    auto cols_of = [&](const float (&t)[4][4], float (&v)[16]) {               // v[4i + j] = (t B)[i][j]  (16 VALU)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * i + 0] = t[i][0] - t[i][2];
            v[4 * i + 1] = t[i][1] + t[i][2];
            v[4 * i + 2] = t[i][2] - t[i][1];
            v[4 * i + 3] = t[i][1] - t[i][3];
            asm volatile("" : "+v"(v[4 * i + 0]), "+v"(v[4 * i + 1]), "+v"(v[4 * i + 2]), "+v"(v[4 * i + 3]));
        }
    };

    f32x4 acc[2][16];      // [channel block of 16][xi]: rows = channels 4 kq + reg, column = tile t16
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[a][x] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A operands of this wave: U ring slot + ((step*4 + 2 hh + cbi)*4 + m)*256 + lane*4
    const int abase = (2 * hh) * 4 * 256 + lane * 4;
    auto a_ptr = [&](int g, int step, int m) { return lds + OFF_U + (g % NU) * VUF + step * (16 * 256) + m * 256 + abase; };

    // bias vector -> LDS (plain loads BEFORE any DMA is in flight; published by the prologue's first barrier)
    for (int i = tid; i < MAX_BIAS; i += kThreads) lds[OFF_B + i] = (bias && i < p.Cout) ? bias[i] : 0.f;

    // ---- prologue of the stream.  In flight, oldest first: raw(0) raw(1) U(0) raw(2) U(1) raw(3).
    issue_raw();
    issue_raw();
    issue_u(0, UINSTR);
    issue_raw();
    issue_u(0, UINSTR);
    issue_raw();
    wait_dma<3 * RINSTR + 2 * UINSTR>();                     // raw(0) has landed (this wave's part)
    barrier_lds();
    float vA[16], vB[16];
    {
        const PatchPair k0 = load_pair(0, 0, 0), k1 = load_pair(0, 0, 1), k2 = load_pair(0, 0, 2);
        float tt[4][4];
        rows_col(tt, 0, k0.p[0].y, k0.p[1].y, k0.p[2].y, k0.p[3].y);
        rows_col(tt, 1, k1.p[0].x, k1.p[1].x, k1.p[2].x, k1.p[3].x);
        rows_col(tt, 2, k1.p[0].y, k1.p[1].y, k1.p[2].y, k1.p[3].y);
        rows_col(tt, 3, k2.p[0].x, k2.p[1].x, k2.p[2].x, k2.p[3].x);
        cols_of(tt, vA);
    }
    wait_dma<2 * RINSTR + UINSTR>();                         // raw(1), U(0) have landed
    barrier_lds();
    f32x4 a0[2], a1[2];                                      // A operands of the current / next group of 8 MFMAs (cbi = 0, 1)
    a0[0] = *reinterpret_cast<const f32x4*>(a_ptr(0, 0, 0));
    a1[0] = *reinterpret_cast<const f32x4*>(a_ptr(0, 0, 0) + 4 * 256);

    // 8 MFMAs of transform points 4m .. 4m+3 for both channel blocks
#define KMSR_MFMA8(M, A0, A1, V)                                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                     \
        acc[0][4 * (M) + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32((A0)[jj], (V)[4 * (M) + jj], acc[0][4 * (M) + jj], 0, 0, 0); \
        acc[1][4 * (M) + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32((A1)[jj], (V)[4 * (M) + jj], acc[1][4 * (M) + jj], 0, 0, 0); \
    }
    // One step = 32 MFMAs on V(cur) plus the NEXT step's V.  On gfx950 the f32 MFMA and the VALU do NOT co-execute (both run on the
    // vector FMA lanes: profiles/r02_ubench_f32_mfma_valu.log -- every VALU instruction adds ~4.5 cycles to the SIMD's MFMA period,
    // every MFMA <-> VALU alternation ~11), so the VALU work is kept minimal and CLUSTERED, not interleaved; only the LDS requests
    // (A operands one 8-MFMA group ahead, the patch one block ahead) are spread, pinned with sched_barrier:
    //   request two column pairs of the next step's patch + A(group 1);  8 MFMAs;  request A(group 2);  8 MFMAs;
    //   VALU block 1 (12): row transform of patch columns 0..2;  request the third column pair + A(group 3);  8 MFMAs;
    //   [this chunk's DMA instructions];  request A(next step's group 0);
    //   VALU block 2 (20): row transform of column 3, column transform;  8 MFMAs.
    // (g, s) = this step; (pg, ps) = where the next step's patch lives; (ng, ns) = where the next step's A operands live.
    auto step = [&](int g, int s, const float (&vc)[16], float (&vn)[16], int pg, int ps, int ng, int ns, bool issue) {
        PatchPair k0 = load_pair(pg, ps, 0);
        const PatchPair k1 = load_pair(pg, ps, 1);
        a0[1] = *reinterpret_cast<const f32x4*>(a_ptr(g, s, 1));
        a1[1] = *reinterpret_cast<const f32x4*>(a_ptr(g, s, 1) + 4 * 256);
        KMSR_SB();
        KMSR_MFMA8(0, a0[0], a1[0], vc);
        KMSR_SB();
        a0[0] = *reinterpret_cast<const f32x4*>(a_ptr(g, s, 2));
        a1[0] = *reinterpret_cast<const f32x4*>(a_ptr(g, s, 2) + 4 * 256);
        KMSR_SB();
        KMSR_MFMA8(1, a0[1], a1[1], vc);
        KMSR_SB();
        float tt[4][4];
        rows_col(tt, 0, k0.p[0].y, k0.p[1].y, k0.p[2].y, k0.p[3].y);
        rows_col(tt, 1, k1.p[0].x, k1.p[1].x, k1.p[2].x, k1.p[3].x);
        rows_col(tt, 2, k1.p[0].y, k1.p[1].y, k1.p[2].y, k1.p[3].y);
        KMSR_SB();
        k0 = load_pair(pg, ps, 2);                           // third column pair into the first one's registers
        a0[1] = *reinterpret_cast<const f32x4*>(a_ptr(g, s, 3));
        a1[1] = *reinterpret_cast<const f32x4*>(a_ptr(g, s, 3) + 4 * 256);
        KMSR_SB();
        KMSR_MFMA8(2, a0[0], a1[0], vc);
        KMSR_SB();
        if (issue) {                                         // (compile-time after inlining) this chunk's 6 DMA instructions
            issue_u(0, UINSTR);
            issue_raw();
        }
        a0[0] = *reinterpret_cast<const f32x4*>(a_ptr(ng, ns, 0));
        a1[0] = *reinterpret_cast<const f32x4*>(a_ptr(ng, ns, 0) + 4 * 256);
        KMSR_SB();
        rows_col(tt, 3, k0.p[0].x, k0.p[1].x, k0.p[2].x, k0.p[3].x);
        cols_of(tt, vn);
        KMSR_SB();
        KMSR_MFMA8(3, a0[1], a1[1], vc);
        KMSR_SB();
    };

    // ---- epilogue of one item: lane-local output transform, bias / ReLU / skip / PixelShuffle, stores, accumulators cleared
    // ---- epilogue of one item: lane-local output transform, bias / ReLU / skip / PixelShuffle, stores, accumulators cleared.
    // Skip tensor: plain loads to registers, placed so that vmcnt (which retires in order) never makes them wait for fresh DMA
    // or for stores: channel block 0's values are requested right after the last chunk's barrier, BEFORE that chunk's DMA
    // instructions; channel block 1's after block 0's outputs are computed (its accumulators are dead) and BEFORE block 0's stores.
    // Output / skip-tensor addressing of an item: buffer operations on one descriptor per (image, 64-channel group) whose record
    // count ends at the group's last real channel, so lanes of padding channels (co >= Cout) and out-of-image lanes (sentinel offset) are dropped by
    // the range check; a lane's byte offset for (channel block cbi, output row a) is computed once per item, registers r add
    // r * H * W * 4 (one v_add).  (Per-element 64-bit addresses with predicates
    // had cost ~10 VALU per store / skip load -- a third of the kernel's non-MFMA VALU, which competes with the MFMAs for issue.)
    struct ItemPos {
        int n, cg, y0, x0, item;
        unsigned vo[2][2];                     // [cbi][a]
    };
    const unsigned HWb = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;
    auto item_pos = [&](int item) {
        const int tile = item % p.tiles, rest = item / p.tiles;
        ItemPos ip{rest / p.cgs, rest % p.cgs, (tile / p.tiles_x) * (2 * TROWS) + 2 * tb,
                   (tile % p.tiles_x) * (2 * TCOLS) + 2 * (tc0 + t16), item, {{0, 0}, {0, 0}}};   // this lane's 2x2 output pixels: rows y0, y0+1, columns x0, x0+1
#pragma unroll
        for (int cbi = 0; cbi < 2; ++cbi)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int co0 = (2 * hh + cbi) * 16 + 4 * kq;               // channel within the item's 64-channel group
                const bool ok = ip.x0 < W && ip.y0 + a < H;
                unsigned v;
                if constexpr (PS) v = (unsigned)((((co0 >> 2) * (2 * H) + 2 * (ip.y0 + a)) * (2 * W) + 2 * ip.x0)) * 4u;   // out[n, co0/4, 2y.., 2x..]
                else v = (unsigned)((co0 * H + ip.y0 + a) * W + ip.x0) * 4u;
                ip.vo[cbi][a] = ok ? v : OOB;
            }
        return ip;
    };
    auto group_rsrc = [&](const float* base, const ItemPos& ip) {      // n, cg are block-uniform
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + ((long long)ip.n * p.Cout + ip.cg * VCO) * HW), 0,
                                                 (unsigned)min(VCO, p.Cout - ip.cg * VCO) * HWb, 0x00020000);
    };
    auto row_sum16 = [&](float v) {      // sum over the 16 lanes of a DPP row (= the 16 tile columns), valid in lane 15 of the row
#define KMSR_ROW_SHR(SH) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xf, 0xf, true))
        KMSR_ROW_SHR(1); KMSR_ROW_SHR(2); KMSR_ROW_SHR(4); KMSR_ROW_SHR(8);
#undef KMSR_ROW_SHR
        return v;
    };
    auto fetch_residual = [&](const ItemPos& ip, int cbi, f32x2 (&rr)[2][4]) {
        if constexpr (!PS) {
            if (res) {                                                     // block-uniform
                const auto rs = group_rsrc(res, ip);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r)                            // out-of-range lanes read 0 and are dropped at the store
                        rr[a][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, ip.vo[cbi][a] + (unsigned)r * HWb, 0, 0));
            }
        }
    };
    auto outputs_of = [&](const ItemPos& ip, int cbi, f32x2 (&Y)[2][4]) {  // [output row a][register r] = (column x0, column x0+1)
        const int co0 = ip.cg * VCO + (2 * hh + cbi) * 16 + 4 * kq;        // channels co0 .. co0+3 <-> registers 0 .. 3
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + OFF_B + co0);   // co0 % 4 == 0; zeros past Cout
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float ma[4][2];                                                // column transform: MA[i][b] = sum_j M[i][j] A[j][b], A^T = [[1,1,1,0],[0,1,-1,-1]]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = acc[cbi][4 * i + 0][r], m1 = acc[cbi][4 * i + 1][r], m2 = acc[cbi][4 * i + 2][r], m3 = acc[cbi][4 * i + 3][r];
                ma[i][0] = m0 + m1 + m2;
                ma[i][1] = m1 - m2 - m3;
            }
            Y[0][r] = f32x2{ma[0][0] + ma[1][0] + ma[2][0] + bb[r], ma[0][1] + ma[1][1] + ma[2][1] + bb[r]};
            Y[1][r] = f32x2{ma[1][0] - ma[2][0] - ma[3][0] + bb[r], ma[1][1] - ma[2][1] - ma[3][1] + bb[r]};
            if (p.relu) {
                Y[0][r] = f32x2{fmaxf(Y[0][r].x, 0.f), fmaxf(Y[0][r].y, 0.f)};
                Y[1][r] = f32x2{fmaxf(Y[1][r].x, 0.f), fmaxf(Y[1][r].y, 0.f)};
            }
        }
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[cbi][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_outputs = [&](const ItemPos& ip, int cbi, const f32x2 (&Y)[2][4], const f32x2 (&rr)[2][4]) {
        const auto ro = group_rsrc(out, ip);
        if constexpr (PS) {
            // conv channel q = co0 + r -> out[n, q >> 2, 2y + (r >> 1), 2x + (r & 1)]: per conv row y this lane writes the two
            // rows 2y, 2y+1 of out channel co0 / 4, columns 2 x0 .. 2 x0 + 3.  The second row's offset is added in the VGPR: passing it as
            // the instruction's SGPR soffset gave intermittently misplaced first dwords in some waves (next to LDS-DMA traffic)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{Y[a][0].x, Y[a][1].x, Y[a][0].y, Y[a][1].y}), ro,
                                                       ip.vo[cbi][a], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{Y[a][2].x, Y[a][3].x, Y[a][2].y, Y[a][3].y}), ro,
                                                       ip.vo[cbi][a] + (unsigned)(8 * W), 0, 0);
            }
        } else {
            float cs[4] = {0.f, 0.f, 0.f, 0.f};
            const int co0 = ip.cg * VCO + (2 * hh + cbi) * 16 + 4 * kq;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x2 v = Y[a][r];
                    if (res) {                                               // block-uniform, as is p.res_mask
                        if (TRAIN && p.res_mask) v = f32x2{rr[a][r].x > 0.f ? v.x : 0.f, rr[a][r].y > 0.f ? v.y : 0.f};
                        else v = f32x2{v.x + rr[a][r].x, v.y + rr[a][r].y};
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ro, ip.vo[cbi][a] + (unsigned)r * HWb, 0, 0);
                    if constexpr (TRAIN) {
                        if (ip.vo[cbi][a] != OOB && co0 + r < p.Cout) cs[r] += v.x + v.y;
                    }
                }
            // per-channel sums of what this wave just stored (the bias gradient of the layer that consumes this tensor as its
            // output gradient): 16 tile columns reduced in the DPP row, one float4 per (item, tile quarter, 4-channel group);
            // fixed order -> run-to-run identical
            if (TRAIN && csum) {                                             // block-uniform
                const f32x4 t = f32x4{row_sum16(cs[0]), row_sum16(cs[1]), row_sum16(cs[2]), row_sum16(cs[3])};
                if (t16 == 15)
                    *reinterpret_cast<f32x4*>(csum + ((long long)ip.item * 4 + tq) * VCO + (2 * hh + cbi) * 16 + 4 * kq) = t;
            }
        }
    };

    // ---- the chunk stream.  Chunk G: step 0 multiplies V(G, channels 0-3) and computes V(G, channels 4-7); the mid-chunk barrier
    //      publishes U(G+1) / raw(G+2) and frees U(G-1)'s and raw(G)'s slots; step 1 multiplies V(G, 4-7), issues the DMA of
    //      U(G+2) / raw(G+4) and computes V(G+1, 0-3).
    int item = it0, c = 0;
    for (int g = 0; g < T; ++g) {
        step(g, 0, vA, vB, g, 1, g, 1, false);
        wait_dma<RINSTR>();                                  // outstanding: U(g+1), raw(g+3): everything but the newest raw chunk has landed
        barrier_lds();
        f32x2 rr0[2][4];
        ItemPos ip{};
        if (c + 1 == nch) {                                  // block-uniform
            ip = item_pos(item);
            fetch_residual(ip, 0, rr0);
        }
        step(g, 1, vB, vA, g + 1, 0, g + 1, 0, true);
        if (++c == nch) {                                    // block-uniform
            c = 0;
            f32x2 Y0[2][4], Y1[2][4], rr1[2][4];
            outputs_of(ip, 0, Y0);
            fetch_residual(ip, 1, rr1);
            store_outputs(ip, 0, Y0, rr0);
            outputs_of(ip, 1, Y1);
            store_outputs(ip, 1, Y1, rr1);
            item += istep;
        }
    }
#undef KMSR_MFMA8
}
#undef KMSR_SB
#undef KMSR_FENCE

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv3x3_winov_packed_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)ceil_div(Cout, VCO) * ceil_div(Cin, VCK) * VUF;
}

extern "C" int kmsr_conv3x3_winov_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, float* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "winograd-vreg pack: bad argument");
    const int nchunks = ceil_div(Cin, VCK);
    const long long total = (long long)ceil_div(Cout, VCO) * nchunks * VUF;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    winov_pack_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, packed, nchunks, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" size_t kmsr_conv3x3_winov_chan_sum_floats(const kmsr_conv3x3_desc* d) {
    if (!d || d->N <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return 0;
    return (size_t)ceil_div(d->W, 2 * TCOLS) * ceil_div(d->H, 2 * TROWS) * ceil_div(d->Cout, VCO) * d->N * 4 * VCO;
}

extern "C" int kmsr_conv3x3_winov_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                                          const float* bias, const float* residual, float* out, void* stream_) {
    return kmsr_conv3x3_winov_fwd_ex_f32(d, x, packed_w, bias, residual, out, nullptr, stream_);
}

extern "C" int kmsr_conv3x3_winov_fwd_ex_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                                             const float* bias, const float* residual, float* out, float* chan_sum,
                                             void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->flags & ~(uint32_t)KMSR_CONV_FLAG_RELU_MASK) return fail(KMSR_E_BADARG, "unknown conv flags 0x%x", d->flags);
    if ((d->flags & KMSR_CONV_FLAG_RELU_MASK) && (!residual || d->relu || d->pixel_shuffle))
        return fail(KMSR_E_BADARG, "KMSR_CONV_FLAG_RELU_MASK needs the mask tensor in `residual` and no relu / pixel_shuffle");
    if (chan_sum && d->pixel_shuffle) return fail(KMSR_E_UNSUPPORTED, "channel sums are not produced together with pixel_shuffle");
    if (reinterpret_cast<uintptr_t>(chan_sum) % 16) return fail(KMSR_E_BADARG, "chan_sum must be 16-byte aligned");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "winograd conv needs stride 1");
    if (d->W % 4) return fail(KMSR_E_UNSUPPORTED, "winograd-vreg conv needs W %% 4 == 0 (W=%d): rows are fetched as aligned 16-byte pieces", d->W);
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(packed_w) % 16 || reinterpret_cast<uintptr_t>(out) % 8 ||
        reinterpret_cast<uintptr_t>(residual) % 8)
        return fail(KMSR_E_BADARG, "winograd-vreg conv needs 16-byte aligned input / weights and 8-byte aligned output / residual");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (bias && d->Cout > MAX_BIAS) return fail(KMSR_E_UNSUPPORTED, "winograd-vreg conv stages the bias vector in LDS: Cout <= %d (Cout=%d)", MAX_BIAS, d->Cout);
    if ((long long)d->H * d->W * VCK >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "image plane exceeds 2^28 elements");
    if ((long long)d->H * d->W * 4 * VCO >= (1ll << 31))      // 32-bit byte offsets inside one 64-channel output group
        return fail(KMSR_E_UNSUPPORTED, "winograd-vreg conv: H*W = %lld exceeds 2^23 pixels per plane", (long long)d->H * d->W);
    WinoVParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, 2 * TCOLS);
    p.tiles = p.tiles_x * ceil_div(d->H, 2 * TROWS);
    p.cgs = ceil_div(d->Cout, VCO);
    p.nchunks = ceil_div(d->Cin, VCK);
    p.relu = d->relu;
    p.res_mask = (d->flags & KMSR_CONV_FLAG_RELU_MASK) ? 1 : 0;
    const long long nitems = (long long)p.tiles * p.cgs * d->N;
    if (nitems > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_items = (int)nitems;
    // persistent workgroups: one per CU, each walks ceil(items / grid) consecutive items
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)std::min<long long>(nitems, cus);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->pixel_shuffle)
        conv3x3_winov_kernel<true, false><<<grid, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, nullptr, p);
    else if (chan_sum || p.res_mask)
        conv3x3_winov_kernel<false, true><<<grid, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, chan_sum, p);
    else
        conv3x3_winov_kernel<false, false><<<grid, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, nullptr, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
