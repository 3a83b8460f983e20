// Winograd F(4x4,3x3) forward convolution on the fp32 matrix cores (semantics: include/kmsr.h, kmsr_conv3x3_wino4_fwd_f32).
//
// Why a second Winograd kernel.  conv3x3_wino_vreg.hip (F(2x2,3x3)) sits at 0.56-0.58 of the fp32 MFMA pipe and its remaining
// overheads (transform VALU that cannot co-issue with the f32 MFMA, the LDS-DMA weight stream, the item hand-over) are structural
// (DESIGN 4.4b).  F(4x4,3x3) needs 36 multiplies per 16 outputs and (ci, co) pair instead of 64: 1.78x fewer MFMAs for the same
// convolution, and -- per output -- fewer transform instructions as well (78 + 54 packed VALU per 6x6 patch / 4x4 block instead of
// 4 x (32 + 24) scalar ones).  Arithmetic stays fp32 (v_mfma_f32_16x16x4_f32 is an exact fma chain); the transforms with the points
// {0, +-1, +-2, inf} re-associate the sums: measured max error vs float64 5e-6 .. 8e-6 of the output scale for one 64 -> 64 layer
// (F(2x2,3x3): 4e-7; tests hold 2e-5), 1.4e-6 through the 33-layer SR body (the skip connections carry the scale).
//
// Design (one persistent 4-wave workgroup per CU, ONE wave per SIMD with the whole 512-entry register file):
//   * MFMA tile 16 channels x 16 tiles x 4 input channels.  A wave owns 32 output channels x 16 tiles (one tile row of 4x4 tiles =
//     4 output rows x 64 columns) for ALL 36 transform points: 2 x 36 accumulator tiles x 4 registers = 288 registers -- more than
//     the 256 AGPRs: the first QVG point groups accumulate in ArchVGPRs through the VGPR form of the instruction (inline asm), the
//     other 224 registers sit in AGPRs.  Lane (tile t = lane & 15, channel kq = lane >> 4) supplies the B operand
//     V[xi][4 s + kq][t] for every xi: exactly the 6x6 input transform B^T d B of ITS patch, which it computes itself (78 packed-fp32
//     VALU per 72 MFMAs) from the raw rows in LDS.  The output transform A^T M A is lane-local too: no exchange, no epilogue barrier.
//   * the workgroup = 2 channel halves x 2 tile rows = 64 channels x (8 rows x 64 columns) per item, PERSISTENT over an interleaved
//     item list (skip-form launches walk it backwards: the next launch of a residual network starts on the lines stored last); the
//     4-channel steps of all its items form ONE stream.
//   * A operands (the pre-transformed weights U = G g G^T, 36 x 64 x 4 floats = 36 KB per step and workgroup) do NOT go through
//     LDS: every wave loads the 18 KB of ITS 32 channels per step with 18 buffer_load_dwordx4 straight into the registers the MFMAs
//     read (packed lane-linear: 1 KB per instruction), 4 .. 9 groups ahead in a 5-slot register ring -- the stream is the same 590 KB
//     for every workgroup of a layer, i.e. L2 / L1 hits.  (The LDS form -- LDS-DMA + ds_read_b128 -- was built and measured: no faster.)
//   * raw input rows (4 channels x 10 rows x 72 columns = 12 KB per step) arrive by buffer LDS-DMA in a 3-slot ring, requested three
//     steps ahead whatever item they belong to; padding (image border, channel tail, end of stream) is an out-of-range buffer
//     offset, for which the hardware writes zeros: the transform has no masks and every wave issues the same 3 DMA instructions
//     per step.  ONE barrier per step.
//   * a step is written as 36 MFMA pairs with slots between them; the raw request, the 12 LDS reads of the next patch, the 18 A
//     requests and the scalar bookkeeping of the cursors sit in those slots one or two instructions at a time (a scalar instruction
//     between two MFMAs costs ~1 cycle, an LDS read 4-8, a 16-byte load 20-50: tools/ubench/mfma_salu.hip, mfma_store.hip).
//     Nothing lane-shaped stays live across the steps: the kernel has no VGPR spills (tests/test_abi.py greps the ISA).
//   * vmcnt retires in order: a step's 18 weight loads, 3 DMA instructions and an item's 32 stores (+ 32 skip-tensor loads) share
//     one queue.  hipcc counts the register loads itself; the DMA is covered by an explicit counted wait before the barrier.
// Requires W % 4 == 0 and 16-byte aligned x / packed_w / out / residual.
#include "kmsr_common.h"
#include "kmsr_wino4_xf.h"

#include <algorithm>
#include <type_traits>

namespace kmsr {
namespace {

using namespace wino4xf;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int QCK = 4;                  // input channels per step (one MFMA K)
constexpr int QCO = 64;                 // output channels per workgroup
constexpr int QTR = 2, QTC = 16;        // tiles per workgroup: 2 x 16 (output 8 rows x 64 columns)
constexpr int QT = 256;                 // threads: 4 waves = (channel half hh) x (tile row tg)
constexpr int QPTS = 36;
constexpr int QUF = QPTS * QCK * QCO;   // 9216 floats = 36 KB of U per step: [hh 2][group 9][cbi 2][lane 64][jj 4]
constexpr int QGROUPS = 9;              // groups of 4 transform points (8 MFMAs: 4 points x 2 channel blocks)
constexpr int QRROWS = 4 * QTR + 2;     // 10 input rows
constexpr int QRQ = (4 * QTC + 8) / 4;  // 18 float4 per row: columns x0-4 .. x0+67
constexpr int QRCOLS = 4 * QRQ;         // 72
constexpr int QPLANE_Q = 192;           // quads per channel plane in LDS (180 real + 12 pad: plane pitch 768 floats = 0 mod 64 banks ->
                                        // the ds_read_b128 of a 16-lane tile row and of the 4 channels fall on disjoint banks)
constexpr int QPLANE = 4 * QPLANE_Q;    // 768 floats
constexpr int QRAWF = QCK * QPLANE;     // 3072 floats = 12 KB per ring slot = 12 wave-instructions of 1 KB
constexpr int QRINSTR = QRAWF / 4 / QT; // 3 DMA instructions per wave and step
constexpr int QNRAW = 3;                // ring slots: step s reads raw(s+1) and requests raw(s+3) into the slot raw(s) has left
constexpr int QNA = 5;                  // register slots for A-operand groups: group g of a step lives in slot g % 5 and is requested 4 .. 9
                                        // groups (>= 24 MFMAs) before its use
constexpr int QVG = 2;                  // accumulator groups kept in ArchVGPRs (2 x 8 tiles x 4 = 64 registers); the other 7 x 32 = 224 in AGPRs
constexpr int QMAX_BIAS = 1024;
constexpr int QOFF_R = 0, QOFF_B = QNRAW * QRAWF, QLDS_FLOATS = QOFF_B + QMAX_BIAS;
struct Wino4Params {
    int N, Cin, H, W, Cout;
    int tiles_x, tiles, cgs, nsteps, relu, n_items;
    int res_mask;                       // W4_TRAIN: the skip tensor is a ReLU-mask source: out = res > 0 ? conv : 0 (instead of conv + res)
};
// epilogue families: bias / ReLU / residual; + fused PixelShuffle(2) (no residual); + the training step's extras (ReLU-mask form of the
// skip tensor, per-channel sums of the stored tensor) -- same semantics as conv3x3_wino_vreg.hip's PS / TRAIN instantiations
enum { W4_PLAIN = 0, W4_PS = 1, W4_TRAIN = 2 };

// packed[cg][step][hh][grp][cbi][lane][jj] = (G g G^T)[xi = 4 grp + jj]  for
// (co, ci) = (cg*64 + hh*32 + cbi*16 + (lane & 15), step*4 + (lane >> 4)), zero padded; xi = 6 i + j.
// G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]; computed in double, rounded once.
// tf (round 4): w is the FORWARD layer's [Cin][Cout][3][3] tensor and the packed convolution is its data gradient, w'[co][ci][u][v] =
// w[ci][co][2-u][2-v] -- the flip + transpose + copy launches torch needed per layer and step are an index swap here (Cin / Cout below are
// those of the PACKED convolution)
__global__ void wino4_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int tf, float* __restrict__ packed, int nsteps,
                                  long long total) {
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const int cbi = (int)((idx >> 8) & 1);
        long long r = idx >> 9;
        const int grp = (int)(r % QGROUPS); r /= QGROUPS;
        const int hh = (int)(r & 1); r >>= 1;
        const int step = (int)(r % nsteps);
        const int cg = (int)(r / nsteps);
        const int xi = 4 * grp + jj, i = xi / 6, j = xi % 6;
        const int cog = cg * QCO + hh * 32 + cbi * 16 + (lane & 15), cig = step * QCK + (lane >> 4);
        double v = 0.0;
        if (cog < Cout && cig < Cin) {
            const float* g = tf ? w + ((long long)cig * Cout + cog) * 9 : w + ((long long)cog * Cin + cig) * 9;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) v += G[i][a] * (double)g[tf ? (2 - a) * 3 + (2 - b) : a * 3 + b] * G[j][b];
        }
        packed[idx] = (float)v;
    }
}

constexpr int q_wait_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }
#define Q_FENCE() asm volatile("" ::: "memory")
#define Q_SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void q_barrier_lds() {   // LDS hand-over only (LDS-DMA in flight is covered by the caller's counted wait)
    Q_FENCE();
    __builtin_amdgcn_s_waitcnt(q_wait_imm(63, 0));
    __builtin_amdgcn_s_barrier();
    Q_FENCE();
}
template <int N>
__device__ __forceinline__ void q_wait_vm() {       // all but the N newest vector-memory operations of this wave have completed
    Q_FENCE();
    __builtin_amdgcn_s_waitcnt(q_wait_imm(N, 15));
    Q_FENCE();
}

template <bool HAS_RES, int MODE>
__global__ void __launch_bounds__(QT, 1) conv3x3_wino4_kernel(const float* __restrict__ xg, const float* __restrict__ wpk,
                                                              const float* __restrict__ bias, const float* __restrict__ res,
                                                              float* __restrict__ out, float* __restrict__ csum, Wino4Params p) {
    static_assert(!(HAS_RES && MODE == W4_PS), "no skip tensor together with PixelShuffle");
    __shared__ __attribute__((aligned(16))) float lds[QLDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = wv & 1, tg = wv >> 1;                    // channel half, tile row
    const int H = p.H, W = p.W, S = p.nsteps;
    const long long HW = (long long)H * W;
    const unsigned HWb = (unsigned)HW * 4u;

    // items of this workgroup: vb, vb + nwg, ...  (vb = XCD-grouped id; see conv3x3_wino_vreg.hip: neighbouring tiles run at the
    // same time on the CUs of one XCD, so halo rows are L2 hits)
    const int nwg = gridDim.x, bq = nwg >> 3, br = nwg & 7, xcd = blockIdx.x & 7;
    const int vb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    if (vb >= p.n_items) return;                            // block-uniform
    const int n_my = (p.n_items - vb + nwg - 1) / nwg;
    // Skip-form launches walk the items BACKWARDS.  In a residual network the forms alternate (conv + ReLU, conv + skip, ...) and every
    // launch reads what the one before it wrote: walking the same way each time, a launch starts on the tensor's oldest part -- long
    // evicted from the 256 MB memory-side cache by the 268 MB that followed -- and evicts the rest just ahead of reading it.
    constexpr bool REV = HAS_RES;
    const int it0 = REV ? p.n_items - 1 - vb : vb, istep = REV ? -nwg : nwg;
    const int T = n_my * S;                                 // steps in this workgroup's stream

    // ---- A-operand stream: step gA of the stream = step cA of item itA (only its cout group matters).  a_cur / a_nxt = byte
    //      offsets of the step being multiplied and of the one after it.
    const auto urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned a_lane = (unsigned)(hh * (QGROUPS * 2 * 256) + lane * 4) * 4u;   // byte offset of this lane inside a step of U
    int gN = 0, cA = 0, itA = it0;                           // a_nxt = stream step gN = step cA of item itA
    auto a_item = [&](int it) { return p.cgs == 1 ? 0u : (unsigned)(((it / p.tiles) % p.cgs) * S) * (unsigned)(QUF * 4); };
    unsigned a_cur = a_item(it0), a_nxt = a_cur;
    unsigned a_nn = a_cur;                                  // the step after a_nxt (computed early in a step, taken over at its end)
    auto a_lookahead = [&]() {                              // a_nn <- the step after a_nxt (past the end of the stream: stays)
        a_nn = a_nxt;
        if (gN + 1 < T) {
            ++gN;
            if (++cA == S) {
                cA = 0;
                itA += istep;
                a_nn = a_item(itA);
            } else {
                a_nn = a_nxt + (unsigned)(QUF * 4);
            }
        }
    };
    auto advance_a = [&]() {                                // prologue form: a_cur <- a_nxt <- the step after it
        a_lookahead();
        a_cur = a_nxt;
        a_nxt = a_nn;
    };
    f32x4 A[QNA][2];
    auto load_a1 = [&](unsigned step_off, int g, int cbi) {  // group g, channel block cbi of the step at step_off -> slot g % QNA
        A[g % QNA][cbi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, a_lane, step_off + (unsigned)((g * 2 + cbi) * 1024), 0));
    };
    auto load_a = [&](unsigned step_off, int g) {
        load_a1(step_off, g, 0);
        load_a1(step_off, g, 1);
    };

    // ---- raw rows: step gR = step cR of item itR.  Lane-linear quads: L = (3 wv + i) * 64 + lane <-> (channel = L / 192,
    //      u = L % 192: row = u / 18, q = u % 18; u >= 180: plane padding).  BUFFER LDS-DMA on one descriptor per image whose
    //      record count ends at the last real channel: out-of-image pieces and plane padding carry an out-of-range offset, channel
    //      tails fall past the end of the records -- the hardware writes zeros for all of them (tools/ubench/buffer_lds_oob.hip), so
    //      the transform needs no masks and no zero page.  (Not global_load_lds: hipcc models that FLAT-encoded instruction as
    //      touching LDS *and* global memory and then drains vmcnt(0) / lgkmcnt(0) at the next counted dependency -- here the first
    //      MFMA that reads a loaded weight register, i.e. a full HBM round trip per step.)
    constexpr unsigned ROOB = 0x80000000u;
    int gR = 0, cR = 0, itR = it0;                          // next raw request = stream step gR = step cR of item itR
    unsigned roff[QRINSTR];
    const float* xstep = xg;                                // first channel of the step to request next
    int ch_left = 0;                                        // channels from there to the end of the image's channels
    int wr_off = 0;                                         // ring slot (float offset) the next request lands in
    auto raw_map = [&]() {                                  // the cursor enters item itR
        const int tile = itR % p.tiles, n = itR / (p.tiles * p.cgs);
        const int ty0 = (tile / p.tiles_x) * (4 * QTR), tx0 = (tile % p.tiles_x) * (4 * QTC);
        xstep = xg + (long long)n * p.Cin * HW;
        ch_left = p.Cin;
        int ln = lane;
        asm volatile("" : "+v"(ln));                         // everything below is recomputed HERE, once per item (hoisted it would be spilled)
#pragma unroll
        for (int i = 0; i < QRINSTR; ++i) {
            const int L = (QRINSTR * wv + i) * 64 + ln;      // this lane's quad of DMA instruction i: channel L / 192, u = L % 192 (u >= 180: plane padding)
            const int ch = L / QPLANE_Q, u = L - ch * QPLANE_Q, row = u / QRQ, q = u - row * QRQ;
            const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * q;
            const bool ok = u < QRROWS * QRQ && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
            roff[i] = ok ? (unsigned)(ch * (int)HW + gy * W + gx) * 4u : ROOB;
        }
    };
    raw_map();
    unsigned rec_cur = (unsigned)p.Cin * HWb;                // records of the next request's descriptor = max(ch_left, 0) planes
    auto issue_dma1 = [&](int i) {                          // DMA instruction i of the request (same descriptor as issue_dma)
        const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xstep), 0, rec_cur, 0x00020000);
        float* rb = lds + QOFF_R + wr_off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t*)(rb + (QRINSTR * wv + i) * 256), 16, roff[i], 0, 0, 0);
    };
    auto issue_dma = [&]() {
        // one descriptor per (image, step): base = the step's first channel, records = the channels that are left (raw-buffer range
        // checks cover the VGPR offset only, not soffset -- so the channel position lives in the base).  Past the end of the stream:
        // an empty descriptor (every lane out of range -> zeros; that slot is never multiplied).
        const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xstep), 0, rec_cur, 0x00020000);
        float* rb = lds + QOFF_R + wr_off;
#pragma unroll
        for (int i = 0; i < QRINSTR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t*)(rb + (QRINSTR * wv + i) * 256), 16, roff[i], 0, 0, 0);
    };
    // the request cursor moves on in three pieces (inside a step they sit in the shadow of MFMAs; scalar instructions issued between
    // two MFMAs of the same wave cost ~1 cycle each there, ~8 in a cluster of their own: tools/ubench/mfma_salu.hip)
    auto raw_next_a = [&]() {
        xstep += (long long)QCK * HW;
        ch_left -= QCK;
        ++gR;
    };
    auto raw_next_b = [&]() {
        if (++cR == S) {
            cR = 0;
            itR += istep;
            if (gR < T) raw_map();
            else ch_left = 0;
        }
    };
    auto raw_next_c = [&]() { rec_cur = (unsigned)max(ch_left, 0) * HWb; };
    auto ring_next = [&](int off) { return off == (QNRAW - 1) * QRAWF ? 0 : off + QRAWF; };
    auto issue_raw = [&]() {                                // prologue form
        issue_dma();
        wr_off = ring_next(wr_off);
        raw_next_a();
        raw_next_b();
        raw_next_c();
    };

    // ---- patch of lane (tile t16, channel kq) in step g: raw rows 4 tg + r (r = 0..5), raw columns 4 t16 + 3 + j (j = 0..5):
    //      one ds_read_b128 (columns 4 t16 + 4 .. + 7, j = 1..4) + one ds_read2_b32 (j = 0 and j = 5) per row.
    const int pbase = kq * QPLANE + (4 * tg) * QRCOLS + 4 * t16 + 3;
    struct Patch { f32x4 mid[6]; f32x2 e[6]; };             // e = (column 0, column 5)
    int rd_off = 0;                                         // ring slot (float offset) of the patch to read next: raw(0), raw(1), ...
    auto load_patch = [&]() {
        const float* rp = lds + QOFF_R + rd_off + pbase;
        Patch P;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            P.mid[r] = *reinterpret_cast<const f32x4*>(rp + r * QRCOLS + 1);
            P.e[r] = f32x2{rp[r * QRCOLS], rp[r * QRCOLS + 5]};
        }
        return P;
    };
    auto read_patch = [&](Patch& P, int k) {                // LDS read k of the 12 (one per MFMA pair inside a step: four waves posting
        const float* rp = lds + QOFF_R + rd_off + pbase;    // all 12 right behind the barrier keep the LDS pipe -- and, in order
        const int r = k >> 1;                               // behind it, every wave's first MFMAs -- busy for ~300 cycles)
        // (The 4-way bank conflict of the two edge dwords -- all 64 lanes on the 16 banks = 3 mod 4, whatever the plane pitch -- is real:
        //  59 % of the LDS-active cycles.  Replacing them by ONE dword per row plus two DPP row shifts of the neighbours' quads removed
        //  83 % of the conflict cycles and made the kernel 1-4 % SLOWER: + 12 VALU per patch on the issue port the MFMAs share, while
        //  no wave ever waits for these reads (SQ_WAIT_INST_LDS 0.1 % of the wave cycles).  profiles/r04_wino4_dpp_edges_null.log)
        if (k & 1) P.e[r] = f32x2{rp[r * QRCOLS], rp[r * QRCOLS + 5]};
        else P.mid[r] = *reinterpret_cast<const f32x4*>(rp + r * QRCOLS + 1);
    };
    // V = B^T d B.  Column pass (T = B^T d) on the column PAIRS (0,5), (1,2), (3,4) -- the ds_read2_b32 result and the two halves of
    // the ds_read_b128 quad are already register pairs: 3 x 12 packed VALU.  Row pass (V[i][:] = T[i][:] B) inside each row on the
    // same pairs: 6 x 7 packed VALU.  78 VALU per patch instead of 144 scalar ones.
    struct TPairs { f32x2 p0[6], p1[6], p2[6]; };           // per row i: (t0,t5), (t1,t2), (t3,t4)
    auto cols_pass = [&](const Patch& P, TPairs& t) {
        bt6p(P.e[0], P.e[1], P.e[2], P.e[3], P.e[4], P.e[5], t.p0[0], t.p0[1], t.p0[2], t.p0[3], t.p0[4], t.p0[5]);
        asm volatile("" : "+v"(t.p0[0]), "+v"(t.p0[1]), "+v"(t.p0[2]), "+v"(t.p0[3]), "+v"(t.p0[4]), "+v"(t.p0[5]));   // keep the VALU work where it is written
#define Q_LOHALF(X) f32x2{(X).x, (X).y}
#define Q_HIHALF(X) f32x2{(X).z, (X).w}
        bt6p(Q_LOHALF(P.mid[0]), Q_LOHALF(P.mid[1]), Q_LOHALF(P.mid[2]), Q_LOHALF(P.mid[3]), Q_LOHALF(P.mid[4]), Q_LOHALF(P.mid[5]),
             t.p1[0], t.p1[1], t.p1[2], t.p1[3], t.p1[4], t.p1[5]);
        asm volatile("" : "+v"(t.p1[0]), "+v"(t.p1[1]), "+v"(t.p1[2]), "+v"(t.p1[3]), "+v"(t.p1[4]), "+v"(t.p1[5]));
        bt6p(Q_HIHALF(P.mid[0]), Q_HIHALF(P.mid[1]), Q_HIHALF(P.mid[2]), Q_HIHALF(P.mid[3]), Q_HIHALF(P.mid[4]), Q_HIHALF(P.mid[5]),
             t.p2[0], t.p2[1], t.p2[2], t.p2[3], t.p2[4], t.p2[5]);
        asm volatile("" : "+v"(t.p2[0]), "+v"(t.p2[1]), "+v"(t.p2[2]), "+v"(t.p2[3]), "+v"(t.p2[4]), "+v"(t.p2[5]));
#undef Q_LOHALF
#undef Q_HIHALF
    };
    f32x2 K41 = f32x2{4.f, 1.f}, K54 = f32x2{-5.f, 4.f}, K15 = f32x2{1.f, -5.f};
    asm volatile("" : "+v"(K41), "+v"(K54), "+v"(K15));     // loop-invariant register pairs (not re-materialised per use)
    auto rows_pass = [&](const TPairs& t, int i0, int i1, float (&v)[QPTS]) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            f32x2 O05, O12, O34;
            bt6row(t.p0[i], t.p1[i], t.p2[i], K41, K54, K15, O05, O12, O34);
            asm volatile("" : "+v"(O05), "+v"(O12), "+v"(O34));
            v[6 * i + 0] = O05.x; v[6 * i + 5] = O05.y;
            v[6 * i + 1] = O12.x; v[6 * i + 2] = O12.y;
            v[6 * i + 3] = O34.x; v[6 * i + 4] = O34.y;
        }
    };

    f32x4 acc[2][QPTS];    // [channel block of 16][xi]: rows = channels 4 kq + reg, column = tile t16

    // bias vector -> LDS (plain stores BEFORE any DMA is in flight; published by the prologue's barrier)
    for (int i = tid; i < QMAX_BIAS; i += QT) lds[QOFF_B + i] = (bias && i < p.Cout) ? bias[i] : 0.f;

    // ---- prologue: raw(0..2), A(0); then V(0)
    issue_raw();
    issue_raw();
    issue_raw();
    advance_a();                                            // a_cur = step 0, a_nxt = step 1
#pragma unroll
    for (int g = 0; g < QNA; ++g) load_a(a_cur, g);
    q_wait_vm<0>();
    q_barrier_lds();
    float vv[QPTS];                                          // V of the step being multiplied; overwritten in place by the next step's V
    {                                                        // (rows 0-2 once the MFMAs of points 0 .. 19 have issued, rows 3-5 after the last group)
        const Patch P = load_patch();
        rd_off = ring_next(rd_off);
        TPairs t;
        cols_pass(P, t);
        rows_pass(t, 0, 6, vv);
    }

    // ---- one step: 72 MFMAs on V(cur) / A(cur), next step's A requested group by group into the registers just consumed, next
    //      step's V computed in three VALU clusters (the f32 MFMA and the VALU do not co-issue on gfx950: cluster, do not interleave)
    // The 288 accumulator registers exceed the 256 AGPRs: hipcc selects the AGPR form for every builtin MFMA of a kernel that needs
    // AGPRs and would copy the surplus tiles in and out of a[0:3] around each use.  The QVG first groups (transform points 0 .. 4 QVG - 1)
    // therefore accumulate in ArchVGPRs through the VGPR form of the instruction (inline asm); they are the first MFMAs of every step, so
    // >= 56 other MFMAs lie between their last write and the epilogue's VALU reads (no MFMA-result hazard to pad by hand).
#define Q_MFMA2(G, JJ, VC, FIRST)                                                                                         \
    {                                                                                                                     \
        constexpr int jj = (JJ);                                                                                          \
        _Pragma("unroll") for (int cbi = 0; cbi < 2; ++cbi) {                                                             \
            if constexpr ((G) < QVG) {                                                                                    \
                if (FIRST)                                                                                                \
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[cbi][4 * (G) + jj]) : "v"(A[(G) % QNA][cbi][jj]), "v"((VC)[4 * (G) + jj])); \
                else                                                                                                      \
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[cbi][4 * (G) + jj]) : "v"(A[(G) % QNA][cbi][jj]), "v"((VC)[4 * (G) + jj])); \
            } else {                                                                                                      \
                acc[cbi][4 * (G) + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(G) % QNA][cbi][jj], (VC)[4 * (G) + jj],    \
                                                                              (FIRST) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[cbi][4 * (G) + jj], 0, 0, 0); \
            }                                                                                                             \
        }                                                                                                                 \
    }
    // ---- output / skip-tensor addressing of an item: buffer operations on one descriptor per (image, 64-channel group) whose record
    //      count ends at the group's last real channel: padding channels and out-of-image lanes are dropped by the range check.
    constexpr unsigned OOB = 0x80000000u;
    const float* cur_res = res;                              // (image, cout group) of the skip tensor / the output of the item being multiplied
    float* cur_out = out;
    unsigned cur_nrec = 0;
    int cur_cg = 0, cur_item = it0;
    int cur_y0 = 0, cur_xt = 0;                              // first output row of this wave's tile row; first column of the item's 16 tiles
    // Only scalars describe the item while its steps run: the per-lane store offsets are derived inside the epilogue.  (Held in
    // VGPRs across the steps they were spilled to scratch, and every reload inside the epilogue was a vmcnt(0) wait on the stores
    // and prefetches in flight.)
    auto set_item = [&](int item) {
        const int tile = item % p.tiles, rest = item / p.tiles;
        const int n = rest / p.cgs, cg = rest % p.cgs;
        cur_y0 = (tile / p.tiles_x) * (4 * QTR) + 4 * tg;
        cur_xt = (tile % p.tiles_x) * (4 * QTC);
        cur_cg = cg;
        cur_item = item;
        if constexpr (MODE == W4_PS) {
            // conv channel q -> out[n, q >> 2, 2y + (q >> 1 & 1), 2x + (q & 1)]: the group's 64 conv channels are 16 output planes of
            // 2H x 2W; this lane's four channels 4 kq .. 4 kq + 3 of a block are ONE plane (hh * 8 + cbi * 4 + kq)
            const int ocs = p.Cout >> 2;
            cur_out = out + ((long long)n * ocs + cg * 16) * (4 * HW);
            cur_nrec = (unsigned)min(16, ocs - cg * 16) * 4u * HWb;
        } else {
            const long long go = ((long long)n * p.Cout + cg * QCO) * HW;
            cur_res = res + go;
            cur_out = out + go;
            cur_nrec = (unsigned)min(QCO, p.Cout - cg * QCO) * HWb;
        }
    };
    // byte offset of (this lane's first channel, row cur_y0 + a, column cur_xt + 4 t16) inside the item's descriptor, OOB when outside the image
    auto lane_offsets = [&](unsigned (&vo)[4]) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                         // derived from the lane id HERE (nothing lane-shaped stays live across the steps)
        const int t = ln & 15, k4 = ln >> 4;
        const int x0 = cur_xt + 4 * t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int y = cur_y0 + a;
            unsigned o;
            if constexpr (MODE == W4_PS) o = (unsigned)((hh * 8 + k4) * 4 * (int)HW + 2 * y * (2 * W) + 2 * x0) * 4u;
            else o = (unsigned)((2 * hh * 16 + 4 * k4) * (int)HW + y * W + x0) * 4u;
            vo[a] = (x0 < W && y < H) ? o : OOB;
        }
    };
    auto koff = [&](int k) { return (unsigned)((k >> 2) * 16 + (k & 3)) * HWb; };      // sub-block k = cbi * 4 + r: channel offset inside the lane's 32
    set_item(it0);

    // FIRST (an item's first step: accumulators start from zero) is a literal at both call sites and folds after inlining.  (Not a
    // generic lambda with a tag type: hipcc's host pass then fails to instantiate the <true, true> kernel.)
    auto step = [&](const bool FIRST) __attribute__((always_inline)) {
        float (&vc)[QPTS] = vv;
        float (&vn)[QPTS] = vv;
        // raw(g+1) was requested at the top of step g-2; since then this wave has issued 18 + 3 + 18 vector-memory operations
        // (+ an item's stores / skip-tensor requests, which only makes the count conservative)
        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();
        q_barrier_lds();                                     // raw(g+1) visible to every wave; raw(g)'s slot free
        Patch P;
        TPairs t;
        Q_SB();
        // 36 MFMA pairs (one transform point, both channel blocks).  Everything else of the step sits BETWEEN pairs, one or two
        // instructions at a time, where the matrix pipe is busy anyway: the raw request raw(g+3) (3 DMA), the 12 LDS reads of the
        // next patch, the 18 A requests, and the scalar bookkeeping of the two cursors.
#define Q_P(G, JJ) Q_MFMA2(G, JJ, vc, FIRST); Q_SB()
        Q_P(0, 0); issue_dma1(0); read_patch(P, 0); Q_SB();
        Q_P(0, 1); issue_dma1(1); read_patch(P, 1); Q_SB();
        Q_P(0, 2); issue_dma1(2); read_patch(P, 2); Q_SB();
        Q_P(0, 3); load_a1(a_cur, 5, 0); read_patch(P, 3); Q_SB();
        Q_P(1, 0); load_a1(a_cur, 5, 1); read_patch(P, 4); Q_SB();
        Q_P(1, 1); raw_next_a(); read_patch(P, 5); Q_SB();
        Q_P(1, 2); read_patch(P, 6); Q_SB();
        Q_P(1, 3); load_a1(a_cur, 6, 0); read_patch(P, 7); Q_SB();
        Q_P(2, 0); load_a1(a_cur, 6, 1); read_patch(P, 8); Q_SB();
        Q_P(2, 1); read_patch(P, 9); Q_SB();
        Q_P(2, 2); read_patch(P, 10); Q_SB();
        Q_P(2, 3); load_a1(a_cur, 7, 0); read_patch(P, 11); Q_SB();
        Q_P(3, 0); load_a1(a_cur, 7, 1); Q_SB();
        Q_P(3, 1); wr_off = rd_off; rd_off = ring_next(rd_off); Q_SB();
        Q_P(3, 2); raw_next_b(); Q_SB();
        Q_P(3, 3); load_a1(a_cur, 8, 0); load_a1(a_cur, 8, 1); Q_SB();
        cols_pass(P, t);                                     // 36 packed VALU
        Q_SB();
        Q_P(4, 0); raw_next_c(); Q_SB();
        Q_P(4, 1); a_lookahead(); Q_SB();
        Q_P(4, 2);
        Q_P(4, 3); load_a1(a_nxt, 4, 0); load_a1(a_nxt, 4, 1); Q_SB();
        rows_pass(t, 0, 3, vn);                              // 21 packed VALU
        Q_SB();
        Q_P(5, 0);
        Q_P(5, 1);
        Q_P(5, 2);
        Q_P(5, 3); load_a1(a_nxt, 0, 0); Q_SB();
        Q_P(6, 0); load_a1(a_nxt, 0, 1); Q_SB();
        Q_P(6, 1);
        Q_P(6, 2);
        Q_P(6, 3); load_a1(a_nxt, 1, 0); Q_SB();
        Q_P(7, 0); load_a1(a_nxt, 1, 1); Q_SB();
        Q_P(7, 1);
        Q_P(7, 2);
        Q_P(7, 3); load_a1(a_nxt, 2, 0); Q_SB();
        Q_P(8, 0); load_a1(a_nxt, 2, 1); Q_SB();
        Q_P(8, 1);
        Q_P(8, 2);
        Q_P(8, 3); load_a1(a_nxt, 3, 0); load_a1(a_nxt, 3, 1); Q_SB();
#undef Q_P
        rows_pass(t, 3, 6, vn);                              // 21 packed VALU (points 18 .. 35: every MFMA of this step has issued)
        Q_SB();
        a_cur = a_nxt;
        a_nxt = a_nn;
    };

    // ---- epilogue of one item: lane-local output transform (A^T M A on register pairs), bias / ReLU / skip.  Sub-block k = cbi * 4 + r
    //      is one output channel of this lane: 4 rows x 4 columns.
    auto sub_block = [&](int k, f32x4 (&Y)[4]) {
        const int cbi = k >> 2, r = k & 3;
        const float bb = lds[QOFF_B + cur_cg * QCO + (2 * hh + cbi) * 16 + 4 * kq + r];
        // A^T M on the column pairs (0,5), (1,2), (3,4): ma?[a] = rows of outputs; 3 x 10 packed VALU
        f32x2 ma0[4], ma1[4], ma2[4];
#define Q_M(I, J) acc[cbi][(I) * 6 + (J)][r]
#define Q_MP(I, J0, J1) f32x2{Q_M(I, J0), Q_M(I, J1)}
        at6p(Q_MP(0, 0, 5), Q_MP(1, 0, 5), Q_MP(2, 0, 5), Q_MP(3, 0, 5), Q_MP(4, 0, 5), Q_MP(5, 0, 5), ma0[0], ma0[1], ma0[2], ma0[3]);
        at6p(Q_MP(0, 1, 2), Q_MP(1, 1, 2), Q_MP(2, 1, 2), Q_MP(3, 1, 2), Q_MP(4, 1, 2), Q_MP(5, 1, 2), ma1[0], ma1[1], ma1[2], ma1[3]);
        at6p(Q_MP(0, 3, 4), Q_MP(1, 3, 4), Q_MP(2, 3, 4), Q_MP(3, 3, 4), Q_MP(4, 3, 4), Q_MP(5, 3, 4), ma2[0], ma2[1], ma2[2], ma2[3]);
#undef Q_MP
#undef Q_M
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            // (m0,m5), (m1,m2), (m3,m4) -> y0 = m0 + s1 + s2, y1 = d1 + 2 d2, y2 = s1 + 4 s2, y3 = d1 + 8 d2 + m5
            const f32x2 SD1 = pk_lo_pm_hi(ma1[a]), SD2 = pk_lo_pm_hi(ma2[a]);      // (s1, d1), (s2, d2)
            f32x2 y01 = fma2(SD2, f32x2{1.f, 2.f}, SD1), y23 = fma2(SD2, f32x2{4.f, 8.f}, SD1);
            y01 += f32x2{bb, bb};
            y23 += f32x2{bb, bb};
            Y[a] = f32x4{y01.x + ma0[a].x, y01.y, y23.x, y23.y + ma0[a].y};
        }
        if (p.relu) {                                        // block-uniform; the empty asm keeps it a branch (hipcc otherwise emits 16 selects)
            asm volatile("");
#pragma unroll
            for (int a = 0; a < 4; ++a) Y[a] = __builtin_elementwise_max(Y[a], f32x4{0.f, 0.f, 0.f, 0.f});
        }
    };
    auto row_sum16 = [&](float v) {      // sum over the 16 lanes of a DPP row (= the 16 tiles of this wave), valid in lane 15 of the row
#define Q_ROW_SHR(SH) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xf, 0xf, true))
        Q_ROW_SHR(1); Q_ROW_SHR(2); Q_ROW_SHR(4); Q_ROW_SHR(8);
#undef Q_ROW_SHR
        return v;
    };
    auto epilogue = [&]() {
        const auto ro = __builtin_amdgcn_make_buffer_rsrc(cur_out, 0, cur_nrec, 0x00020000);
        unsigned vo_cur[4];
        lane_offsets(vo_cur);
        if constexpr (MODE == W4_PS) {
            // sub-block pairs (r = 2 dy, 2 dy + 1) of a channel block are the two interleaved columns of HR row 2 (y0 + a) + dy:
            // 8 consecutive floats = two 16-byte stores per (a, dy)
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                const int cbi = kp >> 1, dy = kp & 1;
                f32x4 Y0[4], Y1[4];
                sub_block(4 * cbi + 2 * dy, Y0);
                sub_block(4 * cbi + 2 * dy + 1, Y1);
                const unsigned ko = (unsigned)(cbi * 4) * 4u * HWb + (unsigned)(dy * 2 * W) * 4u;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{Y0[a].x, Y1[a].x, Y0[a].y, Y1[a].y}), ro, vo_cur[a] + ko, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{Y0[a].z, Y1[a].z, Y0[a].w, Y1[a].w}), ro, vo_cur[a] + ko + 16u, 0, 0);
                }
            }
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cur_res), 0, cur_nrec, 0x00020000);
            // Skip-tensor loads carry the non-temporal hint (aux = 2): that stream is touched once per launch and should not displace
            // what the NEXT launch reads.  The output stores stay temporal: with the alternating item order (top of the kernel) the
            // next launch begins on the lines stored last.  Measured on config 2 (33 launches per step, one gpurun call, two rounds):
            // 9.21 ms/step with both hinted, 9.21 with neither, 9.11 with this split.  (Stand-alone on one layer with fixed buffers the
            // hinted stores look 8 % faster -- there nothing reads the output, and the burst of all 256 workgroups' simultaneous
            // epilogues, 4.2 MB per XCD against a 4 MB L2, is acknowledged sooner; every wave sits that acknowledgement out at its
            // first wait on a younger load: one in-order vmcnt, confirmed with an explicit vmcnt(0) behind the epilogue that costs
            // nothing.  The raw-row DMA stays temporal too: halo rows are re-read by the neighbouring tiles from L2.)
            f32x4 rr[3][4];                                  // skip-tensor rows are requested two sub-blocks ahead
            auto fetch = [&](int k) {
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        rr[k % 3][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_cur[a] + koff(k), 0, 2));
                }
            };
            fetch(0);
            fetch(1);
            float cs[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k + 2 < 8) fetch(k + 2);
                f32x4 Y[4];
                sub_block(k, Y);
                float ssum = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if constexpr (HAS_RES) {
                        if (MODE == W4_TRAIN && p.res_mask) {     // block-uniform
                            const f32x4 m = rr[k % 3][a];
                            Y[a] = f32x4{m.x > 0.f ? Y[a].x : 0.f, m.y > 0.f ? Y[a].y : 0.f, m.z > 0.f ? Y[a].z : 0.f, m.w > 0.f ? Y[a].w : 0.f};
                        } else {
                            Y[a] += rr[k % 3][a];
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Y[a]), ro, vo_cur[a] + koff(k), 0, 0);
                    if constexpr (MODE == W4_TRAIN) {
                        if (vo_cur[a] != OOB) ssum += (Y[a].x + Y[a].y) + (Y[a].z + Y[a].w);
                    }
                }
                if constexpr (MODE == W4_TRAIN) {
                    const int co = cur_cg * QCO + (2 * hh + (k >> 2)) * 16 + 4 * kq + (k & 3);
                    cs[k] = row_sum16(co < p.Cout ? ssum : 0.f);
                }
            }
            // per-channel sums of what this wave just stored (the bias gradient of the layer that consumes this tensor as its output
            // gradient): 16 tiles reduced in the DPP row, two float4 per (item, tile row, 4-channel group); fixed order -> run-to-run identical
            if constexpr (MODE == W4_TRAIN) {
                int ln = lane;
                asm volatile("" : "+v"(ln));                 // address from the lane id here (a lane-shaped pointer held across the steps gets spilled)
                if (csum && (ln & 15) == 15) {                // csum: block-uniform
                    float* o = csum + ((long long)cur_item * QTR + tg) * QCO + 2 * hh * 16 + 4 * (ln >> 4);
                    *reinterpret_cast<f32x4*>(o) = f32x4{cs[0], cs[1], cs[2], cs[3]};
                    *reinterpret_cast<f32x4*>(o + 16) = f32x4{cs[4], cs[5], cs[6], cs[7]};
                }
            }
        }
    };

    // ---- the stream
    int item = it0;
    for (int m = 0; m < n_my; ++m) {
        step(true);
        for (int s = 1; s < S; ++s) step(false);
        epilogue();
        item += istep;
        if (m + 1 < n_my) set_item(item);
    }
#undef Q_MFMA2
}
#undef Q_SB
#undef Q_FENCE

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv3x3_wino4_packed_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)ceil_div(Cout, QCO) * ceil_div(Cin, QCK) * QUF;
}

static int wino4_pack(const float* w, int32_t Cin, int32_t Cout, int tf, float* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "winograd-4x4 pack: bad argument");
    const int nsteps = ceil_div(Cin, QCK);
    const long long total = (long long)ceil_div(Cout, QCO) * nsteps * QUF;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    wino4_pack_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, tf, packed, nsteps, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv3x3_wino4_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, float* packed, void* stream) {
    return wino4_pack(w, Cin, Cout, 0, packed, stream);
}

// transpose_flip = 1: w is [Cout][Cin][3][3] of the forward layer; packs its data-gradient convolution (Cout -> Cin channels) -- same
// convention as kmsr_conv3x3_pack_weights_f32: Cin / Cout are w's own dimensions, the packed convolution's are swapped
extern "C" int kmsr_conv3x3_wino4_pack_weights_ex_f32(const float* w, int32_t Cin, int32_t Cout, int32_t transpose_flip, float* packed, void* stream) {
    if (transpose_flip) return wino4_pack(w, Cout, Cin, 1, packed, stream);
    return wino4_pack(w, Cin, Cout, 0, packed, stream);
}

extern "C" size_t kmsr_conv3x3_wino4_chan_sum_floats(const kmsr_conv3x3_desc* d) {
    if (!d || d->N <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return 0;
    return (size_t)ceil_div(d->W, 4 * QTC) * ceil_div(d->H, 4 * QTR) * ceil_div(d->Cout, QCO) * d->N * QTR * QCO;
}

extern "C" int kmsr_conv3x3_wino4_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                                          const float* residual, float* out, void* stream_) {
    return kmsr_conv3x3_wino4_fwd_ex_f32(d, x, packed_w, bias, residual, out, nullptr, stream_);
}

extern "C" int kmsr_conv3x3_wino4_fwd_ex_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                                             const float* residual, float* out, float* chan_sum, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->flags & ~(uint32_t)KMSR_CONV_FLAG_RELU_MASK) return fail(KMSR_E_BADARG, "unknown conv flags 0x%x", d->flags);
    if ((d->flags & KMSR_CONV_FLAG_RELU_MASK) && (!residual || d->relu || d->pixel_shuffle))
        return fail(KMSR_E_BADARG, "KMSR_CONV_FLAG_RELU_MASK needs the mask tensor in `residual` and no relu / pixel_shuffle");
    if (chan_sum && d->pixel_shuffle) return fail(KMSR_E_UNSUPPORTED, "channel sums are not produced together with pixel_shuffle");
    if (reinterpret_cast<uintptr_t>(chan_sum) % 16) return fail(KMSR_E_BADARG, "chan_sum must be 16-byte aligned");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "winograd conv needs stride 1");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (d->W % 4) return fail(KMSR_E_UNSUPPORTED, "winograd-4x4 conv needs W %% 4 == 0 (W=%d): rows move as aligned 16-byte pieces", d->W);
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(packed_w) % 16 || reinterpret_cast<uintptr_t>(out) % 16 ||
        reinterpret_cast<uintptr_t>(residual) % 16)
        return fail(KMSR_E_BADARG, "winograd-4x4 conv needs 16-byte aligned input / weights / output / residual");
    if (d->Cout > QMAX_BIAS) return fail(KMSR_E_UNSUPPORTED, "winograd-4x4 conv stages the bias vector in LDS: Cout <= %d (Cout=%d)", QMAX_BIAS, d->Cout);
    if ((long long)d->H * d->W * d->Cin * 4 >= (1ll << 31))   // 32-bit byte offsets inside one image of the input
        return fail(KMSR_E_UNSUPPORTED, "winograd-4x4 conv: one input image (Cin*H*W*4 = %lld bytes) exceeds 2 GB", (long long)d->H * d->W * d->Cin * 4);
    if ((long long)d->H * d->W * 4 * QCO >= (1ll << 31))      // 32-bit byte offsets inside one 64-channel output group
        return fail(KMSR_E_UNSUPPORTED, "winograd-4x4 conv: H*W = %lld exceeds 2^23 pixels per plane", (long long)d->H * d->W);
    Wino4Params p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, 4 * QTC);
    p.tiles = p.tiles_x * ceil_div(d->H, 4 * QTR);
    p.cgs = ceil_div(d->Cout, QCO);
    p.nsteps = ceil_div(d->Cin, QCK);
    p.relu = d->relu;
    p.res_mask = (d->flags & KMSR_CONV_FLAG_RELU_MASK) ? 1 : 0;
    if ((long long)p.cgs * p.nsteps * QUF * 4 >= (1ll << 32)) return fail(KMSR_E_UNSUPPORTED, "packed weights exceed 4 GB");
    const long long nitems = (long long)p.tiles * p.cgs * d->N;
    if (nitems > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_items = (int)nitems;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)std::min<long long>(nitems, cus);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->pixel_shuffle) conv3x3_wino4_kernel<false, W4_PS><<<grid, QT, 0, stream>>>(x, packed_w, bias, residual, out, nullptr, p);
    else if (chan_sum || p.res_mask) {
        if (residual) conv3x3_wino4_kernel<true, W4_TRAIN><<<grid, QT, 0, stream>>>(x, packed_w, bias, residual, out, chan_sum, p);
        else conv3x3_wino4_kernel<false, W4_TRAIN><<<grid, QT, 0, stream>>>(x, packed_w, bias, residual, out, chan_sum, p);
    } else if (residual) conv3x3_wino4_kernel<true, W4_PLAIN><<<grid, QT, 0, stream>>>(x, packed_w, bias, residual, out, nullptr, p);
    else conv3x3_wino4_kernel<false, W4_PLAIN><<<grid, QT, 0, stream>>>(x, packed_w, bias, residual, out, nullptr, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
