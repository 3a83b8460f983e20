// Winograd F(2x2,3x3) forward convolution on the fp32 matrix cores, "V in registers" form (semantics: include/kmsr.h;
// same math, epilogue and tolerances as conv3x3_wino.hip / conv3x3_wino_dma.hip).
//
// The LDS-DMA kernel (conv3x3_wino_dma.hip) owns a CU with one 8-wave workgroup and spends ~0.7 us of every 2.4 us chunk
// between two workgroup barriers writing the transformed inputs V to LDS while the matrix pipe idles; prologue and epilogue
// are exposed as well (profiles/r01_wino_dma_experiments.log: pipe busy 51 %).  This kernel is built so that TWO independent
// workgroups share a CU (<= 80 KB of LDS, 4 waves, 256 registers each): whatever one of them waits for -- a barrier, a DMA
// landing, its prologue or its epilogue -- the other one's waves keep the SIMD's matrix pipe busy.
//   * workgroup = 4 waves = 64 output channels x 32 tiles (2 output rows x 64 columns); wave i owns the transform row
//     xi = 4i .. 4i+3 for BOTH 32-channel cout blocks: 8 accumulator tiles (128 registers);
//   * the B operand of v_mfma_f32_32x32x2_f32 -- lane (tile t, half h) supplies V[xi][channel 2kp + h][tile t] -- is what that
//     lane can compute itself from the raw input rows in LDS: row i of B^T d needs only TWO patch rows (d0-d2, d1+d2, d2-d1,
//     d1-d3), so a patch costs 6 ds_read_b64 + 8 VALU, placed by hand in the shadow of the 64-cycle MFMAs of the previous
//     chunk.  V never goes through LDS: no V buffer, no V writes, no B-operand reads, no transform barrier;
//   * the A operands (pre-transformed weights U) of a wave are ITS OWN slice of U (xi row i): a wave-private LDS ring filled by
//     that wave's LDS-DMA (no cross-wave hand-over), packed so that a lane's four xi-column values are one ds_read_b128;
//   * only the raw input rows (4 rows x 72 columns per channel) are shared by the 4 waves: one 4-wave barrier per 4-channel
//     chunk, placed where every MFMA operand of the next step is already in registers;
//   * every wave issues the same 6 DMA instructions per chunk (4 of its U ring, 2 of the raw ring; tail chunks, padding lanes
//     and out-of-image pieces fetch a 16-byte zero page), so one counted `s_waitcnt vmcnt(8)` per chunk is exact;
//   * the skip tensor of a residual launch is staged by the same DMA slots: the U-ring loads that would fetch chunks nch and
//     nch+1 fetch the wave's 32 channels x 64 columns of the skip row instead (no registers, no extra instructions);
//   * the sign of xi row 2 (d2-d1 = -(d1-d2)) is folded into the packed weights, so every row transform is one fma(+-1).
// Epilogue: column transform in registers (M A), the 4 waves exchange their xi-row contributions through LDS (48 KB),
// wave i finalises output row i&1 of cout block i>>1: bias, ReLU, skip, PixelShuffle, stores.
// LDS: U rings 4 x 3 x 4 KB + raw ring 4 x 5 KB + 1 KB dump = 69 KB.  Requires W % 4 == 0 and 16-byte aligned tensors.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int VCK = 4;                 // input channels per chunk
constexpr int VCO = 64;                // output channels per workgroup
constexpr int TCOLS = 32;              // tiles per workgroup: 1 x 32  (output 2 rows x 64 cols)
constexpr int kThreads = 256;
constexpr int UWF = 4 * VCK * 64;      // 1024 floats = 4 KB: one wave's slice of U per chunk ([q = cbk*2 + kp][lane][j])
constexpr int VUF = 4 * UWF;           // 4096 floats of U per chunk and cout group
constexpr int NU = 3;                  // U ring slots per wave (a chunk's U is requested 2 chunks ahead)
constexpr int UINSTR = 4;              // U DMA instructions per wave and chunk
constexpr int RROWS = 4;               // input rows of a tile row
constexpr int RQ = (2 * TCOLS + 8) / 4;            // 18 float4 per row segment: columns tx0-4 .. tx0+67
constexpr int RCOLS = 4 * RQ;                      // 72
constexpr int RLANES = VCK * RROWS * RQ;           // 288 lane-loads per chunk = 4.5 wave instructions
constexpr int RINSTR = 2;                          // raw DMA instructions per wave and chunk (8 slots, 5 real, 3 to the dump)
constexpr int RREAL = (RLANES + 63) / 64;          // 5
constexpr int RAWF = RREAL * 256;                  // 1280 floats = 5 KB per ring slot
constexpr int NRAW = 4;                            // raw ring slots: rows are requested 4 chunks ahead of their transform
constexpr int OFF_U = 0, OFF_R = 4 * NU * UWF, OFF_DUMP = OFF_R + NRAW * RAWF, LOOP_FLOATS = OFF_DUMP + 256;
constexpr int XCHF = 4 * 2 * 16 * 64 * 2;          // epilogue exchange: [wave][cout block][reg][lane] float2 = 64 KB
constexpr int OFF_B = LOOP_FLOATS > XCHF ? LOOP_FLOATS : XCHF;
constexpr int LDS_FLOATS = OFF_B + VCO;
static_assert(LDS_FLOATS * 4 <= 80 * 1024, "two workgroups must fit one CU's 160 KB of LDS");

__device__ __attribute__((aligned(16))) float g_zero_page[4];   // zero-initialised: source of every padded 16-byte piece

struct WinoVParams {
    int N, Cin, H, W, Cout;            // stride 1: Ho = H, Wo = W
    int tiles_x, tiles, cgs, nchunks, relu, n_blocks;
    int stagger_mode, stagger_sleeps;  // EXPERIMENT (to be hard-wired): which first-generation workgroups start late, and by how much
};

// packed[cg][chunk][i][q = cbk*2 + kp][lane = half*32 + l32][j] = s_i * (G g G^T)[xi = 4i + j] for
// (co, ci) = (cg*64 + cbk*32 + l32, chunk*4 + 2kp + half), zero padded; s_2 = -1 (see the header), else +1
__global__ void winov_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ packed, int nchunks,
                                  long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const int q = (int)((idx >> 8) & 3);
        const int i = (int)((idx >> 10) & 3);
        const int chunk = (int)((idx >> 12) % nchunks);
        const int cg = (int)((idx >> 12) / nchunks);
        const int half = lane >> 5, l32 = lane & 31, cbk = q >> 1, kp = q & 1;
        const int cog = cg * VCO + cbk * 32 + l32, cig = chunk * VCK + 2 * kp + half;
        float v = 0.f;
        if (cog < Cout && cig < Cin) {
            const float* g = w + ((long long)cog * Cin + cig) * 9;
            // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  U = G g G^T
            float row[3];   // (G g)[i][:]
            for (int c = 0; c < 3; ++c) {
                const float g0 = g[0 * 3 + c], g1 = g[1 * 3 + c], g2 = g[2 * 3 + c];
                row[c] = i == 0 ? g0 : (i == 1 ? 0.5f * (g0 + g1 + g2) : (i == 2 ? 0.5f * (g0 - g1 + g2) : g2));
            }
            v = j == 0 ? row[0] : (j == 1 ? 0.5f * (row[0] + row[1] + row[2]) : (j == 2 ? 0.5f * (row[0] - row[1] + row[2]) : row[2]));
            if (i == 2) v = -v;
        }
        packed[idx] = v;
    }
}

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]).  The builtin, not inline asm:
// hipcc then KNOWS the counter state and does not re-wait conservatively for LDS reads that were already retired.
constexpr int wait_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }
#define KMSR_FENCE() asm volatile("" ::: "memory")
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

#define KMSR_SB() __builtin_amdgcn_sched_barrier(0)

template <bool PS>
__global__ void __launch_bounds__(kThreads, 2) conv3x3_winov_kernel(const float* __restrict__ xg,
                                                                    const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ res,
                                                                    float* __restrict__ out, WinoVParams p) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int xr = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave = xi row

    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * 2, tx0 = (tile % p.tiles_x) * (2 * TCOLS);
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const int nch = p.nchunks;
    // this wave's slice of the packed weights: chunk k at wme + k * VUF
    const float* __restrict__ wme = wpk + (long long)cg * nch * VUF + xr * UWF + lane * 4;
    const float* zero_page = g_zero_page;
    asm volatile("" : "+s"(zero_page));                      // resolved once (the GOT load would otherwise be repeated per chunk)

    {   // EXPERIMENT: de-phase the two workgroups that share a CU (equal lifetimes keep whatever phase they start with)
        bool late = false;
        const int b = blockIdx.x;
        if (p.stagger_mode == 1) late = b >= 256 && b < 512;
        else if (p.stagger_mode == 2) late = b < 512 && ((b >> 3) & 1);
        else if (p.stagger_mode == 3) late = b < 512 && ((b >> 8) & 1) == 0 && ((b >> 3) & 1);
        else if (p.stagger_mode == 4) late = b >= 256 && b < 512;
        if (late)
            for (int i = 0; i < p.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(32);
        if (p.stagger_mode == 4 && b < 256)                  // mode 4: random-ish phase for everyone in the first generation
            for (int i = 0; i < (b * 7) % (2 * p.stagger_sleeps + 1); ++i) __builtin_amdgcn_s_sleep(16);
    }
    // this wave finalises output row xr & 1 of cout block xr >> 1
    const int orow = xr & 1, ocb = xr >> 1;
    const int ry = ty0 + orow, rx0 = tx0 + 2 * l32;

    // ---- DMA maps.  Raw tile (shared), lane-linear: L = (2 xr + i) * 64 + lane  <-> (segment = L / 18 = channel * 4 + row,
    //      q = L % 18 = float4 within the row); LDS offset = 4 L floats; instruction slots 5..7 are padding and go to the dump.
    int roff[RINSTR], rch[RINSTR];
#pragma unroll
    for (int i = 0; i < RINSTR; ++i) {
        const int L = (RINSTR * xr + i) * 64 + lane;
        const int seg = L / RQ, q = L - seg * RQ, ch = seg / RROWS, row = seg - ch * RROWS;
        const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * q;
        const bool ok = L < RLANES && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
        roff[i] = ok ? (int)(ch * HW + (long long)gy * W + gx) : -1;
        rch[i] = ch;
    }
    auto issue_raw = [&](int chunk) {                        // raw(chunk) -> ring slot chunk % NRAW; past the last chunk: zeros
        const float* __restrict__ xc = xn + (long long)chunk * VCK * HW;
        const int cin_left = p.Cin - chunk * VCK;            // <= 0 past the last chunk
        float* rb = lds + OFF_R + (chunk % NRAW) * RAWF;
#pragma unroll
        for (int i = 0; i < RINSTR; ++i) {
            const bool ok = roff[i] >= 0 && rch[i] < cin_left;
            const float* src = ok ? xc + roff[i] : zero_page;
            const int slot = RINSTR * xr + i;                // wave-uniform
            float* dst = slot < RREAL ? rb + slot * 256 : lds + OFF_DUMP;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
        }
    };
    // U ring slot k % NU of this wave <- chunk k of its U slice; k = nch, nch + 1: the two halves of the wave's skip-tensor slice
    // ([channel 32][column 64] of output row ry, lane = (channel 4 i + lane / 16, columns 4 (lane % 16) ..+3)); beyond: zeros
    float* uring = lds + OFF_U + xr * (NU * UWF);
    const int res_c0 = cg * VCO + ocb * 32 + (lane >> 4);    // + 16 half + 4 i
    const int res_x = tx0 + 4 * (lane & 15);
    const bool res_ok = !PS && res != nullptr && ry < H && res_x < W;
    const float* res_me = res_ok ? res + (((long long)n * p.Cout + res_c0) * H + ry) * W + res_x : zero_page;   // + (16 half + 4 i) * HW
    auto issue_u = [&](int k) {
        float* ub = uring + (k % NU) * UWF;
        if (k < nch) {                                       // wave-uniform
            const float* __restrict__ wk = wme + (long long)k * VUF;
#pragma unroll
            for (int i = 0; i < UINSTR; ++i)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(wk + i * 256), (lds_void_t*)(ub + i * 256), 16, 0, 0);
        } else if (!PS && res != nullptr && k < nch + 2) {
            const int c16 = 16 * (k - nch);
#pragma unroll
            for (int i = 0; i < UINSTR; ++i) {
                const bool ok = res_ok && res_c0 + c16 + 4 * i < p.Cout;
                const float* src = ok ? res_me + (long long)(c16 + 4 * i) * HW : zero_page;
                __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(ub + i * 256), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < UINSTR; ++i)
                __builtin_amdgcn_global_load_lds((glb_void_t*)zero_page, (lds_void_t*)(ub + i * 256), 16, 0, 0);
        }
    };

    // ---- input transform of this wave's xi row: lane (tile l32, half) owns channels 2kp + half, kp = 0, 1.
    //      patch rows r <-> raw rows r; patch columns <-> raw columns 2 l32 + 3 .. + 6, read as the aligned pairs at raw columns
    //      2 l32 + 2, + 4, + 6.  Row transform t = d[ra] + sb * d[rb]:
    //      i = 0: d0 - d2;  i = 1: d1 + d2;  i = 2: d1 - d2 (sign folded into U);  i = 3: d1 - d3.
    const int ra = xr == 0 ? 0 : 1, rbw = xr == 3 ? 3 : 2;
    const float sb = xr == 1 ? 1.f : -1.f;
    const int rbase = (half * RROWS + ra) * RCOLS + 2 * l32 + 2;
    const int rdelta = (rbw - ra) * RCOLS;
    struct Patch { f32x2 a0, a1, a2, b0, b1, b2; };
    auto load_patch = [&](const float* rp, int kp) {         // 6 ds_read_b64
        const float* pa = rp + 2 * kp * RROWS * RCOLS;
        const float* pb = pa + rdelta;
        Patch q;
        q.a0 = *reinterpret_cast<const f32x2*>(pa); q.a1 = *reinterpret_cast<const f32x2*>(pa + 2); q.a2 = *reinterpret_cast<const f32x2*>(pa + 4);
        q.b0 = *reinterpret_cast<const f32x2*>(pb); q.b1 = *reinterpret_cast<const f32x2*>(pb + 2); q.b2 = *reinterpret_cast<const f32x2*>(pb + 4);
        return q;
    };
    auto rows_of = [&](const Patch& q) {                     // 4 fma: t = d[ra] + sb d[rb]
        return f32x4{fmaf(sb, q.b0.y, q.a0.y), fmaf(sb, q.b1.x, q.a1.x), fmaf(sb, q.b1.y, q.a1.y), fmaf(sb, q.b2.x, q.a2.x)};
    };
    auto cols_of = [&](const f32x4& t, float (&v)[4]) {      // 4 add: V = t B
        v[0] = t.x - t.z;
        v[1] = t.y + t.z;
        v[2] = t.z - t.y;
        v[3] = t.y - t.w;
    };
    auto raw_slot = [&](int chunk) { return lds + OFF_R + (chunk % NRAW) * RAWF + rbase; };
    auto u_slot = [&](int chunk) { return uring + (chunk % NU) * UWF + lane * 4; };

    f32x16 acc[2][4];      // [cout block][xi column j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;

    // ---- prologue.  In flight, oldest first: raw(0) raw(1) U(0) raw(2) U(1) raw(3); V(0) needs raw(0), the first MFMAs U(0).
    issue_raw(0);
    issue_raw(1);
    issue_u(0);
    issue_raw(2);
    issue_u(1);
    issue_raw(3);
    wait_dma<3 * RINSTR + 2 * UINSTR>();                     // raw(0) has landed (this wave's part)
    barrier_lds();
    float vA[2][4], vB[2][4];
    {
        const float* rp = raw_slot(0);
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) cols_of(rows_of(load_patch(rp, kp)), vA[kp]);
    }
    wait_dma<2 * RINSTR + UINSTR>();                         // raw(1), U(0) have landed
    f32x4 a0[2], a1[2];                                      // A operands of the current / next step ([q = cbk*2 + kp])
    a0[0] = *reinterpret_cast<const f32x4*>(u_slot(0));
    a1[0] = *reinterpret_cast<const f32x4*>(u_slot(0) + 2 * 256);

    // One chunk = 2 steps (channel pairs kp) of 8 MFMAs, hand-placed and pinned with sched_barrier:
    //   barrier (raw(c+1) of every wave has landed, raw slot c % 4 is free)
    //   step 0: request patch 0 of chunk c+1 and the A operands / patch of step 1; 8 MFMAs with V(c+1)[0] computed between them;
    //           this chunk's 6 DMA instructions; counted wait for U(c+1) / raw(c+2)
    //   step 1: request the A operands of chunk c+1's step 0; 8 MFMAs with V(c+1)[1] computed between them.
    // The first MFMAs after the barrier have their operands in registers already.
    auto step = [&](const f32x4& A0, const f32x4& A1, const float (&v)[4], const Patch& pt, float (&vnext)[4]) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[0], v[0], acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[0], v[0], acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[1], v[1], acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[1], v[1], acc[1][1], 0, 0, 0);
        KMSR_SB();
        f32x4 t = rows_of(pt);
        asm volatile("" : "+v"(t));                          // keep the VALU work HERE (LLVM would sink it to its use in the next chunk)
        KMSR_SB();
        acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[2], v[2], acc[0][2], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[2], v[2], acc[1][2], 0, 0, 0);
        KMSR_SB();
        cols_of(t, vnext);
        asm volatile("" : "+v"(vnext[0]), "+v"(vnext[1]), "+v"(vnext[2]), "+v"(vnext[3]));
        KMSR_SB();
        acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[3], v[3], acc[0][3], 0, 0, 0);
        acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[3], v[3], acc[1][3], 0, 0, 0);
        KMSR_SB();
    };
    auto phase = [&](int chunk, const float (&vc)[2][4], float (&vn)[2][4]) {
        barrier_lds();
        const float* rp = raw_slot(chunk + 1);               // (past the last chunk: zeros, harmless)
        const float* uc = u_slot(chunk);
        const Patch p0 = load_patch(rp, 0);
        a0[1] = *reinterpret_cast<const f32x4*>(uc + 1 * 256);
        a1[1] = *reinterpret_cast<const f32x4*>(uc + 3 * 256);
        const Patch p1 = load_patch(rp, 1);
        KMSR_SB();
        step(a0[0], a1[0], vc[0], p0, vn[0]);
        issue_u(chunk + 2);
        issue_raw(chunk + 4);
        // outstanding, oldest first: raw(c+2) U(c+1) raw(c+3) U(c+2) raw(c+4): the first two must have landed
        wait_dma<2 * RINSTR + UINSTR>();
        KMSR_SB();
        const float* un = u_slot(chunk + 1);                 // wave-private ring: no barrier needed for U
        a0[0] = *reinterpret_cast<const f32x4*>(un);
        a1[0] = *reinterpret_cast<const f32x4*>(un + 2 * 256);
        KMSR_SB();
        step(a0[1], a1[1], vc[1], p1, vn[1]);
    };
    for (int chunk = 0; chunk < nch; chunk += 2) {
        phase(chunk, vA, vB);
        if (chunk + 1 < nch) phase(chunk + 1, vB, vA);
    }

    // ---- epilogue.  Column transform of this wave's xi row: MA[b] = sum_j M[j] A[j][b], A^T = [[1,1,1,0],[0,1,-1,-1]].
    //      Row transform across the 4 waves: Y[0][b] = MA_0 + MA_1 + MA_2, Y[1][b] = MA_1 - MA_2 - MA_3
    //      (xi row 2: U and V both carry the folded sign, so the accumulators hold M itself).
    f32x2 ma[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float m0 = acc[cb][0][reg], m1 = acc[cb][1][reg], m2 = acc[cb][2][reg], m3 = acc[cb][3][reg];
            ma[cb][reg] = f32x2{m0 + m1 + m2, m1 - m2 - m3};
        }
    wait_dma<0>();                                           // every DMA of this wave has landed (incl. the skip-tensor halves)
    // skip values of this wave's outputs, [4g + r] <-> channel co0(g) + r, pixel pair (ry, rx0 .. rx0+1), from its U ring
    f32x2 rres[16];
    const bool has_res = !PS && res != nullptr;              // block-uniform
    if (has_res) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c32 = 8 * g + 4 * half + r;        // channel within the wave's 32; half (c32 >> 4) sits in ring slot (nch + half) % NU
                const float* slot = uring + ((nch + (g >> 1)) % NU) * UWF;
                rres[4 * g + r] = *reinterpret_cast<const f32x2*>(slot + (c32 & 15) * 64 + 2 * l32);
            }
    }
    barrier_lds();                                           // every wave is done with the rings: the exchange buffer may overwrite them
    f32x2* xch = reinterpret_cast<f32x2*>(lds);              // [wave][cout block][reg][lane]
    float* sbias = lds + OFF_B;
    if (tid < VCO) {
        const int co = cg * VCO + tid;
        sbias[tid] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    f32x2 keep[16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            if (cb == ocb) keep[reg] = ma[cb][reg];          // wave-uniform select
            // (xi rows 0 and 3 enter only one output row: their own cout block's part is needed by nobody else)
            if (!(cb == ocb && (xr == 0 || xr == 3))) xch[((xr * 2 + cb) * 16 + reg) * 64 + lane] = ma[cb][reg];
        }
    }
    __syncthreads();
    // Y[orow] = sum over xi rows i' of AT[orow][i'] * MA_i':  row 0: +0 +1 +2;  row 1: +1 -2 -3.  Own contribution from registers.
    int pa_, pb_;
    float ca, cb_, cs;                                       // coefficients of partner a, partner b, self
    if (orow == 0) {          // waves xr = 0 (partners 1, 2) and xr = 2 (partners 0, 1): all +1
        pa_ = xr == 0 ? 1 : 0;
        pb_ = xr == 0 ? 2 : 1;
        ca = 1.f; cb_ = 1.f; cs = 1.f;
    } else {                  // waves xr = 1 (self +1, partners 2: -1, 3: -1) and xr = 3 (self -1, partners 1: +1, 2: -1)
        pa_ = xr == 1 ? 2 : 1;
        pb_ = xr == 1 ? 3 : 2;
        ca = xr == 1 ? -1.f : 1.f; cb_ = -1.f; cs = xr == 1 ? 1.f : -1.f;
    }
    const f32x2* xa = xch + ((pa_ * 2 + ocb) * 16) * 64 + lane;
    const f32x2* xb = xch + ((pb_ * 2 + ocb) * 16) * 64 + lane;
    const int y = ry, x0 = rx0;
    const bool pok2 = y < H && x0 < W;                        // (x0, x0+1) is an aligned pair: inside or outside as a whole
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co0 = cg * VCO + ocb * 32 + 8 * g + 4 * half;              // channels co0 .. co0+3 <-> registers 4g .. 4g+3
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + ocb * 32 + 8 * g + 4 * half);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        f32x2 Y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 oa = xa[(4 * g + r) * 64], ob = xb[(4 * g + r) * 64];
            Y[r] = f32x2{cs * keep[4 * g + r].x + ca * oa.x + cb_ * ob.x + bb[r], cs * keep[4 * g + r].y + ca * oa.y + cb_ * ob.y + bb[r]};
            if (p.relu) Y[r] = f32x2{fmaxf(Y[r].x, 0.f), fmaxf(Y[r].y, 0.f)};
        }
        if constexpr (PS) {
            // conv channel q = co0 + r -> out[n, q>>2, 2y + (r>>1), 2x + (r&1)]: this lane writes rows 2y, 2y+1, cols 2x0 .. 2x0+3
            if (pok2 && co0 < p.Cout) {
                const long long W2 = 2ll * W;
                float* o = out + (((long long)n * (p.Cout >> 2) + (co0 >> 2)) * (2ll * H) + 2 * y) * W2 + 2 * x0;
                *reinterpret_cast<f32x4*>(o) = f32x4{Y[0].x, Y[1].x, Y[0].y, Y[1].y};
                *reinterpret_cast<f32x4*>(o + W2) = f32x4{Y[2].x, Y[3].x, Y[2].y, Y[3].y};
            }
        } else {
            const long long base = ((long long)n * p.Cout + co0) * HW + (long long)y * W + x0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 v = Y[r];
                if (has_res) v = f32x2{v.x + rres[4 * g + r].x, v.y + rres[4 * g + r].y};   // out-of-range pieces were staged as zeros
                if (pok2 && co0 + r < p.Cout) *reinterpret_cast<f32x2*>(out + base + r * HW) = v;
            }
        }
    }
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

extern "C" int kmsr_conv3x3_winov_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                                          const float* bias, const float* residual, float* out, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "winograd conv needs stride 1");
    if (d->W % 4) return fail(KMSR_E_UNSUPPORTED, "winograd-vreg conv needs W %% 4 == 0 (W=%d): rows are fetched as aligned 16-byte pieces", d->W);
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(packed_w) % 16 || reinterpret_cast<uintptr_t>(out) % 8 ||
        reinterpret_cast<uintptr_t>(residual) % 16)
        return fail(KMSR_E_BADARG, "winograd-vreg conv needs 16-byte aligned input / weights / residual and 8-byte aligned output");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if ((long long)d->H * d->W * VCK >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "image plane exceeds 2^29 elements");
    WinoVParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, 2 * TCOLS);
    p.tiles = p.tiles_x * ceil_div(d->H, 2);
    p.cgs = ceil_div(d->Cout, VCO);
    p.nchunks = ceil_div(d->Cin, VCK);
    p.relu = d->relu;
    const long long nblocks = (long long)p.tiles * p.cgs * d->N;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_blocks = (int)nblocks;
    p.stagger_mode = (int)(d->flags & 0xF);
    p.stagger_sleeps = (int)((d->flags >> 4) & 0xFF);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->pixel_shuffle)
        conv3x3_winov_kernel<true><<<p.n_blocks, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else
        conv3x3_winov_kernel<false><<<p.n_blocks, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
