// Winograd F(2x2,3x3) forward convolution on the fp32 matrix cores, "V in registers" form (semantics: include/kmsr.h;
// same math, epilogue and tolerances as conv3x3_wino.hip / conv3x3_wino_dma.hip).
//
// The LDS-DMA kernel (conv3x3_wino_dma.hip) spends ~0.7 us of every 2.4 us chunk between two workgroup barriers writing the
// transformed inputs V to LDS while the matrix pipe idles (profiles/r01_wino_dma_experiments.log: pipe busy 51 %).  Here the
// workgroup's 8 waves are (xi row i = 0..3) x (tile row tb = 0,1) instead of (cout block, tile row, xi half):
//   * a wave multiplies its 4 transform points xi = 4i .. 4i+3 for BOTH 32-channel cout blocks and its 32 tiles: 8 accumulator
//     tiles (128 registers) as before, but the B operand of v_mfma_f32_32x32x2_f32 -- lane (tile t, half h) supplies
//     V[xi][channel 2kp + h][tile t] -- is exactly what that lane can compute itself from the raw input rows in LDS: row i of
//     B^T d needs only TWO patch rows (d0-d2, d1+d2, d2-d1, d1-d3), so a lane's 4 patches of an 8-channel chunk cost
//     24 ds_read_b64 + 32 VALU and no redundancy across waves.  V never goes through LDS: no V buffer, no V writes, no B-operand
//     reads, and -- the point -- no barrier between "transform" and "multiply": the next chunk's V is computed into a second
//     register set inside the current chunk's MFMA phase (VALU / LDS work in the shadow of the 64-cycle MFMAs).
//   * ONE workgroup barrier per 8-channel chunk (hand-over of the LDS-DMA buffers), none around the transform.
//   * the A operands (pre-transformed weights U) are packed so that a lane's four xi-column values are one 16-byte LDS word:
//     8 ds_read_b128 per chunk and wave instead of 32 ds_read_b32.
//   * zero padding / channel tails: lanes of the raw-row DMA that fall outside the image (or past Cin) fetch from a 16-byte
//     zero page in device memory, so the LDS tile already holds the padded patch and the transform needs no masks.
//   * the sign of xi row 2 (d2-d1 = -(d1-d2)) is folded into the packed weights, so every row transform is one fma(+-1).
// Epilogue: column transform in registers (M A), the 4 waves of a tile row exchange their xi-row contributions through LDS
// (128 KB, aliasing U / raw), wave (i, tb) finalises output row i&1 of cout block i>>1: bias, ReLU, skip, PixelShuffle, stores.
// LDS: U 2 x 32 KB + raw 3 x 16 KB = 112 KB in the k-loop.  Requires W % 4 == 0 and 16-byte aligned tensors.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int VCK = 8;                 // input channels per chunk
constexpr int VCO = 64;                // output channels per workgroup
constexpr int TROWS = 2, TCOLS = 32;   // tiles per workgroup: 2 x 32  (output 4 rows x 64 cols)
constexpr int VUF = 16 * VCK * VCO;    // 8192 floats = 32 KB of U per chunk
constexpr int RROWS = 2 * TROWS + 2;   // 6 input rows
constexpr int RQ = (2 * TCOLS + 8) / 4;            // 18 float4 per row segment: columns tx0-4 .. tx0+67
constexpr int RCOLS = 4 * RQ;                      // 72
constexpr int RLANES = VCK * RROWS * RQ;           // 864 lane-loads per chunk
constexpr int kThreads = 512;
constexpr int RINSTR = 2;                          // raw DMA instructions per wave and chunk (8 waves x 2 x 64 = 1024 >= 864)
constexpr int RAWF = 8 * RINSTR * 64 * 4;          // 4096 floats = 16 KB
constexpr int UINSTR = VUF / 4 / kThreads;         // 4 U DMA instructions per wave and chunk
constexpr int NRAW = 3;                            // raw buffers: rows are fetched 3 chunks ahead of their MFMAs (2 ahead of their transform)
constexpr int OFF_U = 0, OFF_R = 2 * VUF;
constexpr int XCHF = 8 * 2 * 16 * 64 * 2;          // epilogue exchange: [wave][cout block][reg][lane] float2 = 128 KB
constexpr int OFF_B = XCHF;                        // bias row behind the exchange buffer
constexpr int LDS_FLOATS = OFF_B + VCO;
static_assert(OFF_R + NRAW * RAWF <= XCHF, "k-loop buffers must fit under the bias row");

__device__ __attribute__((aligned(16))) float g_zero_page[4];   // zero-initialised: source of every padded 16-byte piece

struct WinoVParams {
    int N, Cin, H, W, Cout;            // stride 1: Ho = H, Wo = W
    int tiles_x, tiles, cgs, nchunks, relu, n_blocks;
};

// packed[cg][chunk][i][q = cbk*4 + kp][lane = half*32 + l32][j] = s_i * (G g G^T)[xi = 4i + j] for
// (co, ci) = (cg*64 + cbk*32 + l32, chunk*8 + 2kp + half), zero padded; s_2 = -1 (see the header), else +1
__global__ void winov_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ packed, int nchunks,
                                  long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const int q = (int)((idx >> 8) & 7);
        const int i = (int)((idx >> 11) & 3);
        const int chunk = (int)((idx >> 13) % nchunks);
        const int cg = (int)((idx >> 13) / nchunks);
        const int half = lane >> 5, l32 = lane & 31, cbk = q >> 2, kp = q & 3;
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

__device__ __forceinline__ void barrier_lds() {   // LDS hand-over only: does NOT wait for LDS-DMA in flight (counted by the caller)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool PS>
__global__ void __launch_bounds__(kThreads, 2) conv3x3_winov_kernel(const float* __restrict__ xg,
                                                                    const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ res,
                                                                    float* __restrict__ out, WinoVParams p) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xr = wv & 3, tb = wv >> 2;              // xi row of this wave, tile row of this wave

    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * (2 * TROWS), tx0 = (tile % p.tiles_x) * (2 * TCOLS);
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const float* __restrict__ wcg = wpk + (long long)cg * p.nchunks * VUF;
    const int nch = p.nchunks;

    // ---- LDS-DMA maps (the same for every chunk).  Raw tile, lane-linear: L = (2 wv + i) * 64 + lane  <->
    //      (segment = L / 18 = channel * 6 + row, q = L % 18 = float4 within the row); LDS offset = 4 L floats.
    //      Every wave issues EXACTLY RINSTR raw and UINSTR weight DMA instructions per chunk (padding lanes read the zero
    //      page), so `s_waitcnt vmcnt(RINSTR)` means "everything except the newest raw chunk has landed".
    const float* zero_page = g_zero_page;
    asm volatile("" : "+s"(zero_page));                      // resolved once (the GOT load would otherwise be repeated per chunk)
    int roff[RINSTR], rch[RINSTR];
#pragma unroll
    for (int i = 0; i < RINSTR; ++i) {
        const int L = (RINSTR * wv + i) * 64 + lane;
        const int seg = L / RQ, q = L - seg * RQ, ch = seg / RROWS, row = seg - ch * RROWS;
        const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * q;
        const bool ok = L < RLANES && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
        roff[i] = ok ? (int)(ch * HW + (long long)gy * W + gx) : -1;
        rch[i] = ch;
    }
    auto issue_raw = [&](int chunk) {
        const float* __restrict__ xc = xn + (long long)chunk * VCK * HW;
        const int cin_left = p.Cin - chunk * VCK;
        float* rb = lds + OFF_R + (chunk % NRAW) * RAWF;
#pragma unroll
        for (int i = 0; i < RINSTR; ++i) {
            const bool ok = roff[i] >= 0 && rch[i] < cin_left;
            const float* src = ok ? xc + roff[i] : zero_page;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(rb + (RINSTR * wv + i) * 256), 16, 0, 0);
        }
    };
    auto issue_u = [&](int chunk) {
        const float* __restrict__ wc = wcg + (long long)chunk * VUF;
        float* ub = lds + OFF_U + (chunk & 1) * VUF;
#pragma unroll
        for (int i = 0; i < UINSTR; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wc + ((UINSTR * wv + i) * 64 + lane) * 4),
                                             (lds_void_t*)(ub + (UINSTR * wv + i) * 256), 16, 0, 0);
    };

    // ---- input transform of this wave's xi row: lane (tile l32, half) owns channels 2kp + half, kp = 0..3.
    //      patch rows r <-> raw rows 2 tb + r; patch columns <-> raw columns 2 l32 + 3 .. + 6, read as the aligned pairs at
    //      raw columns 2 l32 + 2, + 4, + 6.  Row transform t = d[ra] + sb * d[rb]:
    //      i = 0: d0 - d2;  i = 1: d1 + d2;  i = 2: d1 - d2 (sign folded into U);  i = 3: d1 - d3.
    const int ra = xr == 0 ? 0 : 1, rbw = xr == 3 ? 3 : 2;
    const float sb = xr == 1 ? 1.f : -1.f;
    const int rbase = (half * RROWS + 2 * tb + ra) * RCOLS + 2 * l32 + 2;
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
    auto transform_now = [&](int chunk, float (&v)[4][4]) {  // prologue only (nothing to overlap with)
        const float* rp = lds + OFF_R + (chunk % NRAW) * RAWF + rbase;
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) cols_of(rows_of(load_patch(rp, kp)), v[kp]);
    };

    f32x16 acc[2][4];      // [cout block][xi column j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;

    // this wave finalises output row xr & 1 of cout block xr >> 1 of its tile row
    const int orow = xr & 1, ocb = xr >> 1;
    const int ry = ty0 + 2 * tb + orow, rx0 = tx0 + 2 * l32;
    // Its slice of the skip tensor (32 channels x 64 columns of row ry, 8 KB) is staged by LDS-DMA during the LAST chunk's MFMA
    // phase into LDS that is free by then -- the raw buffers (waves 0..5) and the idle U buffer (waves 6, 7) -- as
    // [channel 32][column 64]; 8 instructions per wave, lane = (channel 4k + lane/16, columns 4 (lane%16) ..+3).
    float* rstage = wv < 6 ? lds + OFF_R + wv * 2048 : lds + OFF_U + (nch & 1) * VUF + (wv - 6) * 2048;
    auto issue_residual = [&]() {
        if constexpr (!PS) {
            if (res) {                                       // block-uniform
                const int cl = lane >> 4, xq = tx0 + 4 * (lane & 15);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int co = cg * VCO + ocb * 32 + 4 * k + cl;
                    const bool ok = ry < H && xq < W && co < p.Cout;
                    const float* src = ok ? res + (((long long)n * p.Cout + co) * H + ry) * W + xq : zero_page;
                    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(rstage + k * 256), 16, 0, 0);
                }
            }
        }
    };

    // One chunk: hand-over barrier, then 4 steps (channel pairs kp) of 8 MFMAs.  Hand-placed software pipeline, pinned with
    // sched_barrier: step kp requests the A operands and the raw patch of step kp+1 from LDS, and computes the next chunk's
    // V[kp] (4 fma + 4 add) between its MFMAs, so the VALU / LDS work runs in the shadow of the 64-cycle MFMAs.
    // The DMA for later chunks is issued after the first step's MFMAs (anywhere after the barrier is legal).
#define KMSR_SB() __builtin_amdgcn_sched_barrier(0)
    auto phase = [&](int chunk, const float (&vc)[4][4], float (&vn)[4][4]) {
        // outstanding DMA groups, oldest first: raw(chunk+1), U(chunk), raw(chunk+2): all but the newest must have landed
        if (chunk + 2 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RINSTR) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier_lds();                                       // every wave is past chunk-1: U[(chunk+1)&1] and raw[chunk%3] are free
        const float* ub = lds + OFF_U + (chunk & 1) * VUF + xr * (8 * 256) + lane * 4;
        const float* rp = lds + OFF_R + ((chunk + 1 < nch ? chunk + 1 : chunk) % NRAW) * RAWF + rbase;   // (last chunk: harmless re-transform)
        f32x4 a0[2], a1[2];
        Patch pt[2];
        a0[0] = *reinterpret_cast<const f32x4*>(ub);
        a1[0] = *reinterpret_cast<const f32x4*>(ub + 4 * 256);
        pt[0] = load_patch(rp, 0);
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            const int cur = kp & 1, nxt = cur ^ 1;
            if (kp < 3) {
                a0[nxt] = *reinterpret_cast<const f32x4*>(ub + (kp + 1) * 256);
                a1[nxt] = *reinterpret_cast<const f32x4*>(ub + (4 + kp + 1) * 256);
                pt[nxt] = load_patch(rp, kp + 1);
            }
            KMSR_SB();
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur][0], vc[kp][0], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur][0], vc[kp][0], acc[1][0], 0, 0, 0);
            KMSR_SB();
            f32x4 t = rows_of(pt[cur]);
            asm volatile("" : "+v"(t));                      // keep the VALU work HERE (LLVM would sink it to its use in the next phase)
            KMSR_SB();
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur][1], vc[kp][1], acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur][1], vc[kp][1], acc[1][1], 0, 0, 0);
            KMSR_SB();
            cols_of(t, vn[kp]);
            asm volatile("" : "+v"(vn[kp][0]), "+v"(vn[kp][1]), "+v"(vn[kp][2]), "+v"(vn[kp][3]));
            KMSR_SB();
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur][2], vc[kp][2], acc[0][2], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur][2], vc[kp][2], acc[1][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur][3], vc[kp][3], acc[0][3], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur][3], vc[kp][3], acc[1][3], 0, 0, 0);
            KMSR_SB();
            if (kp == 0) {
                if (chunk + 1 < nch) issue_u(chunk + 1);
                if (chunk + 3 < nch) issue_raw(chunk + 3);
                if (chunk + 1 == nch) issue_residual();
            }
        }
    };
#undef KMSR_SB

    // ---- prologue: raw(0), raw(1), U(0), raw(2) in flight; V(0) needs raw(0)
    issue_raw(0);
    if (nch > 1) issue_raw(1);
    issue_u(0);
    if (nch > 2) issue_raw(2);
    if (nch > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RINSTR + UINSTR) : "memory");
    else if (nch > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RINSTR + UINSTR) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UINSTR) : "memory");
    barrier_lds();
    float vA[4][4], vB[4][4];
    transform_now(0, vA);
    for (int chunk = 0; chunk < nch; chunk += 2) {
        phase(chunk, vA, vB);
        if (chunk + 1 < nch) phase(chunk + 1, vB, vA);
    }

    // ---- epilogue.  Column transform of this wave's xi row: MA[b] = sum_j M[j] A[j][b], A^T = [[1,1,1,0],[0,1,-1,-1]].
    //      Row transform across the 4 waves of a tile row: Y[0][b] = MA_0 + MA_1 + MA_2, Y[1][b] = MA_1 - MA_2 - MA_3
    //      (xi row 2: U and V both carry the folded sign, so the accumulators hold M itself).
    f32x2 ma[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float m0 = acc[cb][0][reg], m1 = acc[cb][1][reg], m2 = acc[cb][2][reg], m3 = acc[cb][3][reg];
            ma[cb][reg] = f32x2{m0 + m1 + m2, m1 - m2 - m3};
        }
    // skip values of this wave's outputs, [4g + r] <-> channel co0(g) + r, pixel pair (ry, rx0 .. rx0+1), from the LDS staging
    f32x2 rres[16];
    const bool has_res = !PS && res != nullptr;              // block-uniform
    if (has_res) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's own staging DMA has landed (wave-private region)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rres[4 * g + r] = *reinterpret_cast<const f32x2*>(rstage + (8 * g + 4 * half + r) * 64 + 2 * l32);
    }
    barrier_lds();                                           // every wave is done with U / raw / staging: the exchange buffer may overwrite them
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
            if (!(cb == ocb && (xr == 0 || xr == 3))) xch[((wv * 2 + cb) * 16 + reg) * 64 + lane] = ma[cb][reg];
        }
    }
    __syncthreads();
    // Y[orow] = sum over xi rows i' of AT[orow][i'] * MA_i':  row 0: +0 +1 +2;  row 1: +1 -2 -3.  Own contribution from registers.
    const int wbase = tb * 4;
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
    const f32x2* xa = xch + (((wbase + pa_) * 2 + ocb) * 16) * 64 + lane;
    const f32x2* xb = xch + (((wbase + pb_) * 2 + ocb) * 16) * 64 + lane;
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
        reinterpret_cast<uintptr_t>(residual) % 8)
        return fail(KMSR_E_BADARG, "winograd-vreg conv needs 16-byte aligned input / weights and 8-byte aligned output");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if ((long long)d->H * d->W * VCK >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "image plane exceeds 2^28 elements");
    WinoVParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, 2 * TCOLS);
    p.tiles = p.tiles_x * ceil_div(d->H, 2 * TROWS);
    p.cgs = ceil_div(d->Cout, VCO);
    p.nchunks = ceil_div(d->Cin, VCK);
    p.relu = d->relu;
    const long long nblocks = (long long)p.tiles * p.cgs * d->N;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_blocks = (int)nblocks;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->pixel_shuffle)
        conv3x3_winov_kernel<true><<<p.n_blocks, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else
        conv3x3_winov_kernel<false><<<p.n_blocks, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
