// 3x3 convolution as an implicit GEMM on the gfx950 fp32 matrix cores (see include/kmsr.h for semantics).
//
//   D[co, px] += A[co, k] * B[k, px],  k = (tap, ci),  one v_mfma_f32_32x32x2_f32 per 32 co x 32 px x 2 ci
//   A = weights  (lane l holds A[co = l&31][k = l>>5]),  B = input pixels (lane l holds B[k = l>>5][px = l&31])
//   D: lane l holds px = l&31 and 16 output channels (reg&3) + 8*(reg>>2) + 4*(l>>5)  -> NCHW stores are two
//      128-byte row segments per instruction, and a 2x2 PixelShuffle group (4 consecutive co) sits in 4
//      consecutive registers of ONE lane, so the shuffle is fused into the store as two 8-byte writes.
//
// Block = 4 waves, output tile 8 rows x 32 cols x 64 output channels; wave w owns rows 2w, 2w+1 (2x2 MFMA
// blocks, 64 accumulator VGPRs).  Input channels stream through LDS in chunks of 8 (input halo tile
// 8 x 10 x 34 + the chunk's 9 x 8 x 64 weights = 30 KB), double buffered: the next chunk's global loads are
// issued before the current chunk's 144 MFMAs per wave and written to the other buffer after them, one
// __syncthreads per chunk.  Two blocks per CU (2 waves per SIMD) overlap one block's barrier with the other's MFMAs.
// The stride-2 instance stages 4 channels per chunk (8 x 17 x 68 floats would be 110 KB double buffered = ONE block per CU, whose first-chunk
// load latency and 64-stores-per-lane epilogue nothing covers: 0.40 of the fp32 pipe at Cin = 32, round 4); its chunks are half chunks of
// the same packed weight layout.  Input loads are buffer loads: padding positions and the channels past Cin carry an out-of-range offset and
// come back as zeros, so a load costs no VALU instruction (the select / 64-bit address pairs were 200 of 330 VALU per chunk next to 144 MFMAs).
#include "kmsr_common.h"

#include <algorithm>
#include <type_traits>

namespace kmsr {
namespace {

constexpr int CK = 8;     // input channels per chunk of the PACKED weight layout (and per LDS chunk of the stride-1 instances)
constexpr int TH = 8;     // output tile rows per block
constexpr int TW = 32;    // output tile cols per block (= MFMA N)
constexpr int COB = 64;   // output channels per block (2 MFMA M blocks)
#ifndef KMSR_S2_FETCH
#define KMSR_S2_FETCH 1
#define KMSR_S2_RESTAGE 7
#endif
constexpr int kFetchTap = 1;    // the next chunk's global loads are issued before this tap (after the first tap's 16 MFMAs)
constexpr int kRestageTap = 7;  // ... and written to the other LDS buffer before this tap (of 9)

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct ConvParams {
    const float* x;
    const float* wpk;
    const float* bias;
    const float* res;
    float* out;
    int N, Cin, H, W, Cout, Ho, Wo;
    int tiles_x, tiles, cgs, nchunks, relu, n_blocks;
};

template <int S>
struct ConvCfg {
    static constexpr int IH = (TH - 1) * S + 3;
    static constexpr int IW = (TW - 1) * S + 3;
    // stride 2: row pitch 72 with image column 2 tx0 - 4 at LDS column 0 (the tile's first column, 2 tx0 - 1, at COL0 = 3), so that rows can
    // be staged as 18 aligned 16-byte pieces when W % 4 == 0 (WIDE instance); stride 1: pitch 36, first column at 0
    static constexpr int IWP = S == 2 ? 72 : (IW + 3) / 4 * 4;
    static constexpr int COL0 = S == 2 ? 3 : 0;
    static constexpr int CKC = S == 2 ? 4 : CK;                // input channels per LDS chunk; CK / CKC kernel chunks per packed chunk
    static constexpr int QROW = IWP / 4;                       // 16-byte pieces per staged row (WIDE)
    static constexpr int NQ4 = (CKC * IH * QROW + 255) / 256;  // pieces per thread and chunk (WIDE): 5
    static constexpr int IN_FLOATS = CKC * IH * IWP;
    static constexpr int W_FLOATS = 9 * CKC * COB;
    static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
    static constexpr int IN_ELEMS = CKC * IH * IW;
    static constexpr int NIN = (IN_ELEMS + 255) / 256;
    static constexpr int NW4 = (W_FLOATS / 4 + 255) / 256;
    static constexpr int PACKED_CHUNK_FLOATS = 9 * CK * COB;
};

// packed[cg][chunk][tap][c][co]  <-  w[co_g][ci_g][tap]   (zero padded); transpose_flip: data-gradient weights
__global__ void pack_weights_kernel(const float* __restrict__ w, int Cin, int Cout, int tf, float* __restrict__ packed,
                                    int nchunks, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % COB);
        const int c = (int)((idx / COB) % CK);
        const int tap = (int)((idx / (COB * CK)) % 9);
        const int chunk = (int)((idx / (COB * CK * 9)) % nchunks);
        const int cg = (int)(idx / ((long long)COB * CK * 9 * nchunks));
        const int cop = cg * COB + co, cip = chunk * CK + c;      // channel indices of the PACKED convolution
        const int CinP = tf ? Cout : Cin, CoutP = tf ? Cin : Cout;
        float v = 0.f;
        if (cop < CoutP && cip < CinP) {
            if (!tf) v = w[((long long)cop * Cin + cip) * 9 + tap];
            else v = w[((long long)cip * Cin + cop) * 9 + (8 - tap)];  // w'[co'][ci'][u][v] = w[ci'][co'][2-u][2-v]
        }
        packed[idx] = v;
    }
}

template <int S, bool PS, bool WIDE = false>
__global__ void __launch_bounds__(256, 2) conv3x3_mfma_kernel(const float* __restrict__ xg, const float* __restrict__ wpk,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res, float* __restrict__ out,
                                                              ConvParams p) {
    using Cfg = ConvCfg<S>;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, IWP = Cfg::IWP, IN_FLOATS = Cfg::IN_FLOATS, W_FLOATS = Cfg::W_FLOATS;
    constexpr int BUF = Cfg::BUF_FLOATS, IN_ELEMS = Cfg::IN_ELEMS, NW4 = Cfg::NW4, CKC = Cfg::CKC, COL0 = Cfg::COL0;
    static_assert(!WIDE || S == 2, "16-byte staging is built for the stride-2 tile");
    // WIDE (stride 2, W % 4 == 0, 16-byte aligned x): a staged element is an aligned 16-byte piece of a row -- 5 loads and 5 ds_write_b128 per
    // thread and chunk instead of 18 + 18 dword ones.  Every vector-memory instruction issued inside the MFMA stream holds the wave's issue
    // for 20-50 cycles (tools/ubench/mfma_store.hip): the 18 loads were most of what the "no loads" ablation gave back (18 % of the kernel).
    using stage_t = std::conditional_t<WIDE, f32x4, float>;
    constexpr int NIN = WIDE ? Cfg::NQ4 : Cfg::NIN, QROW = Cfg::QROW;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware remap of the linear block id: blocks that share an XCD (id % 8) get a contiguous range of tiles,
    // so neighbouring tiles (shared halo) and all cout groups of an image meet in one L2.
    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * TH, tx0 = (tile % p.tiles_x) * TW;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    // ---- staging map of the input halo tile (same for every chunk): BYTE offsets from the chunk's first channel plane; zero padding and
    //      the slots past the tile get an offset no buffer covers (the host checks CK * H * W * 4 < 2^32)
    unsigned goff[NIN];
    int loff[NIN];                              // the slot's LDS position inside a buffer (floats)
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
        const int e = tid + 256 * t;
        if constexpr (WIDE) {
            const int c = e / (IH * QROW), rem = e - c * (IH * QROW), r = rem / QROW, k = rem - r * QROW;
            const int gy = ty0 * S - 1 + r, gx = tx0 * S - 4 + 4 * k;           // W % 4 == 0: a piece is entirely inside or outside the row
            const bool ok = e < CKC * IH * QROW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            goff[t] = ok ? (unsigned)(c * HW + (long long)gy * W + gx) * 4u : 0xFFFFFFFFu;
            loff[t] = e < CKC * IH * QROW ? (c * IH + r) * IWP + 4 * k : -1;
        } else {
            const int c = e / (IH * IW), rem = e - c * (IH * IW), r = rem / IW, col = rem - r * IW;
            const int gy = ty0 * S - 1 + r, gx = tx0 * S - 1 + col;
            const bool ok = e < IN_ELEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            goff[t] = ok ? (unsigned)(c * HW + (long long)gy * W + gx) * 4u : 0xFFFFFFFFu;
            loff[t] = e < IN_ELEMS ? (c * IH + r) * IWP + COL0 + col : -1;
        }
#ifdef KMSR_ABL_NOLOAD      // timing ablation (tools/diag/ab_s2.sh): every input load out of range -> zeros, no memory traffic
        goff[t] = 0xFFFFFFFFu;
#endif
    }
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const float* __restrict__ wcg = wpk + (long long)cg * p.nchunks * Cfg::PACKED_CHUNK_FLOATS;

    // One register staging set: the next chunk's global loads are ISSUED after the first tap's MFMAs (their ~80
    // address/load instructions then issue in the shadow of the matrix pipe instead of ahead of it) and are written
    // to the other LDS buffer before tap kRestageTap of the same phase.
    stage_t xinA[NIN];
    f32x4 wregA[NW4];
    auto load_chunk = [&](int chunk, stage_t (&xin)[NIN], f32x4 (&wreg)[NW4]) {
        // the buffer = what is left of this image from the chunk's first channel: channels past Cin are out of range too (channel padding)
        const long long left = ((long long)p.Cin - (long long)chunk * CKC) * HW * 4;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xn + (long long)chunk * CKC * HW), 0, (unsigned)(left < 0xFFFFFFFFll ? left : 0xFFFFFFFFll), 0x00020000);
#pragma unroll
        for (int t = 0; t < NIN; ++t) {
            if constexpr (WIDE) xin[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, goff[t], 0, 0));
            else xin[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[t], 0, 0));
        }
        // kernel chunk -> (packed chunk, part): per tap the part's CKC x COB floats are contiguous in the packed layout
        constexpr int PARTS = CK / CKC, TAP4 = CKC * COB / 4;
        const f32x4* __restrict__ wc = reinterpret_cast<const f32x4*>(wcg + (long long)(chunk / PARTS) * Cfg::PACKED_CHUNK_FLOATS) + (chunk % PARTS) * TAP4;
#pragma unroll
        for (int t = 0; t < NW4; ++t) {
            const int e4 = tid + 256 * t;
            if (e4 < W_FLOATS / 4) wreg[t] = wc[PARTS == 1 ? e4 : (e4 / TAP4) * (PARTS * TAP4) + e4 % TAP4];
        }
    };
    auto store_chunk = [&](int b, const stage_t (&xin)[NIN], const f32x4 (&wreg)[NW4]) {
        float* inb = lds + b * BUF;
        float* wb = inb + IN_FLOATS;
#pragma unroll
        for (int t = 0; t < NIN; ++t)
            if (loff[t] >= 0) *reinterpret_cast<stage_t*>(inb + loff[t]) = xin[t];
#pragma unroll
        for (int t = 0; t < NW4; ++t) {
            const int e4 = tid + 256 * t;
            if (e4 < W_FLOATS / 4) reinterpret_cast<f32x4*>(wb)[e4] = wreg[t];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    const int aoff = half * COB + l32;                                        // + (tap*CKC + 2*kp)*COB + 32*mt
    const int boff = half * (IH * IWP) + (2 * wv * S) * IWP + l32 * S + COL0;  // + (2*kp)*IH*IWP + (nt*S + dy)*IWP + dx

    // one chunk: 9 taps x 4 channel pairs x (2x2) MFMAs out of LDS buffer b
    // The 4 operands of group (tap, kp) are requested one group AHEAD of their MFMAs (register pins keep hipcc from sinking the
    // reads back: it otherwise emits  mfma x3, ds_read, s_waitcnt lgkmcnt(0), mfma  -- an LDS round trip per 4 MFMAs, which the
    // one-workgroup-per-CU stride-2 instantiation cannot hide).
    auto mma_chunk = [&](int b, auto&& fetch, auto&& restage) {
        const float* inb = lds + b * BUF;
        const float* wb = inb + IN_FLOATS;
        auto operands = [&](int tap, int kp, float& a0, float& a1, float& b0, float& b1) {
            const int dy = tap / 3, dx = tap % 3;
            a0 = wb[aoff + (tap * CKC + 2 * kp) * COB];
            a1 = wb[aoff + (tap * CKC + 2 * kp) * COB + 32];
            b0 = inb[boff + (2 * kp) * (IH * IWP) + dy * IWP + dx];
            b1 = inb[boff + (2 * kp) * (IH * IWP) + (S + dy) * IWP + dx];
        };
        float a0, a1, b0, b1;
        operands(0, 0, a0, a1, b0, b1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap == (S == 2 ? KMSR_S2_FETCH : kFetchTap)) fetch();
            if (tap == (S == 2 ? KMSR_S2_RESTAGE : kRestageTap)) restage();
#pragma unroll
            for (int kp = 0; kp < CKC / 2; ++kp) {
                float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
                if (kp + 1 < CKC / 2) operands(tap, kp + 1, na0, na1, nb0, nb1);
                else if (tap + 1 < 9) operands(tap + 1, 0, na0, na1, nb0, nb1);
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        }
    };

    const int nch = (p.Cin + CKC - 1) / CKC;      // p.nchunks counts chunks of the packed layout (CK channels)
    if constexpr (S == 2) {
        // Two register staging sets: chunk k + 2 is REQUESTED during chunk k and written to LDS during chunk k + 1.  One chunk of lookahead is
        // 48 MFMAs of this wave between request and use; a chunk's 204 L2-missing lines drain through the CU's 64-entry miss queue in
        // ~3 500 cycles while the partner workgroup's requests queue behind them: the "no loads" ablation said the
        // waits cost 18 % of the kernel (tools/diag/ab_s2.sh, round 4).
        stage_t xinB[NIN];
        f32x4 wregB[NW4];
        load_chunk(0, xinA, wregA);
        if (nch > 1) load_chunk(1, xinB, wregB);
        store_chunk(0, xinA, wregA);
        __syncthreads();
        for (int chunk = 0; chunk < nch; chunk += 2) {
            {   // even chunk: LDS buffer 0; request chunk + 2 into set A (free since chunk's own restage), write chunk + 1 from set B
                const bool m2 = chunk + 2 < nch, m1 = chunk + 1 < nch;      // block-uniform
                mma_chunk(0, [&] { if (m2) load_chunk(chunk + 2, xinA, wregA); }, [&] { if (m1) store_chunk(1, xinB, wregB); });
                __syncthreads();
            }
            if (chunk + 1 < nch) {
                const bool m2 = chunk + 3 < nch, m1 = chunk + 2 < nch;
                mma_chunk(1, [&] { if (m2) load_chunk(chunk + 3, xinB, wregB); }, [&] { if (m1) store_chunk(0, xinA, wregA); });
                __syncthreads();
            }
        }
    } else {
        load_chunk(0, xinA, wregA);
        store_chunk(0, xinA, wregA);
        __syncthreads();
        for (int chunk = 0; chunk < nch; ++chunk) {
            const int b = chunk & 1;
            const bool more = chunk + 1 < nch;   // block-uniform
            mma_chunk(b, [&] { if (more) load_chunk(chunk + 1, xinA, wregA); },
                      [&] { if (more) store_chunk(b ^ 1, xinA, wregA); });   // the other buffer was last read before the previous barrier
            __syncthreads();
        }
    }

    // ---- epilogue: bias, ReLU, residual, (PixelShuffle) store.  No per-element branches around loads: the bias goes
    //      through LDS (one 16-byte read per 4 channels), residuals are fetched 16 at a time from clamped-safe
    //      addresses before any of them is consumed, so the block pays one memory round trip per 16 outputs, not 16.
    float* sbias = lds;   // all MFMA reads of the staging buffers retired at the last barrier
    if (tid < COB) {
        const int co = cg * COB + tid;
        sbias[tid] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    __syncthreads();
    const int x = tx0 + l32;
    const bool xok = x < p.Wo;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int y = ty0 + 2 * wv + nt;
            const bool pok = xok && y < p.Ho;
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + mt * 32 + 8 * g + 4 * half);
                v[4 * g + 0] = acc[mt][nt][4 * g + 0] + b4.x;
                v[4 * g + 1] = acc[mt][nt][4 * g + 1] + b4.y;
                v[4 * g + 2] = acc[mt][nt][4 * g + 2] + b4.z;
                v[4 * g + 3] = acc[mt][nt][4 * g + 3] + b4.w;
            }
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (PS) {
                // out[n, co/4, 2y + (co/2)%2, 2x + co%2]; the 4 registers of group g are one 2x2 output block
                const long long W2 = 2ll * p.Wo;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co0 = cg * COB + mt * 32 + 8 * g + 4 * half;
                    if (pok && co0 < p.Cout) {
                        float* o = out + (((long long)n * (p.Cout >> 2) + (co0 >> 2)) * (2ll * p.Ho) + 2 * y) * W2 + 2 * x;
                        *reinterpret_cast<f32x2*>(o) = f32x2{v[4 * g + 0], v[4 * g + 1]};
                        *reinterpret_cast<f32x2*>(o + W2) = f32x2{v[4 * g + 2], v[4 * g + 3]};
                    }
                }
            } else {
                const long long plane = (long long)p.Ho * p.Wo;
                const long long pix = (long long)y * p.Wo + x;
                long long idx[16];
                bool ok[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cg * COB + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    ok[r] = pok && co < p.Cout;
                    idx[r] = ok[r] ? ((long long)n * p.Cout + co) * plane + pix : 0;
                }
                if (res) {   // block-uniform
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = res[idx[r]];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += rv[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
#ifdef KMSR_ABL_NOSTORE     // timing ablation: the stores are compiled but never executed
                    if (ok[r] && p.N < 0) out[idx[r]] = v[r];
#else
                    if (ok[r]) out[idx[r]] = v[r];
#endif
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Small-Cout direct kernel (Cout <= 8: the SR tail conv F -> bands).  With 3..7 output channels an MFMA
// tile would be >= 78 % zero padding, while the fp32 VALU pipe has the same 157 TF peak, so this path is plain
// v_fma: each thread owns 4 consecutive pixels x COUT channels, the input halo tile streams through LDS in
// chunks of 4 channels (double buffered), and the weights are wave-uniform -> scalar loads, SGPR operands.
// Packed layout for this path: [ceil(Cin/4)*4][9][COUT].
// ------------------------------------------------------------------------------------------------
constexpr int SCK = 4;            // input channels per LDS chunk
constexpr int STH = 16, STW = 64; // output tile
constexpr int SIH = STH + 2, SIW = STW + 2, SIWP = 68;

template <int COUT>
__global__ void __launch_bounds__(256) conv3x3_smallco_kernel(const float* __restrict__ xg, const float* __restrict__ wpk,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ res, float* __restrict__ out,
                                                               ConvParams p) {
    constexpr int IN_FLOATS = SCK * SIH * SIWP;
    constexpr int NIN = (IN_FLOATS + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * IN_FLOATS];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    int bid = blockIdx.x;
    const int tile = bid % p.tiles;
    const int n = bid / p.tiles;
    const int ty0 = (tile / p.tiles_x) * STH, tx0 = (tile % p.tiles_x) * STW;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    // staging map over the LDS image itself: slot L = tid + 256 t of the [channel][row][SIWP] tile -> (c, r, col); its global BYTE offset from the
    // chunk's first channel plane, or an offset no buffer covers for zero padding, the two pitch columns and the slots past the tile; channels
    // past Cin fall off the end of the chunk's buffer.  A load is one buffer instruction and a stage write one ds_write_b32 at a compile-time
    // offset from ONE base register -- neither costs a VALU instruction on a kernel that is VALU-bound (round 4).
    unsigned goff[NIN];
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
        const int e = tid + 256 * t;
        const int c = e / (SIH * SIWP), rem = e - c * (SIH * SIWP), r = rem / SIWP, col = rem - r * SIWP;
        const int gy = ty0 - 1 + r, gx = tx0 - 1 + col;
        const bool ok = e < IN_FLOATS && col < SIW && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[t] = ok ? (unsigned)(c * HW + (long long)gy * W + gx) * 4u : 0xFFFFFFFFu;
    }
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    float xin[NIN];
    auto load_chunk = [&](int chunk) {
        const long long left = ((long long)p.Cin - (long long)chunk * SCK) * HW * 4;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xn + (long long)chunk * SCK * HW), 0, (unsigned)(left < 0xFFFFFFFFll ? left : 0xFFFFFFFFll), 0x00020000);
#pragma unroll
        for (int t = 0; t < NIN; ++t) xin[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[t], 0, 0));
    };
    auto store_chunk = [&](int b) {
        float* inb = lds + b * IN_FLOATS + tid;
#pragma unroll
        for (int t = 0; t < NIN; ++t)
            if (256 * t + 255 < IN_FLOATS || tid + 256 * t < IN_FLOATS) inb[256 * t] = xin[t];
    };

    // Accumulators as packed pairs of adjacent pixels: v_pk_fma_f32 issues two FMAs per slot with the (wave-uniform) weight
    // broadcast from its SGPR -- this layer is VALU-bound (2 * 9 * Cin * COUT flops per output pixel on the vector pipe).
    // Pixel pairs at odd tap offsets are read as their own aligned register pairs (ds_read2_b32), not shuffled.
    f32x2 acc[COUT][2];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co][0] = acc[co][1] = f32x2{0.f, 0.f};

    const int nchunks = (p.Cin + SCK - 1) / SCK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int b = chunk & 1;
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);
        const float* inb = lds + b * IN_FLOATS + ty * SIWP + 4 * tx;
        const float* __restrict__ wc = wpk + (long long)chunk * SCK * 9 * COUT;   // wave-uniform
#pragma unroll 1   // keep one channel's LDS values live at a time: ~100 VGPRs -> 4 waves/SIMD instead of 2
        for (int c = 0; c < SCK; ++c) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* rowp = inb + (c * SIH + dy) * SIWP;
                const f32x4 a = *reinterpret_cast<const f32x4*>(rowp);
                const f32x2 v45 = *reinterpret_cast<const f32x2*>(rowp + 4);
                const f32x2 v12 = f32x2{rowp[1], rowp[2]}, v34 = f32x2{rowp[3], rowp[4]};
                const f32x2 pa[3] = {f32x2{a.x, a.y}, v12, f32x2{a.z, a.w}};
                const f32x2 pb[3] = {f32x2{a.z, a.w}, v34, v45};
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const float wv = wc[(c * 9 + dy * 3 + dx) * COUT + co];
                        const f32x2 w2 = f32x2{wv, wv};
                        acc[co][0] = __builtin_elementwise_fma(w2, pa[dx], acc[co][0]);
                        acc[co][1] = __builtin_elementwise_fma(w2, pb[dx], acc[co][1]);
                    }
                }
            }
        }
        if (chunk + 1 < nchunks) store_chunk(b ^ 1);
        __syncthreads();
    }

    const int y = ty0 + ty, x = tx0 + 4 * tx;
    if (y < p.Ho) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float bv = bias ? bias[co] : 0.f;
            const long long base = (((long long)n * COUT + co) * p.Ho + y) * p.Wo + x;
            float v[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                v[px] = acc[co][px >> 1][px & 1] + bv;
                if (p.relu) v[px] = fmaxf(v[px], 0.f);
            }
            if (x + 3 < p.Wo && (p.Wo & 3) == 0) {
                if (res) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(res + base);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                *reinterpret_cast<f32x4*>(out + base) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int px = 0; px < 4; ++px)
                    if (x + px < p.Wo) out[base + px] = v[px] + (res ? res[base + px] : 0.f);
            }
        }
    }
}

// packed[ci][tap][co] <- w[co][ci][tap], ci zero-padded to a multiple of SCK
__global__ void pack_weights_smallco_kernel(const float* __restrict__ w, int Cin, int Cout, int tf,
                                            float* __restrict__ packed, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int CinP = tf ? Cout : Cin, CoutP = tf ? Cin : Cout;
        const int co = (int)(idx % CoutP);
        const int tap = (int)((idx / CoutP) % 9);
        const int ci = (int)(idx / ((long long)CoutP * 9));
        float v = 0.f;
        if (ci < CinP) v = !tf ? w[((long long)co * Cin + ci) * 9 + tap] : w[((long long)ci * Cin + co) * 9 + (8 - tap)];
        packed[idx] = v;
    }
}

constexpr int kSmallCoMax = 8;

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv3x3_packed_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    if (Cout <= kSmallCoMax) return (size_t)round_up(Cin, SCK) * 9 * Cout;   // small-Cout direct kernel layout
    return (size_t)ceil_div(Cout, COB) * ceil_div(Cin, CK) * 9 * CK * COB;
}

extern "C" int kmsr_conv3x3_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, int32_t transpose_flip,
                                             float* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "pack_weights: bad argument");
    const int CinP = transpose_flip ? Cout : Cin, CoutP = transpose_flip ? Cin : Cout;
    if (CoutP <= kSmallCoMax) {
        const long long total = (long long)round_up(CinP, SCK) * 9 * CoutP;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
        pack_weights_smallco_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, transpose_flip,
                                                                                        packed, total);
        KMSR_HIP_CHECK(hipGetLastError());
        return KMSR_OK;
    }
    const int nchunks = ceil_div(CinP, CK);
    const long long total = (long long)ceil_div(CoutP, COB) * nchunks * 9 * CK * COB;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    pack_weights_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, transpose_flip, packed,
                                                                             nchunks, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv3x3_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                                    const float* bias, const float* residual, float* out, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1 && d->stride != 2) return fail(KMSR_E_UNSUPPORTED, "conv3x3 stride %d not supported (1 or 2)", d->stride);
    if (d->flags) return fail(KMSR_E_UNSUPPORTED, "conv flags 0x%x are honoured by kmsr_conv3x3_winov_fwd_ex_f32 only", d->flags);
    if (d->stride == 2 && d->pixel_shuffle) return fail(KMSR_E_UNSUPPORTED, "pixel_shuffle needs stride 1");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    ConvParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.Ho = (d->H - 1) / d->stride + 1;
    p.Wo = (d->W - 1) / d->stride + 1;
    if ((long long)d->Cin * d->H * d->W >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "per-image input exceeds 2^31 elements");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if ((long long)CK * d->H * d->W * 4 >= (1ll << 32))
        return fail(KMSR_E_UNSUPPORTED, "conv3x3: %d x %d planes exceed the 32-bit buffer offsets of a channel chunk", d->H, d->W);
    if (d->Cout <= kSmallCoMax) {
        if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "stride 2 with Cout <= %d is not implemented", kSmallCoMax);
        if (d->pixel_shuffle) return fail(KMSR_E_UNSUPPORTED, "pixel_shuffle with Cout <= %d is not implemented", kSmallCoMax);
        p.tiles_x = ceil_div(p.Wo, STW);
        p.tiles = p.tiles_x * ceil_div(p.Ho, STH);
        p.relu = d->relu;
        const long long nb = (long long)p.tiles * d->N;
        if (nb > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
        switch (d->Cout) {
#define KMSR_SC(C_) case C_: conv3x3_smallco_kernel<C_><<<(int)nb, 256, 0, stream>>>(x, packed_w, bias, residual, out, p); break;
            KMSR_SC(1) KMSR_SC(2) KMSR_SC(3) KMSR_SC(4) KMSR_SC(5) KMSR_SC(6) KMSR_SC(7) KMSR_SC(8)
#undef KMSR_SC
        }
        KMSR_HIP_CHECK(hipGetLastError());
        return KMSR_OK;
    }
    p.tiles_x = ceil_div(p.Wo, TW);
    p.tiles = p.tiles_x * ceil_div(p.Ho, TH);
    p.cgs = ceil_div(d->Cout, COB);
    p.nchunks = ceil_div(d->Cin, CK);
    p.relu = d->relu;
    const long long nblocks = (long long)p.tiles * p.cgs * d->N;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_blocks = (int)nblocks;
    if (d->pixel_shuffle)
        conv3x3_mfma_kernel<1, true><<<p.n_blocks, 256, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else if (d->stride == 2 && d->W % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0)
        conv3x3_mfma_kernel<2, false, true><<<p.n_blocks, 256, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else if (d->stride == 2)
        conv3x3_mfma_kernel<2, false><<<p.n_blocks, 256, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else
        conv3x3_mfma_kernel<1, false><<<p.n_blocks, 256, 0, stream>>>(x, packed_w, bias, residual, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
