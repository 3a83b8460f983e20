// Fused bf16 residual block  y = x + conv2(relu(conv1(x) + b1)) + b2  on channels-last bf16 activations with 64 channels
// (semantics: include/kmsr.h, kmsr_resblock_fwd_bf16; BASELINE config 4, the HBM-bound large-tile path).
//
// Layer by layer (conv3x3_bf16.hip) a res-block moves 5 feature maps through HBM (x, t, t, x, y: 670 MB at 1024 x 1024) and the 33
// body launches are 84 % of the config-4 tile at half the HBM peak.  Here the intermediate t never leaves the CU: one launch reads
// x once (plus the 2-pixel halo, L2 hits) and writes y once -- 2 maps -- and recomputes conv1 on the 1-pixel halo ring of each tile.
//
//   tile     : 8 x 16 output pixels; conv1 is evaluated on the 10 x 18 region around it from a 12 x 20 input tile.
//   MFMA     : v_mfma_f32_32x32x16_bf16, D[32 co][32 px] += A[32 co][16 k] B[16 k][32 px], k = 16 input channels of one tap, fp32
//              accumulation in the SAME (tap, k-step) order as conv3x3_bf16_rw_kernel.  The N dimension is a FLATTENED pixel index of
//              the region (180 conv1 pixels = 6 N-tiles, 128 conv2 pixels = 4): a tap is one constant LDS offset for all lanes.
//   waves    : four 512-register waves = (co half) x (pixel half); each keeps ITS 32 output channels of BOTH weight slabs in
//              registers (2 x 144) and multiplies 3 conv1 + 2 conv2 N-tiles per tile: 180 MFMAs per 128 pixels (144 layer by layer
//              at the same accumulation order; + 25 % for the halo ring).
//   LDS      : input tile double-buffered (next tile by buffer LDS-DMA during the products; pixel pitch 144 B, the pad lane and
//              out-of-image pixels are out-of-range offsets -> zeros = conv1's zero padding), t tile (bf16, ZERO outside the image:
//              conv2's zero padding applies to t, not to conv1 of padded x), staging tile for 16-byte coalesced stores: 120 KB.
//   parity   : every rounding of the two-launch path is kept -- t = bf16(relu(acc1 + b1)), c = bf16(acc2 + b2), y = bf16(c + x) --
//              so the output is BIT-IDENTICAL to kmsr_conv3x3_fwd_bf16 x 2 (tests/test_bf16_gpu.py).
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int RTH = 8, RTW = 16;                 // output tile
constexpr int RMH = RTH + 2, RMW = RTW + 2;      // conv1 region 10 x 18
constexpr int RIH = RTH + 4, RIW = RTW + 4;      // input tile 12 x 20
constexpr int RC = 64, RCP = RC + 8;             // channels; pixel pitch in bf16 elements (144 B: conflict-free 16-byte reads)
constexpr int RUP = RC / 8 + 1;                  // 16-byte units per pixel incl. the pad unit
constexpr int RNI = (RIH * RIW * RUP + 63) / 64; // 34 DMA instructions per input tile
constexpr int RNIW = (RNI + 3) / 4;              // 9 per wave
constexpr int RIN = RNIW * 4 * 512;              // bf16 elements of one input buffer (36 instruction slots)
constexpr int RM_PIX = RMH * RMW;                // 180
constexpr int RPF = 3;                           // B-operand groups in flight ahead of their MFMAs
constexpr int RT_ELEMS = 192 * RCP;              // t tile: 6 N-tiles of 32 pixels
constexpr int RS_ELEMS = RTH * RTW * RCP;        // staging tile
constexpr size_t RLDS_BYTES = (size_t)(2 * RIN + RT_ELEMS + RS_ELEMS) * 2 + 2 * RC * sizeof(float);

struct RbParams {
    int N, H, W, tiles_x, tiles, n_items;
};

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ unsigned cvt_pk(f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }   // v_cvt_pk_bf16_f32
__device__ __forceinline__ f32x2_t unpk(unsigned u) { return f32x2_t{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)}; }
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk(unsigned u) {                     // ReLU of a packed bf16 pair: v_pk_max_i16 with 0
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, u), i16x2_t{0, 0}));
}
// Nothing may move across the end of an MFMA phase, and the phase drains before the epilogue's first VALU instruction: a VALU read of a
// v_mfma_f32_32x32x16_bf16 result is NOT interlocked on gfx950 -- it returns the old register contents until 11 wait states after the MFMA
// issued (tools/ubench/mfma_bf16_hazards.hip, profiles/r04_ubench_mfma_bf16_hazards.log: s_nop 9 fails on every lane, s_nop 10 passes 1.5e9
// checks).  hipcc's hazard recognizer inserts exactly those 11 for the builtin, but the inline-asm MFMAs here are invisible to it, and once the
// epilogue was written with packed operations the scheduler moved its first accumulator reads up between the last MFMAs of the phase: accumulator
// registers 0 / 1 of half the tiles came out wrong in some runs (tools/diag/dbg_rb.py).  The other suspects were measured innocent by the same
// microbenchmark: a VALU write to SrcB right behind the MFMA, and VALU writes to SrcC / SrcB right in front of it, never change the result
// (DESIGN 4.12 fact 11).  The fence = no reordering across it + 12 wait states.
#define RB_DRAIN()                                                                        \
    {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_nop 11" ::: "memory");                                            \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }
// v_mfma_f32_32x32x16_bf16 with explicit register classes: accumulator in VGPRs, weight operand IDX (0 .. 71 over both layers) in AGPRs
// for IDX < 64, in VGPRs otherwise (256 + 32 registers of weights per wave)
#define RB_MFMA(ACC, AOP, IDX, BOP)                                                                                   \
    {                                                                                                                 \
        if ((IDX) < 64) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(AOP), "v"(BOP)); \
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(AOP), "v"(BOP));                \
    }

__global__ void __launch_bounds__(256, 1) resblock_bf16_kernel(const unsigned short* __restrict__ xg, const unsigned short* __restrict__ w1,
                                                                const float* __restrict__ b1, const unsigned short* __restrict__ w2,
                                                                const float* __restrict__ b2, unsigned short* __restrict__ out, RbParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];        // the ONLY LDS object: in0 | in1 | t | staging | bias
    unsigned short* tl = smem + 2 * RIN;
    unsigned short* st = tl + RT_ELEMS;
    float* sbias = reinterpret_cast<float*>(st + RS_ELEMS);
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hco = wv & 1, ph = wv >> 1;          // output-channel half, pixel half
    const int H = p.H, W = p.W;

    const int per = (p.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
    int item = blockIdx.x * per;
    const int item_end = min(item + per, p.n_items);
    if (item >= item_end) return;

    // ---- DMA stream of an input tile: unit uidx = i * 64 + lane is piece uidx % 9 of tile pixel uidx / 9 (piece 8 = the pad unit)
    int ur[RNIW], uc[RNIW], urel[RNIW];
#pragma unroll
    for (int k = 0; k < RNIW; ++k) {
        const int uidx = (wv * RNIW + k) * 64 + lane, pix = uidx / RUP, piece = uidx - pix * RUP, r = pix / RIW, c = pix - r * RIW;
        const bool live = pix < RIH * RIW && piece < RC / 8;
        ur[k] = live ? r : -100000;                  // dead lanes fail the row check of every tile
        uc[k] = c;
        urel[k] = ((r - 2) * W + (c - 2)) * (RC * 2) + 16 * piece;       // bytes from the tile's first OUTPUT pixel
    }
    struct Origin { int n, ty0, tx0; };
    auto tile_origin = [&](int it) {
        Origin o;
        o.n = it / p.tiles;
        const int t = it - o.n * p.tiles;
        o.ty0 = (t / p.tiles_x) * RTH;
        o.tx0 = (t % p.tiles_x) * RTW;
        return o;
    };
    auto issue_tile = [&](const Origin& o, int buf) {
        const unsigned nrec = (unsigned)H * (unsigned)W * (unsigned)(RC * 2);          // bytes of one image (< 4 GiB, host-checked)
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xg + (long long)o.n * H * W * RC), 0, nrec, 0x00020000);
        const unsigned tbase = ((unsigned)o.ty0 * (unsigned)W + (unsigned)o.tx0) * (unsigned)(RC * 2);
#pragma unroll
        for (int k = 0; k < RNIW; ++k) {
            const int i = wv * RNIW + k;
            if (i < RNI) {
                const bool ok = (unsigned)(o.ty0 - 2 + ur[k]) < (unsigned)H && (unsigned)(o.tx0 - 2 + uc[k]) < (unsigned)W;
                const unsigned off = ok ? tbase + (unsigned)urel[k] : nrec;            // modulo-2^32 sum; nrec itself is out of range
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + buf * RIN + i * 512), 16, off, 0, 0, 0);
            }
        }
    };

    // ---- weights -> registers (this wave's 32 output channels of both layers: A lane l holds row l & 31, k-chunk 8 (l >> 5) + j)
    bf16x8 a1[9][4], a2[9][4];
    {
        const unsigned short* s1 = w1 + (long long)(32 * hco + l32) * RC + 8 * half;
        const unsigned short* s2 = w2 + (long long)(32 * hco + l32) * RC + 8 * half;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                a1[tap][ks] = *reinterpret_cast<const bf16x8*>(s1 + tap * 64 * RC + 16 * ks);
                a2[tap][ks] = *reinterpret_cast<const bf16x8*>(s2 + tap * 64 * RC + 16 * ks);
            }
    }
    if (tid < RC) {
        sbias[tid] = b1 ? b1[tid] : 0.f;
        sbias[RC + tid] = b2 ? b2[tid] : 0.f;
    }

    // ---- per-lane pixel maps of the N-tiles (tile-invariant)
    //      conv1: N-tile 3 ph + j, flattened region pixel P = 32 (3 ph + j) + l32 -> (r, c) = (P / 18, P % 18); P >= 180: dummy (clamped)
    int m_in[3], m_rc[3];                           // element offset of input pixel (r, c) + this lane's k half; r | c << 8 | live << 16
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int P = 32 * (3 * ph + j) + l32, Pc = min(P, RM_PIX - 1), r = Pc / RMW, c = Pc - r * RMW;
        m_in[j] = (r * RIW + c) * RCP + 8 * half;
        m_rc[j] = r | (c << 8) | ((P < RM_PIX ? 1 : 0) << 16);
    }
    //      conv2: N-tile 2 ph + j = output rows 2 (2 ph + j) + (l32 >> 4), column l32 & 15
    const int o_r0 = 4 * ph + (l32 >> 4), o_c = l32 & 15;

    // ---- store pieces of a thread: 128 pixels x 8 units of 16 bytes = 4 per thread
    int go[4], sr[4], sc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = tid + 256 * t, pix = e >> 3;
        sr[t] = pix / RTW;
        sc[t] = pix % RTW;
        go[t] = (sr[t] * W + sc[t]) * RC + 8 * (e & 7);
    }

    int buf = 0;
    Origin cur = tile_origin(item);
    issue_tile(cur, 0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);
    for (; item < item_end; ++item, buf ^= 1) {
        const bool more = item + 1 < item_end;
        const Origin nxt = more ? tile_origin(item + 1) : cur;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                        // A: this tile's input (waited for below / above) is visible; t, staging and the other buffer are free
        asm volatile("" ::: "memory");
        if (more) issue_tile(nxt, buf ^ 1);
        const unsigned short* il = smem + buf * RIN;

        // ---- conv1 on this wave's 3 N-tiles: 108 MFMAs.  B operands RPF groups ahead in a register ring (one wave per SIMD: nothing
        //      else covers an LDS round trip).  The MFMAs are inline asm with the register class in the constraints -- accumulators in
        //      ArchVGPRs (the epilogues read them: the builtin kept them in AGPRs, 75 v_accvgpr_read per tile), the first 64 of the 72
        //      weight operands in AGPRs, the rest in VGPRs.
        {
            f32x16 acc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            auto fetch = [&](int g, bf16x8 (&b)[3]) {
                const int tap = g >> 2, ks = g & 3, dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const bf16x8*>(il + m_in[j] + (dy * RIW + dx) * RCP + 16 * ks);
            };
            bf16x8 ring[RPF][3];
#pragma unroll
            for (int q = 0; q < RPF; ++q) fetch(q, ring[q]);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));       // the zeroed accumulators EXIST here (not re-materialised
            __builtin_amdgcn_sched_barrier(0);                                  // behind an MFMA whose SrcB registers they would reuse: RB_DRAIN)
#pragma unroll
            for (int g = 0; g < 36; ++g) {
                bf16x8 cb[3] = {ring[g % RPF][0], ring[g % RPF][1], ring[g % RPF][2]};
                asm volatile("" : "+v"(cb[0]), "+v"(cb[1]), "+v"(cb[2]));
                if (g + RPF < 36) fetch(g + RPF, ring[g % RPF]);
#pragma unroll
                for (int j = 0; j < 3; ++j) RB_MFMA(acc[j], a1[g >> 2][g & 3], g, cb[j]);
            }
            RB_DRAIN();
            // epilogue 1: t = bf16(relu(acc + b1)), ZERO outside the image, -> t tile [P][co].  Packed fp32 adds, one v_cvt_pk per pair,
            // ReLU on the packed bf16 pair as max_i16(., 0) (rounding keeps the sign: same result as relu before the rounding), the
            // image mask as an AND; the 12 dummy pixels of the last N-tile land in the tile's spare rows.
            f32x4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(sbias + 32 * hco + 8 * g + 4 * half);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int r = m_rc[j] & 255, c = (m_rc[j] >> 8) & 255;
                const unsigned msk = ((unsigned)(cur.ty0 - 1 + r) < (unsigned)H && (unsigned)(cur.tx0 - 1 + c) < (unsigned)W) ? 0xFFFFFFFFu : 0u;
                unsigned short* tp = tl + (32 * (3 * ph + j) + l32) * RCP + 32 * hco + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2_t v01 = f32x2_t{acc[j][4 * g + 0], acc[j][4 * g + 1]} + f32x2_t{bv[g][0], bv[g][1]};
                    const f32x2_t v23 = f32x2_t{acc[j][4 * g + 2], acc[j][4 * g + 3]} + f32x2_t{bv[g][2], bv[g][3]};
                    *reinterpret_cast<u32x2*>(tp + 8 * g) = u32x2{relu_pk(cvt_pk(v01)) & msk, relu_pk(cvt_pk(v23)) & msk};
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                        // B: t complete
        asm volatile("" ::: "memory");

        // ---- conv2 on this wave's 2 N-tiles (4 output rows): 72 MFMAs
        {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            const unsigned short* tp = tl + (o_r0 * RMW + o_c) * RCP + 8 * half;
            auto fetch = [&](int g, bf16x8 (&b)[2]) {
                const int tap = g >> 2, ks = g & 3, dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tp + ((2 * j + dy) * RMW + dx) * RCP + 16 * ks);
            };
            bf16x8 ring[RPF][2];
#pragma unroll
            for (int q = 0; q < RPF; ++q) fetch(q, ring[q]);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 36; ++g) {
                bf16x8 cb[2] = {ring[g % RPF][0], ring[g % RPF][1]};
                asm volatile("" : "+v"(cb[0]), "+v"(cb[1]));
                if (g + RPF < 36) fetch(g + RPF, ring[g % RPF]);
#pragma unroll
                for (int j = 0; j < 2; ++j) RB_MFMA(acc[j], a2[g >> 2][g & 3], 36 + g, cb[j]);
            }
            RB_DRAIN();
            // epilogue 2: c = bf16(acc + b2); y = bf16(c + x) with x from the input tile's centre -> staging [pixel][co]
            f32x4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(sbias + RC + 32 * hco + 8 * g + 4 * half);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = o_r0 + 2 * j;
                const unsigned short* xp = il + ((row + 2) * RIW + (o_c + 2)) * RCP + 32 * hco + 4 * half;
                unsigned short* sp = st + (row * RTW + o_c) * RCP + 32 * hco + 4 * half;
                u32x2 xr[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) xr[g] = *reinterpret_cast<const u32x2*>(xp + 8 * g);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned c01 = cvt_pk(f32x2_t{acc[j][4 * g + 0], acc[j][4 * g + 1]} + f32x2_t{bv[g][0], bv[g][1]});
                    const unsigned c23 = cvt_pk(f32x2_t{acc[j][4 * g + 2], acc[j][4 * g + 3]} + f32x2_t{bv[g][2], bv[g][3]});
                    *reinterpret_cast<u32x2*>(sp + 8 * g) = u32x2{cvt_pk(unpk(c01) + unpk(xr[g][0])), cvt_pk(unpk(c23) + unpk(xr[g][1]))};
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                        // C: staging complete
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0): the next tile has landed (requested two MFMA phases ago) BEFORE this
                                                             // tile's stores go out -- they retire under the next tile's products
        {
            u32x4 val[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = tid + 256 * t;
                val[t] = *reinterpret_cast<const u32x4*>(st + (e >> 3) * RCP + 8 * (e & 7));
            }
            unsigned short* ob = out + (((long long)cur.n * H + cur.ty0) * W + cur.tx0) * RC;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (cur.ty0 + sr[t] < H && cur.tx0 + sc[t] < W) *reinterpret_cast<u32x4*>(ob + go[t]) = val[t];
        }
        cur = nxt;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_resblock_fwd_bf16(int32_t N, int32_t H, int32_t W, const void* x, const void* packed_w1, const float* bias1,
                                      const void* packed_w2, const float* bias2, void* out, void* stream_) {
    if (N <= 0 || H <= 0 || W <= 0) return fail(KMSR_E_BADARG, "bf16 res-block: bad shape");
    if (!x || !packed_w1 || !packed_w2 || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (x == out) return fail(KMSR_E_BADARG, "bf16 res-block: in-place is not supported (tiles read their neighbours' halo)");
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16 || reinterpret_cast<uintptr_t>(packed_w1) % 16 ||
        reinterpret_cast<uintptr_t>(packed_w2) % 16)
        return fail(KMSR_E_BADARG, "bf16 res-block: tensors must be 16-byte aligned");
    if ((long long)H * W * RC * 2 > 0xFFFFFFF0ll) return fail(KMSR_E_UNSUPPORTED, "bf16 res-block: one image must stay below 4 GiB (32-bit buffer offsets)");
    RbParams p{};
    p.N = N; p.H = H; p.W = W;
    p.tiles_x = ceil_div(W, RTW);
    p.tiles = p.tiles_x * ceil_div(H, RTH);
    const long long items = (long long)p.tiles * N;
    if (items > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "bf16 res-block: grid too large");
    p.n_items = (int)items;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)std::min<long long>(items, cus);
    KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RLDS_BYTES));
    resblock_bf16_kernel<<<grid, 256, RLDS_BYTES, static_cast<hipStream_t>(stream_)>>>(
        static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(packed_w1), bias1, static_cast<const unsigned short*>(packed_w2),
        bias2, static_cast<unsigned short*>(out), p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
