// bf16 3x3 convolution, channels-last, on v_mfma_f32_32x32x16_bf16 (semantics: include/kmsr.h).
//
//   D[co, px] += A[co, k] * B[k, px],  k = 16 consecutive input channels of one tap.
//   A lane l holds A[co = l&31][k = 8*(l>>5) + j], j < 8  -> 16 bytes of the packed weights [tap][co][ci]
//   B lane l holds B[k = 8*(l>>5) + j][px = l&31]         -> 16 bytes of the channels-last input tile [row][col][ci]
//   D lane l holds px = l&31 and 16 output channels (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// Persistent workgroups (4 waves, one per CU) stream 8x32-pixel tiles of one 64-channel output group: halo tile [10][34][Cin] bf16
// in LDS (pixel rows padded by 16 B against bank conflicts), 9 taps x Cin/16 MFMA steps, epilogue (bias, ReLU, bf16 rounding)
// written to an LDS staging tile and stored as full channel rows (16 B per lane, consecutive lanes consecutive addresses),
// residual added in fp32 on the way out.  PixelShuffle(2) is a permutation of the staging tile: conv channel 4c+2i+j of pixel
// (y,x) lands in channel c of pixel (2y+i, 2x+j).  Round 1's kernel kept the weight slab in LDS and ran the tile loop serially
// (load, barrier, MFMA, barrier, epilogue, barrier, store); the kernel below replaced it (history: profiles/r01_config4_bf16_timing.log).
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BTH = 8, BTW = 32;          // output tile (rows x cols)
constexpr int BIH = BTH + 2, BIW = BTW + 2;
constexpr int BCO = 64;                   // output channels per workgroup pass
constexpr int PADE = 8;                   // row padding in bf16 elements (16 bytes): conflict-free 16-B reads at 128-B strides

struct Bf16Params {
    int N, Cin, H, W, Cout;               // padded channel counts
    int tiles_x, tiles, cgs, n_items, relu;
};

__device__ __forceinline__ unsigned short f2bf(float f) {   // round-to-nearest-even (NaN stays NaN via the hardware cast)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
// two values -> one packed dword with ONE v_cvt_pk_bf16_f32 (two scalar casts cost cvt, cvt, shift, or)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// packed[cg][tap][co 64][Cin_p] bf16  <-  w[co][ci][tap] fp32
// tf != 0: the data-gradient convolution's weights w'[ci][co][u][v] = w[co][ci][2-u][2-v] (the packed conv then maps the
// ORIGINAL Cout channels to the original Cin channels; Cin_p / Cout_p are the padded input / output counts of THAT conv)
__global__ void bf16_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int Cin_p, int Cout_p, int tf,
                                 unsigned short* __restrict__ packed, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Cin_p);
        const int co = (int)((idx / Cin_p) % BCO);
        const int tap = (int)((idx / ((long long)Cin_p * BCO)) % 9);
        const int cg = (int)(idx / ((long long)Cin_p * BCO * 9));
        const int cog = cg * BCO + co;
        float v = 0.f;
        if (tf) {
            if (cog < Cin && ci < Cout) v = w[((long long)ci * Cin + cog) * 9 + (8 - tap)];
        } else if (cog < Cout && ci < Cin) {
            v = w[((long long)cog * Cin + ci) * 9 + tap];
        }
        packed[idx] = f2bf(v);
    }
}

// [N,H,W,Cp] bf16 (channels-last) -> [N,Cp,H,W] fp32 planes: the weight gradient's operands (K = pixels must be contiguous
// per lane for every MFMA shape, which channels-last bf16 is not; the products of bf16 values are exact in fp32).
// thread = one pixel, loops over the 8-channel groups of its row; a store instruction writes 64 consecutive pixels of one plane.
__global__ void __launch_bounds__(256) nhwc_bf16_to_nchw_f32_kernel(const unsigned short* __restrict__ in, float* __restrict__ out,
                                                                    int Cp, long long HW, long long total_px) {
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    if (px >= total_px) return;
    const long long n = px / HW, q = px - n * HW;
    const u32x4* src = reinterpret_cast<const u32x4*>(in + px * Cp);
    float* dst = out + n * Cp * HW + q;
    for (int g = 0; g < Cp / 8; ++g) {
        const u32x4 v = src[g];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[(long long)(8 * g + 2 * j) * HW] = __builtin_bit_cast(float, v[j] << 16);
            dst[(long long)(8 * g + 2 * j + 1) * HW] = __builtin_bit_cast(float, v[j] & 0xffff0000u);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Register-weight, double-buffered tile loop (Cin <= 64).  A serial tile loop pays two HBM round trips per tile (input tile,
// residual) that nothing overlaps -- one workgroup of four waves per CU is all an 83 KB weight slab in LDS leaves room for.  Here
//   * every wave keeps ITS A operands -- the whole 64 x 9 x Cin slab, 16 B per lane and (tap, k-step, co half): 288 VGPRs at
//     Cin = 64 -- in registers (one wave per SIMD owns all 512 VGPRs anyway), which frees the LDS for two input buffers plus a
//     separate staging tile and halves the LDS operand traffic of the MFMA phase;
//   * the NEXT item's halo tile travels HBM -> LDS by LDS-DMA while the current one is multiplied: the LDS image keeps the padded
//     pixel pitch (Cin + 8 elements), every 9th 16-byte lane of the dense DMA stream is a pad lane with an out-of-range buffer offset (the hardware writes zeros), as
//     do pixels outside the image;
//   * the residual pieces are requested before the MFMA phase and consumed after it; two raw barriers per tile instead of four.
// ------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;

template <int KS, int NW = 4>
struct RwGeo {
    static constexpr int Cin = 16 * KS, CP = Cin + PADE, kvec = Cin / 8, UP = kvec + 1;     // 16-byte units per pixel incl. the pad unit
    static constexpr int NPIX = BIH * BIW, NUNITS = NPIX * UP, NINSTR = (NUNITS + 63) / 64, NIW = (NINSTR + NW - 1) / NW;
    static constexpr int TILE_ELEMS = NIW * NW * 64 * 8;                                     // bf16 elements of one input buffer
    static constexpr int ST_ELEMS = (2 * BTH) * (2 * BTW) * (16 + PADE) > BTH * BTW * (BCO + PADE) ? (2 * BTH) * (2 * BTW) * (16 + PADE)
                                                                                                  : BTH * BTW * (BCO + PADE);
    static constexpr size_t LDS_BYTES = (size_t)(2 * TILE_ELEMS + ST_ELEMS) * 2 + BCO * sizeof(float);
};

struct Origin { int n, ty0, tx0, cg; };      // image, tile origin and output-channel group of a work item

// MT = 32-channel output blocks actually computed (1 for layers with <= 32 output channels, e.g. the 7-band tail: half the MFMAs).
// NW = waves per workgroup: 4 (wave w owns tile rows 2w, 2w+1 and all MT blocks) or 8 (MT = 2 only: waves w and w + 4 own the same
// rows and ONE 32-channel block each -- half the weight slab per wave, so two waves fit a SIMD and cover each other's LDS / VALU
// latencies in the epilogue and store phases; the matrix pipe sees the same MFMAs).
template <bool PS, int KS, bool RELU, int MT, int NW>
__global__ void __launch_bounds__(64 * NW, 1) conv3x3_bf16_rw_kernel(const unsigned short* __restrict__ xg,
                                                                 const unsigned short* __restrict__ wpk,
                                                                 const float* __restrict__ bias,
                                                                 const unsigned short* __restrict__ res,
                                                                 unsigned short* __restrict__ out, Bf16Params p) {
    using G = RwGeo<KS, NW>;
    constexpr int NT = 64 * NW;                            // threads
    constexpr int MTW = NW == 8 ? 1 : MT;                  // 32-channel blocks per wave
    static_assert(NW == 4 || (NW == 8 && MT == 2), "8 waves split the two channel blocks");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];        // the ONLY LDS object: in0 | in1 | staging | bias
    constexpr int Cin = G::Cin, CP = G::CP, kvec = G::kvec, UP = G::UP;
    unsigned short* st = smem + 2 * G::TILE_ELEMS;
    float* sbias = reinterpret_cast<float*>(st + G::ST_ELEMS);
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = NW == 8 ? (wv & 3) : wv, mt0 = NW == 8 ? (wv >> 2) : 0;      // row pair and first channel block of this wave
    const int H = p.H, W = p.W;
    const int per_cg = p.N * p.tiles;

    const int per = (p.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
    int item = blockIdx.x * per;
    const int item_end = min(item + per, p.n_items);
    if (item >= item_end) return;

    // Lane constants of the DMA stream: unit uidx = i * 64 + lane is piece `uidx % UP` of tile pixel `uidx / UP` (piece kvec = the pad
    // unit).  The loads are buffer loads: a lane whose pixel lies outside the image (or that is a pad / tail lane) gets an out-of-range
    // offset and the hardware writes ZEROS to its LDS slot (tools/ubench/buffer_lds_oob.hip) -- the conv's zero padding for free.
    int ur[G::NIW], uc[G::NIW], urel[G::NIW];
#pragma unroll
    for (int k = 0; k < G::NIW; ++k) {
        const int uidx = (wv * G::NIW + k) * 64 + lane, pix = uidx / UP, piece = uidx - pix * UP, r = pix / BIW, c = pix - r * BIW;
        const bool live = pix < G::NPIX && piece < kvec;
        ur[k] = live ? r : -100000;                         // dead lanes fail the row check of every tile
        uc[k] = c;
        urel[k] = ((r - 1) * W + (c - 1)) * (Cin * 2) + 16 * piece;      // bytes from the tile's first OUTPUT pixel
    }
    auto issue_tile = [&](const Origin& o, int buf) {     // halo tile of the item at `o` -> input buffer `buf`, by LDS-DMA
        const int n = o.n, ty0 = o.ty0, tx0 = o.tx0;
        const unsigned nrec = (unsigned)H * (unsigned)W * (unsigned)(Cin * 2);       // bytes of one image (< 4 GiB, host-checked)
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xg + (long long)n * H * W * Cin), 0, nrec, 0x00020000);
        const unsigned tbase = ((unsigned)ty0 * (unsigned)W + (unsigned)tx0) * (unsigned)(Cin * 2);      // wave-uniform
#pragma unroll
        for (int k = 0; k < G::NIW; ++k) {
            const int i = wv * G::NIW + k;                  // wave-uniform instruction index
            if (i < G::NINSTR) {
                const bool ok = (unsigned)(ty0 - 1 + ur[k]) < (unsigned)H && (unsigned)(tx0 - 1 + uc[k]) < (unsigned)W;
                const unsigned off = ok ? tbase + (unsigned)urel[k] : nrec;      // modulo-2^32 sum; nrec itself is out of range
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + buf * G::TILE_ELEMS + i * 512), 16, off, 0, 0, 0);
            }
        }
    };

    // store / residual pieces of a thread: tile-invariant element offsets from the tile's first pixel (32-bit: a tile spans < 2^31)
    constexpr int NST = BTH * BTW * 8 / NT;
    int go[NST], sr[NST], sc[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int e = tid + NT * t, pix = e >> 3;
        sr[t] = pix / BTW;
        sc[t] = pix % BTW;
        go[t] = (sr[t] * W + sc[t]) * p.Cout + 8 * (e & 7);
    }
    auto tile_origin = [&](int it) {                        // a few scalar divisions: once per item, never per piece
        Origin o;
        o.cg = it / per_cg;
        const int rem = it - o.cg * per_cg;
        o.n = rem / p.tiles;
        const int tile = rem - o.n * p.tiles;
        o.ty0 = (tile / p.tiles_x) * BTH;
        o.tx0 = (tile % p.tiles_x) * BTW;
        return o;
    };
    // residual piece t of the item at `o` -> rv[t] (a piece outside the image or past the channel tail reads the tile's first piece instead)
    u32x4 rv[NST];
    auto residual_piece = [&](const Origin& o, const unsigned short* __restrict__ rb, int pieces, int t) {
        const bool okp = ((tid + NT * t) & 7) < pieces && o.ty0 + sr[t] < H && o.tx0 + sc[t] < W;
        // non-temporal: the skip tensor is touched once per launch and should not displace what the next launch reads (config-4 tile 4.13-4.36 -> 4.0 ms)
        rv[t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rb + (okp ? go[t] : 0)));
    };
    auto residual_base = [&](const Origin& o) { return res + (((long long)o.n * H + o.ty0) * W + o.tx0) * p.Cout + o.cg * BCO; };

    bf16x8 areg[9][KS][MTW];
    int cur_cg = -1, buf = 0;
    Origin cur = tile_origin(item), nxt_carry = cur;
    issue_tile(cur, 0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);                          // the first tile (every later one is waited for before the stores of its predecessor)
    for (; item < item_end; ++item, buf ^= 1, cur = nxt_carry) {
        const int cg = cur.cg, n = cur.n, ty0 = cur.ty0, tx0 = cur.tx0;
        const bool more = item + 1 < item_end;
        const Origin nxt = more ? tile_origin(item + 1) : cur;
        nxt_carry = nxt;
        if (cg != cur_cg) {                                 // this output group's weights -> registers, bias -> LDS
            // PixelShuffle: MFMA row i = 8g + 4h + r of a 32-row block multiplies with the weights of conv channel 16h + 4r + g, so
            // that a lane's 4 consecutive accumulators are 4 ADJACENT shuffled channels (c = 4h + r) of ONE sub-pixel (g = 2i + j):
            // one 8-byte staging write per group instead of four 2-byte ones
            const int arow = PS ? 16 * ((l32 >> 2) & 1) + 4 * (l32 & 3) + (l32 >> 3) : l32;
            const unsigned short* __restrict__ wsrc = wpk + (long long)cg * 9 * BCO * Cin + (long long)arow * Cin + 8 * half;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        areg[tap][ks][mt] = *reinterpret_cast<const bf16x8*>(wsrc + (tap * BCO + 32 * (mt0 + mt)) * Cin + 16 * ks);
        }
        // No vmcnt wait here: this item's tile was waited for before the previous item's stores went out (below), so those stores
        // -- which count on vmcnt too -- retire under this item's MFMA phase instead of stalling its start.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): every LDS read of the previous item retired
        if (cg != cur_cg) {
            if (cur_cg >= 0) __builtin_amdgcn_s_barrier();  // previous item's bias readers are done
            if (tid < BCO) sbias[tid] = (bias && cg * BCO + tid < p.Cout) ? bias[cg * BCO + tid] : 0.f;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            cur_cg = cg;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) issue_tile(nxt, buf ^ 1);                 // that buffer's readers finished before the barrier above
        if constexpr (!PS) {
            if (res) {                                      // this item's residual pieces: in flight during the MFMA phase
                const unsigned short* rb = residual_base(cur);
                const int pc = min(BCO, p.Cout - cg * BCO) / 8;
#pragma unroll
                for (int t = 0; t < NST; ++t) residual_piece(cur, rb, pc, t);
            }
        }

        const long long tile_base = (((long long)n * H + ty0) * W + tx0) * p.Cout + cg * BCO;      // wave-uniform

        f32x16 acc[MTW][2];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
        const unsigned short* il = smem + buf * G::TILE_ELEMS;
        const unsigned short* bp = il + ((2 * rp) * BIW + l32) * CP + 8 * half;       // + ((nt+dy)*BIW + dx)*CP + 16*ks
        constexpr int NG = 9 * KS;
        auto fetch_b = [&](int g, bf16x8& b0, bf16x8& b1) {
            const int tap = g / KS, ks = g % KS, dy = tap / 3, dx = tap % 3;
            b0 = *reinterpret_cast<const bf16x8*>(bp + (dy * BIW + dx) * CP + 16 * ks);
            b1 = *reinterpret_cast<const bf16x8*>(bp + ((1 + dy) * BIW + dx) * CP + 16 * ks);
        };
        bf16x8 cb0, cb1;
        fetch_b(0, cb0, cb1);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            bf16x8 nb0 = cb0, nb1 = cb1;
            if (g + 1 < NG) fetch_b(g + 1, nb0, nb1);       // one group ahead: with one wave per SIMD nothing else covers an LDS round trip
            asm volatile("" : "+v"(cb0), "+v"(cb1));
            const int tap = g / KS, ks = g % KS;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[tap][ks][mt], cb0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[tap][ks][mt], cb1, acc[mt][1], 0, 0, 0);
            }
            cb0 = nb0; cb1 = nb1;
        }

        // ---- epilogue into the staging tile (its own LDS region: no barrier before it -- the previous item's staging reads ended
        //      before this item's first barrier).  Non-PS: st[(row*32 + col)][64 co].  PS: st[((2row+i)*64 + 2col+j)][16 c]
        constexpr int SP = PS ? 16 + PADE : BCO + PADE;
        // the 32 bias values of this lane in ONE batch of LDS reads (cycle stamps, tools/diag/make_stamped_bf16.py: read -> wait ->
        // add per group of 4 channels exposed the LDS latency 16 times, 1900 cycles for 170 instructions)
        f32x4 bv[MTW][4];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (PS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[mt][g][r] = sbias[(mt0 + mt) * 32 + 16 * half + 4 * r + g];
                } else {
                    bv[mt][g] = *reinterpret_cast<const f32x4*>(sbias + (mt0 + mt) * 32 + 8 * g + 4 * half);
                }
            }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) asm volatile("" : "+v"(bv[mt][0]), "+v"(bv[mt][1]), "+v"(bv[mt][2]), "+v"(bv[mt][3]));
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int row = 2 * rp + nt;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co0 = (mt0 + mt) * 32 + 8 * g + 4 * half;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][nt][4 * g + r] + bv[mt][g][r];
                        if constexpr (RELU) v[r] = fmaxf(v[r], 0.f);
                    }
                    if constexpr (PS) {
                        // shuffled channels c = 8 mt + 4 half + r of sub-pixel (i, j) = (g >> 1, g & 1)
                        *reinterpret_cast<uint2*>(st + ((2 * row + (g >> 1)) * (2 * BTW) + 2 * l32 + (g & 1)) * SP + 8 * (mt0 + mt) + 4 * half) =
                            make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    } else {
                        *reinterpret_cast<uint2*>(st + (row * BTW + l32) * SP + co0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    }
                }
            }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): staging complete
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0): the next tile (requested a whole MFMA phase ago) and the residual pieces
        if constexpr (PS) {
            const int Cq = p.Cout >> 2;
            const int H2 = 2 * H, W2 = 2 * W;
            constexpr int NPS = (2 * BTH) * (2 * BTW) * 2 / NT;    // 8 pieces per thread: all staging reads first, then the stores
            u32x4 val[NPS];
#pragma unroll
            for (int t = 0; t < NPS; ++t) {
                const int e = tid + NT * t;
                val[t] = *reinterpret_cast<const u32x4*>(st + (e >> 1) * SP + 8 * (e & 1));
            }
#pragma unroll
            for (int t = 0; t < (int)(sizeof(val) / sizeof(val[0])); t += 4) asm volatile("" : "+v"(val[t]), "+v"(val[t + 1]), "+v"(val[t + 2]), "+v"(val[t + 3]));
            unsigned short* ob = out + (((long long)n * H2 + 2 * ty0) * W2 + 2 * tx0) * Cq + cg * 16;      // wave-uniform
#pragma unroll
            for (int t = 0; t < NPS; ++t) {
                const int e = tid + NT * t;
                const int piece = e & 1, pix = e >> 1, r = pix / (2 * BTW), c = pix % (2 * BTW);
                if (2 * ty0 + r < H2 && 2 * tx0 + c < W2) *reinterpret_cast<u32x4*>(ob + (r * W2 + c) * Cq + 8 * piece) = val[t];
            }
        } else {
            const int pieces = min(BCO, p.Cout - cg * BCO) / 8;
            // batches (stamps: one piece at a time -- LDS read, wait, add, store -- took 2100 cycles without and 4000 with the residual):
            // all staging reads, then the residual adds, then the stores, then the next item's residual requests
            u32x4 val[NST];
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                const int e = tid + NT * t;
                val[t] = *reinterpret_cast<const u32x4*>(st + (e >> 3) * SP + 8 * (e & 7));
            }
#pragma unroll
            for (int t = 0; t < (int)(sizeof(val) / sizeof(val[0])); t += 4) asm volatile("" : "+v"(val[t]), "+v"(val[t + 1]), "+v"(val[t + 2]), "+v"(val[t + 3]));
            if (res) {
#pragma unroll
                for (int t = 0; t < NST; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {           // low halves by a shift, high halves by a mask: 4 + 2 + 1 instructions per pair
                        const float a0 = __builtin_bit_cast(float, val[t][q] << 16) + __builtin_bit_cast(float, rv[t][q] << 16);
                        const float a1 = __builtin_bit_cast(float, val[t][q] & 0xffff0000u) + __builtin_bit_cast(float, rv[t][q] & 0xffff0000u);
                        val[t][q] = pack2bf(a0, a1);
                    }
            }
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                const bool okt = ((tid + NT * t) & 7) < pieces && ty0 + sr[t] < H && tx0 + sc[t] < W;
                if (okt) *reinterpret_cast<u32x4*>(out + tile_base + go[t]) = val[t];
            }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv3x3_bf16_packed_weight_elems(int32_t Cin_p, int32_t Cout_p) {
    if (Cin_p <= 0 || Cout_p <= 0) return 0;
    return (size_t)ceil_div(Cout_p, BCO) * 9 * BCO * Cin_p;
}

extern "C" int kmsr_conv3x3_bf16_pack_weights(const float* w, int32_t Cin, int32_t Cout, int32_t Cin_p, int32_t Cout_p,
                                              void* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0 || Cin_p < Cin || Cout_p < Cout || Cin_p % 16)
        return fail(KMSR_E_BADARG, "bf16 pack: bad argument (Cin_p must be a multiple of 16 and >= Cin)");
    const long long total = (long long)kmsr_conv3x3_bf16_packed_weight_elems(Cin_p, Cout_p);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    bf16_pack_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, Cin_p, Cout_p, 0,
                                                                          static_cast<unsigned short*>(packed), total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv3x3_bf16_pack_weights_dgrad(const float* w, int32_t Cin, int32_t Cout, int32_t Cin_p, int32_t Cout_p,
                                                    void* packed, void* stream) {
    // the gradient convolution reads Cout_p channels and writes Cin_p channels
    if (!w || !packed || Cin <= 0 || Cout <= 0 || Cin_p < Cin || Cout_p < Cout || Cout_p % 16)
        return fail(KMSR_E_BADARG, "bf16 dgrad pack: bad argument (Cout_p must be a multiple of 16 and >= Cout)");
    const long long total = (long long)kmsr_conv3x3_bf16_packed_weight_elems(Cout_p, Cin_p);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    bf16_pack_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, Cout_p, Cin_p, 1,
                                                                          static_cast<unsigned short*>(packed), total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv3x3_fwd_bf16(const kmsr_conv3x3_desc* d, const void* x, const void* packed_w, const float* bias,
                                     const void* residual, void* out, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "bf16 conv: stride 1 only");
    if (d->Cin % 16 || d->Cin > 64) return fail(KMSR_E_UNSUPPORTED, "bf16 conv: padded Cin must be 16, 32, 48 or 64 (got %d)", d->Cin);
    if (d->Cout % 16) return fail(KMSR_E_UNSUPPORTED, "bf16 conv: padded Cout must be a multiple of 16 (got %d)", d->Cout);
    if (d->pixel_shuffle && (d->Cout % 64)) return fail(KMSR_E_SHAPE, "bf16 conv: pixel_shuffle needs Cout %% 64 == 0");
    if (!d->pixel_shuffle && (d->Cout % 64) && d->Cout > 64) return fail(KMSR_E_UNSUPPORTED, "bf16 conv: Cout > 64 must be a multiple of 64");
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(out) % 16 || reinterpret_cast<uintptr_t>(residual) % 16)
        return fail(KMSR_E_BADARG, "bf16 conv: tensors must be 16-byte aligned");
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if ((long long)d->H * d->W * d->Cin * 2 > 0xFFFFFFF0ll)
        return fail(KMSR_E_UNSUPPORTED, "bf16 conv: one image must stay below 4 GiB (32-bit buffer offsets)");
    Bf16Params p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.relu = d->relu;
    p.tiles_x = ceil_div(d->W, BTW);
    p.tiles = p.tiles_x * ceil_div(d->H, BTH);
    p.cgs = ceil_div(d->Cout, BCO);
    const long long items = (long long)p.tiles * p.cgs * d->N;
    if (items > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_items = (int)items;
    const int grid = (int)std::min<long long>(items, 256);
    const int ks = d->Cin / 16;
    const bool w8 = d->Cout > 32;                   // two 32-channel blocks: 8 waves, one block per wave
    const size_t shmem = ks == 1 ? (w8 ? RwGeo<1, 8>::LDS_BYTES : RwGeo<1, 4>::LDS_BYTES)
                       : ks == 2 ? (w8 ? RwGeo<2, 8>::LDS_BYTES : RwGeo<2, 4>::LDS_BYTES)
                       : ks == 3 ? (w8 ? RwGeo<3, 8>::LDS_BYTES : RwGeo<3, 4>::LDS_BYTES)
                                 : (w8 ? RwGeo<4, 8>::LDS_BYTES : RwGeo<4, 4>::LDS_BYTES);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const auto* xs = static_cast<const unsigned short*>(x);
    const auto* ws = static_cast<const unsigned short*>(packed_w);
    const auto* rs = static_cast<const unsigned short*>(residual);
    auto* os = static_cast<unsigned short*>(out);
    auto launch = [&](auto kern, int threads) -> int {
        KMSR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        kern<<<grid, threads, shmem, stream>>>(xs, ws, bias, rs, os, p);
        return KMSR_OK;
    };
    int rc = KMSR_OK;
#define KMSR_BF16_LAUNCH(PS_, RL_, MT_, NW_)                                                        \
    switch (ks) {                                                                                   \
        case 1: rc = launch(conv3x3_bf16_rw_kernel<PS_, 1, RL_, MT_, NW_>, 64 * NW_); break;        \
        case 2: rc = launch(conv3x3_bf16_rw_kernel<PS_, 2, RL_, MT_, NW_>, 64 * NW_); break;        \
        case 3: rc = launch(conv3x3_bf16_rw_kernel<PS_, 3, RL_, MT_, NW_>, 64 * NW_); break;        \
        default: rc = launch(conv3x3_bf16_rw_kernel<PS_, 4, RL_, MT_, NW_>, 64 * NW_); break;       \
    }
    if (d->pixel_shuffle) {
        if (d->relu) { KMSR_BF16_LAUNCH(true, true, 2, 8) } else { KMSR_BF16_LAUNCH(true, false, 2, 8) }
    } else if (d->Cout <= 32) {                     // one 32-channel block is enough (the 7-band tail: Cout_p = 16)
        if (d->relu) { KMSR_BF16_LAUNCH(false, true, 1, 4) } else { KMSR_BF16_LAUNCH(false, false, 1, 4) }
    } else {
        if (d->relu) { KMSR_BF16_LAUNCH(false, true, 2, 8) } else { KMSR_BF16_LAUNCH(false, false, 2, 8) }
    }
#undef KMSR_BF16_LAUNCH
    if (rc != KMSR_OK) return rc;
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

// ---- backward legs (kmsr_conv3x3_{dgrad,wgrad}_bf16)
extern "C" int kmsr_conv3x3_dgrad_bf16(const kmsr_conv3x3_desc* d, const void* dy, const void* packed_wT, void* dx, void* stream) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    kmsr_conv3x3_desc g = *d;                  // the data gradient is the same convolution from Cout to Cin channels
    g.Cin = d->Cout;
    g.Cout = d->Cin;
    g.relu = 0;
    g.pixel_shuffle = 0;
    g.flags = 0;
    if (d->pixel_shuffle) return fail(KMSR_E_UNSUPPORTED, "bf16 dgrad: un-shuffle the gradient first (pixel_shuffle forward)");
    return kmsr_conv3x3_fwd_bf16(&g, dy, packed_wT, nullptr, nullptr, dx, stream);
}

extern "C" size_t kmsr_conv3x3_wgrad_bf16_workspace_bytes(const kmsr_conv3x3_desc* d) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return 0;
    const size_t planes = (size_t)d->N * ((size_t)d->Cin + d->Cout) * d->H * d->W * sizeof(float);
    return planes + kmsr_conv3x3_wgrad_workspace_bytes(d);
}

extern "C" int kmsr_conv3x3_wgrad_bf16(const kmsr_conv3x3_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->Cin % 8 || d->Cout % 8) return fail(KMSR_E_UNSUPPORTED, "bf16 wgrad: padded channel counts must be multiples of 8");
    if (!x || !dy || !dw || !workspace) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(dy) % 16) return fail(KMSR_E_BADARG, "bf16 wgrad: tensors must be 16-byte aligned");
    const size_t need = kmsr_conv3x3_wgrad_bf16_workspace_bytes(d);
    if (workspace_bytes < need) return fail(KMSR_E_NOMEM, "bf16 wgrad workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long long HW = (long long)d->H * d->W, total_px = (long long)d->N * HW;
    float* xf = static_cast<float*>(workspace);
    float* dyf = xf + (size_t)d->N * d->Cin * HW;
    float* ws = dyf + (size_t)d->N * d->Cout * HW;
    const unsigned blocks = (unsigned)((total_px + 255) / 256);
    nhwc_bf16_to_nchw_f32_kernel<<<blocks, 256, 0, stream>>>(static_cast<const unsigned short*>(x), xf, d->Cin, HW, total_px);
    nhwc_bf16_to_nchw_f32_kernel<<<blocks, 256, 0, stream>>>(static_cast<const unsigned short*>(dy), dyf, d->Cout, HW, total_px);
    KMSR_HIP_CHECK(hipGetLastError());
    kmsr_conv3x3_desc f = *d;
    f.stride = 1;
    f.flags = 0;
    return kmsr_conv3x3_wgrad_f32(&f, xf, dyf, dw, ws, kmsr_conv3x3_wgrad_workspace_bytes(&f), stream_);
}
