// bf16 form of the collapsed x4 upsampler (conv5x5_ps4.hip): a 5x5 convolution on the LR grid from F channels to 16 * bands sub-pixel
// channels with PixelShuffle(4) fused into the store, for the channels-last bf16 path (BASELINE config 4: 7-band 1024x1024 LR tiles).
//
// Why: layer by layer the bf16 upsampler writes and re-reads two HR-sized feature maps (1 GB + 4.3 GB at config 4's tile: up1 0.34 ms,
// up2 1.43 ms at 0.23 of HBM, tail 0.95 ms = 2.7 of the tile's 6.06 ms, profiles/r02_config4_*).  up1 -> PS -> up2 -> PS -> tail has no
// non-linearity, so away from the border it IS one 5x5 convolution on the LR grid (weights composed on the host in float64,
// kmsr_amd/sr.py: compose_upsampler, rounded to bf16 once); the outer HR frame is recomputed layer by layer on four border strips
// (kmsr_amd/sr_bf16.py).  Per LR pixel 2*25*F*16*bands flops instead of 2*9*F*(4F + 16F + 16*bands); the output is written once.
//
// Implicit GEMM on v_mfma_f32_32x32x16_bf16: D[co, px] += A[co, k] * B[k, px], k = 16 input channels of one tap.
//   A lane l holds A[co = l&31][k = 8*(l>>5) + j], j < 8  -> 16 bytes of the packed weights [tap][co][16 ci]
//   B lane l holds B[k = 8*(l>>5) + j][px = l&31]         -> 16 bytes of the channels-last halo tile [row][col][16 ci]
//   D lane l holds px = l&31 and 16 output channels (reg&3) + 8*(reg>>2) + 4*(l>>5): registers 4g .. 4g+3 are the four HR columns
//   of one HR row of one band -> one 8-byte store per lane and group, 256 contiguous bytes per half wave.
// Block = 4 waves, tile = 8 rows x 32 LR columns x 64 sub-pixel channels (wave w: rows 2w, 2w+1 x two 32-channel blocks = 4
// accumulator tiles).  Input channels stream through LDS in chunks of 16 (halo tile 12 x 36 pixels x 32 B + the chunk's
// 25 x 64 x 32 B of weights = 65 KB, single buffer, TWO blocks per CU cover each other's loads); the next chunk's 17 16-byte
// pieces per thread are in flight in registers while the current chunk is multiplied.  LDS images are unpadded; the 16-byte unit
// of (index i, k half h) sits at 2 i + (h ^ ((i >> 3) & 1)): the ds_read_b128 lane groups then touch 16 different bank quads
// (index i = pixel column or output channel) -- conflict-free without the 50 % padding a 48-byte pitch would cost.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int VK = 5, VT = VK * VK;       // kernel size, taps
constexpr int VCK = 16;                   // input channels per chunk (one MFMA k-step)
constexpr int VTH = 8, VTW = 32;          // LR output tile per block
constexpr int VCOB = 64;                  // sub-pixel output channels per block
constexpr int VIH = VTH + VK - 1, VIW = VTW + VK - 1;                 // 12 x 36 halo tile
constexpr int V_IN_UNITS = VIH * VIW * 2;                              // 16-byte units: 864
constexpr int V_W_UNITS = VT * VCOB * 2;                               // 3200
constexpr int V_NIN = (V_IN_UNITS + 255) / 256;                        // 4 per thread
constexpr int V_NW = (V_W_UNITS + 255) / 256;                          // 13 per thread
constexpr int V_LDS_BYTES = (V_IN_UNITS + V_W_UNITS) * 16;             // 65024

struct Up16Params {
    int N, Cin, H, W, Cout;               // Cin padded to a multiple of 16; Cout = 16 * bands
    int tiles_x, tiles, cgs, nchunks, n_blocks;
};

// packed[cg][chunk][tap][co 64][16 ci] bf16  <-  w[co_g][ci_g][tap] fp32 (zero padded)
__global__ void pack5_bf16_kernel(const float* __restrict__ w, int Cin, int Cout, int nchunks, unsigned short* __restrict__ packed,
                                  long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % VCK);
        const int co = (int)((idx / VCK) % VCOB);
        const int tap = (int)((idx / (VCK * VCOB)) % VT);
        const int chunk = (int)((idx / ((long long)VCK * VCOB * VT)) % nchunks);
        const int cg = (int)(idx / ((long long)VCK * VCOB * VT * nchunks));
        const int cog = cg * VCOB + co, cig = chunk * VCK + c;
        const float v = (cog < Cout && cig < Cin) ? w[((long long)cog * Cin + cig) * VT + tap] : 0.f;
        __bf16 b = (__bf16)v;
        packed[idx] = __builtin_bit_cast(unsigned short, b);
    }
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
// 16-byte unit of (index i, k half h) inside a row of 2 * n units: bank-conflict-free ds_read_b128 without padding
__device__ __forceinline__ int swz(int i, int h) { return 2 * i + (h ^ ((i >> 3) & 1)); }

__global__ void __launch_bounds__(256, 2) conv5x5_ps4_bf16_kernel(const unsigned short* __restrict__ xg,
                                                                  const unsigned short* __restrict__ wpk,
                                                                  const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                                  Up16Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[V_LDS_BYTES + VCOB * 4];
    u32x4* lin = reinterpret_cast<u32x4*>(lds);                       // [row 12][unit 72]
    u32x4* lw = lin + V_IN_UNITS;                                     // [tap 25][unit 128]
    float* sbias = reinterpret_cast<float*>(lds + V_LDS_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware remap (as conv5x5_ps4.hip): the blocks of one XCD get a contiguous range of tiles
    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * VTH, tx0 = (tile % p.tiles_x) * VTW;
    const int H = p.H, W = p.W;

    // ---- staging: the chunk's 16-byte pieces of this thread.  Input piece e <-> (row, col, k half); weight piece e is linear.
    long long goff[V_NIN];                 // element offset of the piece inside the image for chunk 0, -1 = outside (zeros)
    int lin_unit[V_NIN];
#pragma unroll
    for (int t = 0; t < V_NIN; ++t) {
        const int e = tid + 256 * t;
        const int k = e & 1, pix = e >> 1, r = pix / VIW, col = pix - r * VIW;
        const int gy = ty0 - VK / 2 + r, gx = tx0 - VK / 2 + col;
        const bool ok = e < V_IN_UNITS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[t] = ok ? ((long long)gy * W + gx) * p.Cin + 8 * k : -1;
        lin_unit[t] = e < V_IN_UNITS ? r * (2 * VIW) + swz(col, k) : -1;
    }
    int lw_unit[V_NW];
#pragma unroll
    for (int t = 0; t < V_NW; ++t) {
        const int e = tid + 256 * t;
        const int k = e & 1, co = (e >> 1) & (VCOB - 1), tap = e >> 7;
        lw_unit[t] = e < V_W_UNITS ? tap * (2 * VCOB) + swz(co, k) : -1;
    }
    const unsigned short* __restrict__ xn = xg + (long long)n * H * W * p.Cin;
    const unsigned short* __restrict__ wcg = wpk + (long long)cg * p.nchunks * (V_W_UNITS * 8);

    u32x4 xin[V_NIN], wreg[V_NW];
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int t = 0; t < V_NIN; ++t) {
            const bool ok = goff[t] >= 0;
            const u32x4 v = *reinterpret_cast<const u32x4*>(xn + (ok ? goff[t] + chunk * VCK : 0));   // safe address + select
            xin[t] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
        const u32x4* __restrict__ wc = reinterpret_cast<const u32x4*>(wcg + (long long)chunk * (V_W_UNITS * 8));
#pragma unroll
        for (int t = 0; t < V_NW; ++t) {
            const int e = tid + 256 * t;
            wreg[t] = wc[e < V_W_UNITS ? e : 0];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int t = 0; t < V_NIN; ++t)
            if (lin_unit[t] >= 0) lin[lin_unit[t]] = xin[t];
#pragma unroll
        for (int t = 0; t < V_NW; ++t)
            if (lw_unit[t] >= 0) lw[lw_unit[t]] = wreg[t];
    };

    f32x16 acc[2][2];                      // [32-channel block][tile row of this wave]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    if (tid < VCOB) {
        const int co = cg * VCOB + tid;
        sbias[tid] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    const int nch = p.nchunks;
    load_chunk(0);
    for (int chunk = 0; chunk < nch; ++chunk) {
        __syncthreads();                   // every wave has finished reading the previous chunk
        store_chunk();
        __syncthreads();
        if (chunk + 1 < nch) load_chunk(chunk + 1);        // in flight (registers) while this chunk is multiplied
#pragma unroll
        for (int tap = 0; tap < VT; ++tap) {
            const int dy = tap / VK, dx = tap % VK;
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, lw[tap * (2 * VCOB) + swz(l32, half)]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, lw[tap * (2 * VCOB) + swz(32 + l32, half)]);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, lin[(2 * wv + dy) * (2 * VIW) + swz(l32 + dx, half)]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, lin[(2 * wv + 1 + dy) * (2 * VIW) + swz(l32 + dx, half)]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue: bias (fp32) + bf16 rounding + PixelShuffle(4) store into [N, bands, 4H, 4W] bf16 planes
    //      co = 16*band + 4*a + b  ->  out[n, band, 4y + a, 4x + b]; register 4g + j of a group holds b = j
    const int x = tx0 + l32;
    const long long W4 = 4ll * W, H4 = 4ll * H;
    const int bands = p.Cout >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int y = ty0 + 2 * wv + nt;
            const bool pok = x < W && y < H;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = mt * 32 + 8 * g + 4 * half;           // first of 4 consecutive channels (b = 0..3)
                const int co0 = cg * VCOB + col;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + col);
                const u32x2 v = {pack2bf16(acc[mt][nt][4 * g + 0] + b4.x, acc[mt][nt][4 * g + 1] + b4.y),
                                 pack2bf16(acc[mt][nt][4 * g + 2] + b4.z, acc[mt][nt][4 * g + 3] + b4.w)};
                if (pok && co0 < p.Cout) {
                    const int band = co0 >> 4, a = (co0 >> 2) & 3;
                    unsigned short* o = out + (((long long)n * bands + band) * H4 + 4 * y + a) * W4 + 4 * x;
                    *reinterpret_cast<u32x2*>(o) = v;
                }
            }
        }
    }
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv5x5_ps4_bf16_packed_weight_elems(int32_t Cin_p, int32_t Cout) {
    if (Cin_p <= 0 || Cout <= 0 || Cin_p % VCK) return 0;
    return (size_t)ceil_div(Cout, VCOB) * (Cin_p / VCK) * (V_W_UNITS * 8);
}

extern "C" int kmsr_conv5x5_ps4_bf16_pack_weights(const float* w, int32_t Cin, int32_t Cout, int32_t Cin_p, void* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0 || Cin_p < Cin || Cin_p % VCK) return fail(KMSR_E_BADARG, "conv5x5_ps4 bf16 pack: bad argument");
    const int nchunks = Cin_p / VCK;
    const long long total = (long long)ceil_div(Cout, VCOB) * nchunks * (V_W_UNITS * 8);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    pack5_bf16_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, nchunks, static_cast<unsigned short*>(packed), total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv5x5_ps4_fwd_bf16(int32_t N, int32_t Cin_p, int32_t H, int32_t W, int32_t Cout, const void* x, const void* packed_w,
                                         const float* bias, void* out, void* stream) {
    if (N <= 0 || Cin_p <= 0 || H <= 0 || W <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "bad conv5x5_ps4 shape");
    if (Cin_p % VCK) return fail(KMSR_E_SHAPE, "conv5x5_ps4 bf16 needs the channels-last channel count padded to 16 (Cin_p=%d)", Cin_p);
    if (Cout % 16) return fail(KMSR_E_SHAPE, "conv5x5_ps4 needs Cout %% 16 == 0 (Cout=%d)", Cout);
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(packed_w) % 16 || reinterpret_cast<uintptr_t>(out) % 8)
        return fail(KMSR_E_BADARG, "conv5x5_ps4 bf16 needs 16-byte aligned input / weights and an 8-byte aligned output");
    Up16Params p{};
    p.N = N; p.Cin = Cin_p; p.H = H; p.W = W; p.Cout = Cout;
    p.tiles_x = ceil_div(W, VTW);
    p.tiles = p.tiles_x * ceil_div(H, VTH);
    p.nchunks = Cin_p / VCK;
    p.cgs = ceil_div(Cout, VCOB);
    const long long nb = (long long)p.tiles * p.cgs * N;
    if (nb > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_blocks = (int)nb;
    conv5x5_ps4_bf16_kernel<<<p.n_blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(packed_w), bias, static_cast<unsigned short*>(out), p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
