// 5x5 convolution with a x4 PixelShuffle fused into the store: the collapsed form of the SR upsampler.
//
// In the EDSR-shaped SR network nothing non-linear sits between  up1 (3x3, F->4F) -> PixelShuffle(2) -> up2 (3x3,
// F->4F) -> PixelShuffle(2) -> tail (3x3, F->bands),  so away from the image border that chain IS one linear map:
// a 5x5 convolution on the LR grid from F channels to bands*16 sub-pixel channels followed by PixelShuffle(4)
// (host side: kmsr_amd/sr.py composes the 5x5 weights from the three layers and recomputes the outer 4-pixel HR frame,
// where up2's and tail's own zero padding reach, with the layer-by-layer kernels).  Per LR pixel that is
// 2*25*F*16*bands flops instead of 2*9*F*(4F + 16F + 16*bands), and the two HR-sized feature maps never exist.
//
// Same implicit-GEMM mapping as conv3x3.hip (v_mfma_f32_32x32x2_f32, block = 4 waves, tile 8 rows x 32 cols x 64
// output channels), k = (tap of 25, ci).  Input channels stream through LDS in chunks of 4 (halo tile 4 x 12 x 36 plus
// the chunk's 25 x 4 x 64 weights = 32.5 KB), double buffered, two blocks per CU.  Epilogue: the 4 registers of one
// accumulator group are 4 consecutive HR columns of one HR row, so the shuffle is a 16-byte store per group and
// a wave writes 512 contiguous bytes per instruction.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

constexpr int UK = 5, UT = UK * UK;   // kernel size, taps
constexpr int UCK = 4;                // input channels per LDS chunk
constexpr int UTH = 8, UTW = 32;      // LR output tile per block
constexpr int UCOB = 64;              // sub-pixel output channels per block
constexpr int UIH = UTH + UK - 1, UIW = UTW + UK - 1, UIWP = 36;
constexpr int U_IN_FLOATS = UCK * UIH * UIWP;
constexpr int U_W_FLOATS = UT * UCK * UCOB;
constexpr int U_BUF = U_IN_FLOATS + U_W_FLOATS;
constexpr int U_IN_ELEMS = UCK * UIH * UIW;
constexpr int U_NIN = (U_IN_ELEMS + 255) / 256;
constexpr int U_NW4 = (U_W_FLOATS / 4 + 255) / 256;
constexpr int kUFetchTap = 1, kURestageTap = 20;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct UpParams {
    int N, Cin, H, W, Cout;
    int tiles_x, tiles, cgs, nchunks, n_blocks;
    int cg0;                              // first cout group of this launch (the launch covers groups cg0 .. cg0 + cgs - 1)
};

// packed[cg][chunk][tap][c][co] <- w[co_g][ci_g][tap]  (zero padded)
__global__ void pack5_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ packed, int nchunks,
                             long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % UCOB);
        const int c = (int)((idx / UCOB) % UCK);
        const int tap = (int)((idx / (UCOB * UCK)) % UT);
        const int chunk = (int)((idx / (UCOB * UCK * UT)) % nchunks);
        const int cg = (int)(idx / ((long long)UCOB * UCK * UT * nchunks));
        const int cog = cg * UCOB + co, cig = chunk * UCK + c;
        packed[idx] = (cog < Cout && cig < Cin) ? w[((long long)cog * Cin + cig) * UT + tap] : 0.f;
    }
}

// N32 full 32-channel blocks (32x32x2 MFMA tiles) + REM * 16 channels (16x16x4 tiles) of real output channels in the cout groups
// of this launch: <2,0> = 64, <1,1> = 48 (3 bands), <1,0> = 32, <0,1> = 16 (the second group of 5 bands).  A half-empty 32-row
// block would spend 64-cycle MFMAs on zero rows; the 16-row remainder costs 4 x 32 cycles per tap instead of 4 x 64.
template <int N32, int REM>
__global__ void __launch_bounds__(256, 2) conv5x5_ps4_kernel(const float* __restrict__ xg, const float* __restrict__ wpk,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             UpParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * U_BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware remap (as conv3x3.hip): the blocks of one XCD get a contiguous range of tiles
    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = p.cg0 + bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * UTH, tx0 = (tile % p.tiles_x) * UTW;
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    int goff[U_NIN];
#pragma unroll
    for (int t = 0; t < U_NIN; ++t) {
        const int e = tid + 256 * t;
        const int c = e / (UIH * UIW), rem = e - c * (UIH * UIW), r = rem / UIW, col = rem - r * UIW;
        const int gy = ty0 - UK / 2 + r, gx = tx0 - UK / 2 + col;
        const bool ok = e < U_IN_ELEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[t] = ok ? (int)(c * HW + (long long)gy * W + gx) : -1;
    }
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const float* __restrict__ wcg = wpk + (long long)cg * p.nchunks * U_W_FLOATS;

    float xin[U_NIN];
    f32x4 wreg[U_NW4];
    auto load_chunk = [&](int chunk) {
        const float* __restrict__ xc = xn + (long long)chunk * UCK * HW;
        const int cin_left = p.Cin - chunk * UCK;
#pragma unroll
        for (int t = 0; t < U_NIN; ++t) {
            const int c = (tid + 256 * t) / (UIH * UIW);
            const bool ok = goff[t] >= 0 && c < cin_left;
            const float v = xc[ok ? goff[t] : 0];   // safe address + select: no branch, no per-load wait
            xin[t] = ok ? v : 0.f;
        }
        const f32x4* __restrict__ wc = reinterpret_cast<const f32x4*>(wcg + (long long)chunk * U_W_FLOATS);
#pragma unroll
        for (int t = 0; t < U_NW4; ++t) {
            const int e4 = tid + 256 * t;
            wreg[t] = wc[e4 < U_W_FLOATS / 4 ? e4 : 0];
        }
    };
    auto store_chunk = [&](int b) {
        float* inb = lds + b * U_BUF;
        float* wb = inb + U_IN_FLOATS;
#pragma unroll
        for (int t = 0; t < U_NIN; ++t) {
            const int e = tid + 256 * t;
            const int c = e / (UIH * UIW), rem = e - c * (UIH * UIW), r = rem / UIW, col = rem - r * UIW;
            if (e < U_IN_ELEMS) inb[(c * UIH + r) * UIWP + col] = xin[t];
        }
#pragma unroll
        for (int t = 0; t < U_NW4; ++t) {
            const int e4 = tid + 256 * t;
            if (e4 < U_W_FLOATS / 4) reinterpret_cast<f32x4*>(wb)[e4] = wreg[t];
        }
    };

    f32x16 acc[N32 > 0 ? N32 : 1][2];
#pragma unroll
    for (int m = 0; m < N32; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    f32x4 racc[2][2];                                                 // remainder tiles [px row nt][column half]: 16 channels x 16 pixels
#pragma unroll
    for (int q = 0; q < 2; ++q) racc[q][0] = racc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int aoff = half * UCOB + l32;                               // + (tap*UCK + 2*kp)*UCOB + 32*mt
    const int boff = half * (UIH * UIWP) + (2 * wv) * UIWP + l32;     // + (2*kp)*UIH*UIWP + (nt + dy)*UIWP + dx
    // remainder (v_mfma_f32_16x16x4_f32): A lane = (channel 32 N32 + l16, input channel kq), B lane = (input channel kq, pixel l16)
    const int raoff = kq * UCOB + 32 * N32 + l16;                     // + tap*UCK*UCOB
    const int rboff = kq * (UIH * UIWP) + (2 * wv) * UIWP + l16;      // + (nt + dy)*UIWP + dx + 16*ph

    const int nch = p.nchunks;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int chunk = 0; chunk < nch; ++chunk) {
        const int b = chunk & 1;
        const bool more = chunk + 1 < nch;   // block-uniform
        const float* inb = lds + b * U_BUF;
        const float* wb = inb + U_IN_FLOATS;
#pragma unroll
        for (int tap = 0; tap < UT; ++tap) {
            const int dy = tap / UK, dx = tap % UK;
            if (tap == kUFetchTap && more) load_chunk(chunk + 1);
            if (tap == kURestageTap && more) store_chunk(b ^ 1);   // the other buffer was last read before the previous barrier
            if constexpr (N32 > 0) {
#pragma unroll
                for (int kp = 0; kp < UCK / 2; ++kp) {
                    const float b0 = inb[boff + (2 * kp) * (UIH * UIWP) + dy * UIWP + dx];
                    const float b1 = inb[boff + (2 * kp) * (UIH * UIWP) + (1 + dy) * UIWP + dx];
#pragma unroll
                    for (int mt = 0; mt < N32; ++mt) {
                        const float a = wb[aoff + (tap * UCK + 2 * kp) * UCOB + 32 * mt];
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[mt][1], 0, 0, 0);
                    }
                }
            }
            if constexpr (REM) {                                      // all UCK = 4 input channels in one K = 4 step
                const float ra = wb[raoff + tap * UCK * UCOB];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        const float rb = inb[rboff + (nt + dy) * UIWP + dx + 16 * ph];
                        racc[nt][ph] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra, rb, racc[nt][ph], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias (through LDS) + PixelShuffle(4) store
    //      co = 16*band + 4*a + b  ->  out[n, band, 4y + a, 4x + b]; register r of a group holds b = r & 3
    float* sbias = lds;
    if (tid < UCOB) {
        const int co = cg * UCOB + tid;
        sbias[tid] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    __syncthreads();
    const int x = tx0 + l32;
    const long long W4 = 4ll * W, H4 = 4ll * H;
    const int bands = p.Cout >> 4;
#pragma unroll
    for (int mt = 0; mt < N32; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int y = ty0 + 2 * wv + nt;
            const bool pok = x < W && y < H;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = mt * 32 + 8 * g + 4 * half;     // first of 4 consecutive channels (b = 0..3)
                const int co0 = cg * UCOB + col;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + col);
                const f32x4 v = {acc[mt][nt][4 * g + 0] + b4.x, acc[mt][nt][4 * g + 1] + b4.y,
                                 acc[mt][nt][4 * g + 2] + b4.z, acc[mt][nt][4 * g + 3] + b4.w};
                if (pok && co0 < p.Cout) {
                    const int band = co0 >> 4, a = (co0 >> 2) & 3;
                    float* o = out + (((long long)n * bands + band) * H4 + 4 * y + a) * W4 + 4 * x;
                    *reinterpret_cast<f32x4*>(o) = v;
                }
            }
        }
    }
    if constexpr (REM) {       // D lane holds channels 32 N32 + 4 kq + r (r = sub-pixel column b), pixel l16 of its 16-pixel half row
        const int col = 32 * N32 + 4 * kq, co0 = cg * UCOB + col;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + col);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int y = ty0 + 2 * wv + nt, xx = tx0 + 16 * ph + l16;
                if (xx < W && y < H && co0 < p.Cout) {
                    const int band = co0 >> 4, a = (co0 >> 2) & 3;
                    float* o = out + (((long long)n * bands + band) * H4 + 4 * y + a) * W4 + 4 * xx;
                    *reinterpret_cast<f32x4*>(o) = racc[nt][ph] + b4;
                }
            }
    }
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv5x5_ps4_packed_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)ceil_div(Cout, UCOB) * ceil_div(Cin, UCK) * U_W_FLOATS;
}

extern "C" int kmsr_conv5x5_ps4_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, float* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "conv5x5_ps4 pack: bad argument");
    const int nchunks = ceil_div(Cin, UCK);
    const long long total = (long long)ceil_div(Cout, UCOB) * nchunks * U_W_FLOATS;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    pack5_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, packed, nchunks, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv5x5_ps4_fwd_f32(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, const float* x,
                                        const float* packed_w, const float* bias, float* out, void* stream) {
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "bad conv5x5_ps4 shape");
    if (Cout % 16) return fail(KMSR_E_SHAPE, "conv5x5_ps4 needs Cout %% 16 == 0 (Cout=%d)", Cout);
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if ((long long)Cin * H * W >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "per-image input exceeds 2^31 elements");
    UpParams p{};
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.tiles_x = ceil_div(W, UTW);
    p.tiles = p.tiles_x * ceil_div(H, UTH);
    p.nchunks = ceil_div(Cin, UCK);
    if ((long long)p.tiles * ceil_div(Cout, UCOB) * N > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int full = Cout / UCOB, rem = Cout % UCOB;        // cout groups with 64 real channels; channels of the last, partial group
    if (full) {
        p.cg0 = 0; p.cgs = full; p.n_blocks = p.tiles * full * N;
        conv5x5_ps4_kernel<2, 0><<<p.n_blocks, 256, 0, st>>>(x, packed_w, bias, out, p);
    }
    if (rem) {                                              // 16, 32 or 48 (Cout % 16 == 0)
        p.cg0 = full; p.cgs = 1; p.n_blocks = p.tiles * N;
        if (rem == 48) conv5x5_ps4_kernel<1, 1><<<p.n_blocks, 256, 0, st>>>(x, packed_w, bias, out, p);
        else if (rem == 32) conv5x5_ps4_kernel<1, 0><<<p.n_blocks, 256, 0, st>>>(x, packed_w, bias, out, p);
        else conv5x5_ps4_kernel<0, 1><<<p.n_blocks, 256, 0, st>>>(x, packed_w, bias, out, p);
    }
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
