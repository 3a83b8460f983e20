// Winograd F(2x2,3x3) forward convolution on the fp32 matrix cores (semantics: include/kmsr.h).
//
//   V_xi[ci, tile] = (B^T d B)_xi      input transform of the 4x4 patch of every 2x2 output tile   (xi = 0..15)
//   U_xi[co, ci]   = (G g G^T)_xi      weight transform, done once by the pack kernel
//   M_xi[co, tile] = sum_ci U_xi[co,ci] * V_xi[ci,tile]       16 independent GEMMs -> v_mfma_f32_32x32x2_f32
//   Y[co, tile]    = A^T M A           2x2 outputs per tile
//
// Workgroup = 4 waves = 64 output channels x 64 tiles (2 tile rows x 32 tile cols = 4 x 64 output pixels).  Wave w owns
// (co block w&1, tile row w>>1) and ALL 16 xi: 16 accumulator tiles = 256 AGPRs (one workgroup per CU, one wave per
// SIMD), so for every (co, tile) the 16 M values sit in ONE lane at one register index and A^T M A is lane-local.
// Input channels stream in chunks of 16: U chunk 16x16x64 and V chunk 16x16x64 floats (64 KB each) in ONE LDS stage.
// A chunk's MFMA phase (128 MFMAs per wave, ~3.7 us) is long enough for the next chunk's raw 4x4 patches and U rows
// (register staged, issued right after the first MFMAs) to land; they are transformed and written between two
// barriers.  (A double-buffered 8-channel variant was slower: its 1.9 us phases are shorter than a global round trip.)
#include "kmsr_common.h"
#ifndef KMSR_WINO_ABLATE
#define KMSR_WINO_ABLATE 0  // diagnostic builds: 1 no global loads in loop, 2 no MFMA, 3 no restage, 4 no epilogue stores
#endif

#include <algorithm>

namespace kmsr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int WCK = 16;                // input channels per chunk
constexpr int WCO = 64;                // output channels per workgroup
constexpr int TROWS = 2, TCOLS = 32;   // tiles per workgroup: 2 x 32  (output 4 rows x 64 cols)
constexpr int NT = TROWS * TCOLS;      // 64 tiles
constexpr int UF = 16 * WCK * WCO;     // 8192 floats
constexpr int VF = 16 * WCK * NT;      // 8192 floats
constexpr int WBUF = UF + VF;

struct WinoParams {
    int N, Cin, H, W, Cout;            // stride 1: Ho = H, Wo = W
    int tiles_x, tiles, cgs, nchunks, relu, n_blocks;
};

// packed[cg][chunk][xi][c][co] = (G g G^T)[xi] for (co, ci) = (cg*64+co, chunk*8+c), zero padded
__global__ void wino_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ packed, int nchunks,
                                 long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % WCO);
        const int c = (int)((idx / WCO) % WCK);
        const int xi = (int)((idx / (WCO * WCK)) % 16);
        const int chunk = (int)((idx / (WCO * WCK * 16)) % nchunks);
        const int cg = (int)(idx / ((long long)WCO * WCK * 16 * nchunks));
        const int cog = cg * WCO + co, cig = chunk * WCK + c;
        float v = 0.f;
        if (cog < Cout && cig < Cin) {
            const float* g = w + ((long long)cog * Cin + cig) * 9;
            // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  U = G g G^T
            const int i = xi >> 2, j = xi & 3;
            float row[3];   // (G g)[i][:]
            for (int q = 0; q < 3; ++q) {
                const float g0 = g[0 * 3 + q], g1 = g[1 * 3 + q], g2 = g[2 * 3 + q];
                row[q] = i == 0 ? g0 : (i == 1 ? 0.5f * (g0 + g1 + g2) : (i == 2 ? 0.5f * (g0 - g1 + g2) : g2));
            }
            v = j == 0 ? row[0] : (j == 1 ? 0.5f * (row[0] + row[1] + row[2]) : (j == 2 ? 0.5f * (row[0] - row[1] + row[2]) : row[2]));
        }
        packed[idx] = v;
    }
}

template <bool PS>
__global__ void __launch_bounds__(256, 1) conv3x3_wino_kernel(const float* __restrict__ xg, const float* __restrict__ wpk,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res, float* __restrict__ out,
                                                              WinoParams p) {
    __shared__ __attribute__((aligned(16))) float lds[WBUF];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cbk = wv & 1, tb = wv >> 1;

    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * (2 * TROWS), tx0 = (tile % p.tiles_x) * (2 * TCOLS);
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    // ---- input staging: thread = (tile t = tid & 63, channel sub-lane cs = tid >> 6); channels cs + 4j (j < 4) of the chunk
    const int st = tid & 63, cs = tid >> 6, str = st >> 5, stc = st & 31;
    // The 4x4 patch of tile (str, stc) covers rows ty0+2str-1 .. +2 and columns c-1 .. c+2 with c = tx0 + 2 stc (even).
    // Each row is fetched as three ALIGNED float2 (c-2,c-1) (c,c+1) (c+2,c+3): neighbouring lanes are exactly 8 bytes
    // apart, so every load instruction is one contiguous 512-byte request (a dword gather with an 8-byte lane stride
    // costs the texture-address path twice as many instructions at half the efficiency).  W is even on this path, so
    // an aligned pair is entirely inside or entirely outside the image.
    const int py0 = ty0 + 2 * str - 1, pc0 = tx0 + 2 * stc - 2;
    int poff[12];       // [row r][pair k]: offset of the pair inside one channel plane, -1 = outside (zero padding)
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int y = py0 + e / 3, x = pc0 + 2 * (e % 3);
        poff[e] = (y >= 0 && y < H && x >= 0 && x + 1 < W) ? y * W + x : -1;
    }
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const float* __restrict__ wcg = wpk + (long long)cg * p.nchunks * UF;

    constexpr int NP = WCK / 4;            // patches per thread per chunk
    constexpr int NU = UF / 4 / 256;       // float4 of U per thread per chunk
    f32x2 praw[NP][12];
    bool pok[NP];
    f32x4 ureg[NU];
    // The next chunk is fetched as 16 pieces (3 float2 of one patch row + one float4 of U each).
    auto load_piece = [&](int chunk, int k) {       // k = 0..15 (compile-time after unrolling): patch j = k/4, row r = k%4
        const int j = k >> 2, r = k & 3;
        const int c = chunk * WCK + cs + 4 * j;
        const bool okc = c < p.Cin;
        const float* __restrict__ xc = xn + (long long)(okc ? c : 0) * HW;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = r * 3 + q;
            // raw load from a clamped-safe address; the zero-padding / channel mask is applied at transform time so
            // that nothing consumes the value (and waits for it) inside the MFMA phase
            praw[j][e] = *reinterpret_cast<const f32x2*>(xc + (poff[e] >= 0 ? poff[e] : 0));
        }
        pok[j] = okc;
        static_assert(NU == 16, "one U float4 per piece");
        ureg[k] = reinterpret_cast<const f32x4*>(wcg + (long long)chunk * UF)[tid + 256 * k];
    };
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < 16; ++k) load_piece(chunk, k);
    };
    // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    auto transform_store = [&](float* vb, const f32x2 (&pr)[12], bool okc, int c) {
        float d[16], t[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {            // columns c-1, c, c+1, c+2 out of the three aligned pairs (+ padding mask)
            d[r * 4 + 0] = (okc && poff[r * 3 + 0] >= 0) ? pr[r * 3 + 0].y : 0.f;
            d[r * 4 + 1] = (okc && poff[r * 3 + 1] >= 0) ? pr[r * 3 + 1].x : 0.f;
            d[r * 4 + 2] = (okc && poff[r * 3 + 1] >= 0) ? pr[r * 3 + 1].y : 0.f;
            d[r * 4 + 3] = (okc && poff[r * 3 + 2] >= 0) ? pr[r * 3 + 2].x : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {            // rows: t = B^T d
            t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
            t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
            t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
            t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {            // cols: V = t B
            vb[((i * 4 + 0) * WCK + c) * NT + st] = t[i * 4 + 0] - t[i * 4 + 2];
            vb[((i * 4 + 1) * WCK + c) * NT + st] = t[i * 4 + 1] + t[i * 4 + 2];
            vb[((i * 4 + 2) * WCK + c) * NT + st] = t[i * 4 + 2] - t[i * 4 + 1];
            vb[((i * 4 + 3) * WCK + c) * NT + st] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };
    auto store_chunk = [&]() {
        float* ub = lds;
        float* vb = ub + UF;
#pragma unroll
        for (int t = 0; t < NU; ++t) reinterpret_cast<f32x4*>(ub)[tid + 256 * t] = ureg[t];
#pragma unroll
        for (int j = 0; j < NP; ++j) transform_store(vb, praw[j], pok[j], cs + 4 * j);
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

    const int aoff = half * WCO + cbk * 32 + l32;     // + (xi*WCK + 2*kp) * WCO
    const int boff = half * NT + tb * 32 + l32;       // + (xi*WCK + 2*kp) * NT
    const float* ub = lds;
    const float* vb = lds + UF;

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const bool more = chunk + 1 < p.nchunks;     // block-uniform
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
#if KMSR_WINO_ABLATE != 1 && KMSR_WINO_ABLATE != 3
            if (xi == 1 && more) load_chunk(chunk + 1);      // one burst right after the first MFMAs: ~3.5 us to land
                                                             // (spreading the pieces over the xi steps measured slower)
#endif
#pragma unroll
            for (int kp = 0; kp < WCK / 2; ++kp) {
                const float a = ub[aoff + (xi * WCK + 2 * kp) * WCO];
                const float bv = vb[boff + (xi * WCK + 2 * kp) * NT];
#if KMSR_WINO_ABLATE == 2
                asm volatile("" ::"v"(a), "v"(bv));
#else
                acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[xi], 0, 0, 0);
#endif
            }
        }
        __syncthreads();                                     // every wave has finished reading this chunk
        if (more) {
#if KMSR_WINO_ABLATE != 3
            store_chunk();                                   // transform + restage the next chunk
#endif
            __syncthreads();
        }
    }

    // ---- epilogue: lane-local output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), bias, ReLU, residual, store
    float* sbias = lds;
    if (tid < WCO) {
        const int co = cg * WCO + tid;
        sbias[tid] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    __syncthreads();
#if KMSR_WINO_ABLATE == 4
    if (p.relu != 12345) { if (tid == 0 && acc[0][0] == 1.2345f) out[0] = acc[3][2]; return; }
#endif
    const int y0 = ty0 + 2 * tb, x0 = tx0 + 2 * l32;          // top-left output pixel of this lane's tile
    const bool row0 = y0 < H, row1 = y0 + 1 < H, col0 = x0 < W, col1 = x0 + 1 < W;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co0 = cg * WCO + cbk * 32 + 8 * g + 4 * half;              // channels co0 .. co0+3 <-> registers 4g .. 4g+3
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + cbk * 32 + 8 * g + 4 * half);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        float Y[4][4];                                                       // [channel r][2*a + b]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int reg = 4 * g + r;
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = acc[0 * 4 + j][reg], m1 = acc[1 * 4 + j][reg], m2 = acc[2 * 4 + j][reg], m3 = acc[3 * 4 + j][reg];
                s0[j] = m0 + m1 + m2;
                s1[j] = m1 - m2 - m3;
            }
            Y[r][0] = s0[0] + s0[1] + s0[2] + bb[r];
            Y[r][1] = s0[1] - s0[2] - s0[3] + bb[r];
            Y[r][2] = s1[0] + s1[1] + s1[2] + bb[r];
            Y[r][3] = s1[1] - s1[2] - s1[3] + bb[r];
            if (p.relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Y[r][q] = fmaxf(Y[r][q], 0.f);
            }
        }
        if constexpr (PS) {
            // conv channel q = co0 + r -> out[n, q>>2, 2y + (r>>1), 2x + (r&1)]: the lane owns a 4x4 output block
            if (co0 < p.Cout && col0) {
                const long long W2 = 2ll * W;
                float* o = out + (((long long)n * (p.Cout >> 2) + (co0 >> 2)) * (2ll * H) + 2 * y0) * W2 + 2 * x0;
#pragma unroll
                for (int a = 0; a < 2; ++a) {            // output pixel row y0 + a
                    if (a == 0 ? !row0 : !row1) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {        // sub-row i of the shuffle
                        float* orow = o + (2 * a + i) * W2;
                        const f32x2 left = {Y[2 * i + 0][2 * a + 0], Y[2 * i + 1][2 * a + 0]};      // pixel x0
                        const f32x2 right = {Y[2 * i + 0][2 * a + 1], Y[2 * i + 1][2 * a + 1]};     // pixel x0 + 1
                        *reinterpret_cast<f32x2*>(orow) = left;
                        if (col1) *reinterpret_cast<f32x2*>(orow + 2) = right;
                    }
                }
            }
        } else {
            // W is even and x0 is even: (x0, x0+1) is an aligned pair that is inside or outside the image as a whole, so
            // every row of the lane's 2x2 block is ONE 8-byte access and a wave's store is a contiguous 512-byte row piece
            const long long plane = (long long)H * W;
            const long long base = ((long long)n * p.Cout + co0) * plane + (long long)y0 * W + x0;
            f32x2 r0[4], r1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                r0[r] = f32x2{0.f, 0.f};
                r1[r] = f32x2{0.f, 0.f};
            }
            if (res) {   // block-uniform; all loads issued before any is consumed, from clamped-safe addresses
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool okc = co0 + r < p.Cout && col0;
                    const f32x2 a0 = *reinterpret_cast<const f32x2*>(res + ((okc && row0) ? base + r * plane : 0));
                    const f32x2 a1 = *reinterpret_cast<const f32x2*>(res + ((okc && row1) ? base + r * plane + W : 0));
                    r0[r] = (okc && row0) ? a0 : f32x2{0.f, 0.f};
                    r1[r] = (okc && row1) ? a1 : f32x2{0.f, 0.f};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (co0 + r >= p.Cout || !col0) continue;
                float* o = out + base + r * plane;
                if (row0) *reinterpret_cast<f32x2*>(o) = f32x2{Y[r][0] + r0[r].x, Y[r][1] + r0[r].y};
                if (row1) *reinterpret_cast<f32x2*>(o + W) = f32x2{Y[r][2] + r1[r].x, Y[r][3] + r1[r].y};
            }
        }
    }
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" size_t kmsr_conv3x3_wino_packed_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)ceil_div(Cout, WCO) * ceil_div(Cin, WCK) * UF;
}

extern "C" int kmsr_conv3x3_wino_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, float* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "wino pack: bad argument");
    const int nchunks = ceil_div(Cin, WCK);
    const long long total = (long long)ceil_div(Cout, WCO) * nchunks * UF;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    wino_pack_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, packed, nchunks, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv3x3_wino_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                                         const float* bias, const float* residual, float* out, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "winograd conv needs stride 1");
    if (d->W % 2) return fail(KMSR_E_UNSUPPORTED, "winograd conv needs an even width (W=%d): patch rows are fetched as aligned float2", d->W);
    if (reinterpret_cast<uintptr_t>(x) % 8 || reinterpret_cast<uintptr_t>(out) % 8 || reinterpret_cast<uintptr_t>(residual) % 8)
        return fail(KMSR_E_BADARG, "winograd conv needs 8-byte aligned tensors");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if ((long long)d->H * d->W >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "image plane exceeds 2^31 elements");
    WinoParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, 2 * TCOLS);
    p.tiles = p.tiles_x * ceil_div(d->H, 2 * TROWS);
    p.cgs = ceil_div(d->Cout, WCO);
    p.nchunks = ceil_div(d->Cin, WCK);
    p.relu = d->relu;
    const long long nblocks = (long long)p.tiles * p.cgs * d->N;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_blocks = (int)nblocks;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->pixel_shuffle)
        conv3x3_wino_kernel<true><<<p.n_blocks, 256, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else
        conv3x3_wino_kernel<false><<<p.n_blocks, 256, 0, stream>>>(x, packed_w, bias, residual, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
