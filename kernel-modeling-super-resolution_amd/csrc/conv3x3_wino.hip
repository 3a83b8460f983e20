// Winograd F(2x2,3x3) forward convolution on the fp32 matrix cores (semantics: include/kmsr.h).
//
//   V_xi[ci, tile] = (B^T d B)_xi      input transform of the 4x4 patch of every 2x2 output tile   (xi = 0..15)
//   U_xi[co, ci]   = (G g G^T)_xi      weight transform, done once by the pack kernel
//   M_xi[co, tile] = sum_ci U_xi[co,ci] * V_xi[ci,tile]       16 independent GEMMs -> v_mfma_f32_32x32x2_f32
//   Y[co, tile]    = A^T M A           2x2 outputs per tile
//
// Workgroup = 8 waves = 64 output channels x 64 tiles (2 tile rows x 32 tile cols = 4 x 64 output pixels), one workgroup
// per CU, two waves per SIMD.  Wave = (co block, tile row, xi-row half): 8 accumulator tiles (128 AGPRs); for every
// (co, tile) a wave's 8 M values sit in ONE lane at one register index, so its half of A^T M A is lane-local and the two
// halves meet through a 64 KB LDS exchange (details at the kernel).
// Input channels stream in chunks of 16: U chunk 16x16x64 and V chunk 16x16x64 floats (64 KB each) in ONE LDS stage.
// A chunk's MFMA phase (128 MFMAs per wave, ~3.7 us) is long enough for the next chunk's raw 4x4 patches and U rows
// (register staged, issued right after the first MFMAs) to land; they are transformed and written between two
// barriers.  (A double-buffered 8-channel variant was slower: its 1.9 us phases are shorter than a global round trip.)
#include "kmsr_common.h"

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

constexpr int kWinoThreads = 512;

template <bool PS>
__global__ void __launch_bounds__(kWinoThreads, 2) conv3x3_wino_kernel(const float* __restrict__ xg,
                                                                       const float* __restrict__ wpk,
                                                                       const float* __restrict__ bias,
                                                                       const float* __restrict__ res,
                                                                       float* __restrict__ out, WinoParams p) {
    // Workgroup = 8 waves, two per SIMD: wave = (co block cbk, tile row tb, xi-row half ih).  Wave (.,.,ih) multiplies the
    // 8 transform points xi = 4 i + j with i in {2 ih, 2 ih + 1} (8 accumulator tiles = 128 AGPRs), so while one wave of a
    // SIMD is stuck issuing its load burst or restaging, its partner keeps the matrix pipe busy.  Y = A^T (M A) is linear
    // in the xi rows: the partners (wave, wave^4) exchange 2 partial sums per output through LDS and each finalises ONE
    // of the tile's two output rows.
    __shared__ __attribute__((aligned(16))) float lds[WBUF];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cbk = wv & 1, tb = (wv >> 1) & 1, ih = wv >> 2;

    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int tile = bid % p.tiles;
    bid /= p.tiles;
    const int cg = bid % p.cgs;
    const int n = bid / p.cgs;
    const int ty0 = (tile / p.tiles_x) * (2 * TROWS), tx0 = (tile % p.tiles_x) * (2 * TCOLS);
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    // ---- input staging: thread = (tile st = tid & 63, channel sub-lane cs = tid >> 6 in 0..7); channels cs, cs + 8 of the chunk
    const int st = tid & 63, cs = tid >> 6, str = st >> 5, stc = st & 31;
    // The 4x4 patch of tile (str, stc) covers rows ty0+2str-1 .. +2 and columns c-1 .. c+2 with c = tx0 + 2 stc (even).
    // Each row is fetched as three ALIGNED float2 (c-2,c-1) (c,c+1) (c+2,c+3): neighbouring lanes are exactly 8 bytes
    // apart, so every load instruction is one contiguous 512-byte request.  W is even on this path, so an aligned pair
    // is entirely inside or entirely outside the image.
    const int py0 = ty0 + 2 * str - 1, pc0 = tx0 + 2 * stc - 2;
    int poff[12];       // [row r][pair k]: offset of the pair inside one channel plane, -1 = outside (zero padding)
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int y = py0 + e / 3, x = pc0 + 2 * (e % 3);
        poff[e] = (y >= 0 && y < H && x >= 0 && x + 1 < W) ? y * W + x : -1;
    }
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const float* __restrict__ wcg = wpk + (long long)cg * p.nchunks * UF;

    constexpr int NP = WCK / 8;                       // patches per thread per chunk (2)
    constexpr int NU = UF / 4 / kWinoThreads;         // float4 of U per thread per chunk (8)
    f32x2 praw[NP][12];
    bool pok[NP];
    f32x4 ureg[NU];
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = chunk * WCK + cs + 8 * j;
            const bool okc = c < p.Cin;
            const float* __restrict__ xc = xn + (long long)(okc ? c : 0) * HW;
#pragma unroll
            for (int e = 0; e < 12; ++e)   // raw loads from clamped-safe addresses; the padding mask is applied at transform time
                praw[j][e] = *reinterpret_cast<const f32x2*>(xc + (poff[e] >= 0 ? poff[e] : 0));
            pok[j] = okc;
        }
        const f32x4* __restrict__ wc = reinterpret_cast<const f32x4*>(wcg + (long long)chunk * UF);
#pragma unroll
        for (int t = 0; t < NU; ++t) ureg[t] = wc[tid + kWinoThreads * t];
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
        for (int t = 0; t < NU; ++t) reinterpret_cast<f32x4*>(ub)[tid + kWinoThreads * t] = ureg[t];
#pragma unroll
        for (int j = 0; j < NP; ++j) transform_store(vb, praw[j], pok[j], cs + 8 * j);
    };

    f32x16 acc[8];      // local index il*4 + j  <->  xi = (2 ih + il)*4 + j
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int aoff = half * WCO + cbk * 32 + l32 + (8 * ih) * WCK * WCO;     // + (a*WCK + 2*kp) * WCO
    const int boff = half * NT + tb * 32 + l32 + (8 * ih) * WCK * NT;        // + (a*WCK + 2*kp) * NT
    const float* ub = lds;
    const float* vb = lds + UF;

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const bool more = chunk + 1 < p.nchunks;     // block-uniform
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (a == 1 && more) load_chunk(chunk + 1);       // one burst right after the first MFMAs: ~3.5 us to land
#pragma unroll
            for (int kp = 0; kp < WCK / 2; ++kp) {
                const float av = ub[aoff + (a * WCK + 2 * kp) * WCO];
                const float bv = vb[boff + (a * WCK + 2 * kp) * NT];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
        __syncthreads();                                     // every wave has finished reading this chunk
        if (more) {
            store_chunk();                                   // transform + restage the next chunk
            __syncthreads();
        }
    }

    // ---- epilogue.  Own half of the column transform: MA[il][b], il = 0,1 (xi rows 2ih, 2ih+1), b = 0,1 (A^T = [[1,1,1,0],[0,1,-1,-1]]).
    //      Row transform: Y[0][b] = MA_0 + MA_1 + MA_2,  Y[1][b] = MA_1 - MA_2 - MA_3  -> per wave a "keep" part for the
    //      output row it finalises (row ih) and a "send" part for the partner's row.
    f32x2* xch = reinterpret_cast<f32x2*>(lds);              // [8 waves][16 regs][64 lanes] float2 = 64 KB (U/V are dead)
    float* sbias = lds + 8 * 16 * 64 * 2;
    if (tid < WCO) {
        const int co = cg * WCO + tid;
        sbias[tid] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    f32x2 keep[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        float ma[2][2];
#pragma unroll
        for (int il = 0; il < 2; ++il) {
            const float m0 = acc[il * 4 + 0][reg], m1 = acc[il * 4 + 1][reg], m2 = acc[il * 4 + 2][reg], m3 = acc[il * 4 + 3][reg];
            ma[il][0] = m0 + m1 + m2;
            ma[il][1] = m1 - m2 - m3;
        }
        f32x2 send;
        if (ih == 0) {      // xi rows 0,1: Y0 += MA_0 + MA_1 (keep), Y1 += MA_1 (send)
            keep[reg] = f32x2{ma[0][0] + ma[1][0], ma[0][1] + ma[1][1]};
            send = f32x2{ma[1][0], ma[1][1]};
        } else {            // xi rows 2,3: Y0 += MA_2 (send), Y1 += -MA_2 - MA_3 (keep)
            keep[reg] = f32x2{-ma[0][0] - ma[1][0], -ma[0][1] - ma[1][1]};
            send = f32x2{ma[0][0], ma[0][1]};
        }
        xch[(wv * 16 + reg) * 64 + lane] = send;
    }
    __syncthreads();
    const int y = ty0 + 2 * tb + ih, x0 = tx0 + 2 * l32;     // this wave finalises output row `ih` of its tiles
    const bool pok2 = y < H && x0 < W;                        // (x0, x0+1) is an aligned pair: inside or outside as a whole
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co0 = cg * WCO + cbk * 32 + 8 * g + 4 * half;              // channels co0 .. co0+3 <-> registers 4g .. 4g+3
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + cbk * 32 + 8 * g + 4 * half);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        f32x2 Y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 o = xch[((wv ^ 4) * 16 + 4 * g + r) * 64 + lane];
            Y[r] = f32x2{keep[4 * g + r].x + o.x + bb[r], keep[4 * g + r].y + o.y + bb[r]};
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
            const long long plane = (long long)H * W;
            const long long base = ((long long)n * p.Cout + co0) * plane + (long long)y * W + x0;
            f32x2 rr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rr[r] = f32x2{0.f, 0.f};
            if (res) {   // block-uniform; loads issued together from clamped-safe addresses
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = pok2 && co0 + r < p.Cout;
                    const f32x2 v = *reinterpret_cast<const f32x2*>(res + (ok ? base + r * plane : 0));
                    rr[r] = ok ? v : f32x2{0.f, 0.f};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (pok2 && co0 + r < p.Cout)
                    *reinterpret_cast<f32x2*>(out + base + r * plane) = f32x2{Y[r].x + rr[r].x, Y[r].y + rr[r].y};
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
    if (d->flags) return fail(KMSR_E_UNSUPPORTED, "conv flags 0x%x are honoured by kmsr_conv3x3_winov_fwd_ex_f32 only", d->flags);
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
        conv3x3_wino_kernel<true><<<p.n_blocks, kWinoThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else
        conv3x3_wino_kernel<false><<<p.n_blocks, kWinoThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
