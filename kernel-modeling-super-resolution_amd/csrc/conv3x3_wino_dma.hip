// Winograd F(2x2,3x3) forward convolution, LDS-DMA variant (semantics: include/kmsr.h, same math and epilogue as
// conv3x3_wino.hip).  The ablation of the register-staged kernel (profiles/r01_wino_ablation_4wave.log) shows its
// 32 VMEM instructions per thread and 16-channel chunk are not hidden by the matrix pipe, and their 80 staging VGPRs
// leave no room to prefetch anything else.  Here nothing global is staged through registers:
//   * the chunk's raw input rows (8 channels x 6 rows x 72 columns, 16-byte aligned, halo included) and its
//     pre-transformed weights U (8 x 16 x 64) go global -> LDS with `global_load ... lds` (2 + 4 instructions per thread
//     and chunk, lane-linear destinations), double buffered, issued a whole transform + MFMA phase ahead;
//   * the 4x4 patches are read from the raw LDS tile (3 ds_read_b64 per patch row), transformed (B^T d B) and written
//     to the single V buffer between two barriers; out-of-image rows / columns / channels are masked at transform time,
//     so lanes whose 16 bytes lie outside the image simply do not issue their load.
// LDS: U 2 x 32 KB + raw 2 x 16 KB + V 32 KB = 128 KB; the epilogue's 64 KB exchange buffer aliases the U buffers.
// Requires W % 4 == 0 and 16-byte aligned tensors (the register-staged kernel serves everything else).
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int DCK = 8;                 // input channels per chunk
constexpr int DCO = 64;                // output channels per workgroup
constexpr int TROWS = 2, TCOLS = 32;   // tiles per workgroup: 2 x 32  (output 4 rows x 64 cols)
constexpr int NT = TROWS * TCOLS;      // 64 tiles
constexpr int DUF = 16 * DCK * DCO;    // 8192 floats = 32 KB
constexpr int DVF = 16 * DCK * NT;     // 8192 floats = 32 KB
constexpr int RROWS = 2 * TROWS + 2;   // 6 input rows
constexpr int RQ = (2 * TCOLS + 8) / 4;            // 18 float4 per row segment: columns tx0-4 .. tx0+67
constexpr int RCOLS = 4 * RQ;                      // 72
constexpr int RLANES = DCK * RROWS * RQ;           // 864 lane-loads per chunk
constexpr int kThreads = 512;
constexpr int RINSTR = 2;                          // raw DMA instructions per wave and chunk (8 waves x 2 x 64 = 1024 >= 864)
constexpr int RAWF = 8 * RINSTR * 64 * 4;          // 4096 floats = 16 KB
constexpr int UINSTR = DUF / 4 / kThreads;         // 4 U DMA instructions per wave and chunk
constexpr int NRAW = 3;                            // raw buffers: input rows are fetched 3 chunks ahead (HBM latency), U 2 ahead (L2)
constexpr int OFF_U = 0, OFF_V = 2 * DUF, OFF_R = 2 * DUF + DVF, OFF_B = OFF_R + NRAW * RAWF;
constexpr int LDS_FLOATS = OFF_B + DCO;

struct WinoParams {
    int N, Cin, H, W, Cout;            // stride 1: Ho = H, Wo = W
    int tiles_x, tiles, cgs, nchunks, relu, n_blocks;
};

// packed[cg][chunk][xi][c][co] = (G g G^T)[xi] for (co, ci) = (cg*64+co, chunk*8+c), zero padded
__global__ void winod_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ packed, int nchunks,
                                  long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % DCO);
        const int c = (int)((idx / DCO) % DCK);
        const int xi = (int)((idx / (DCO * DCK)) % 16);
        const int chunk = (int)((idx / (DCO * DCK * 16)) % nchunks);
        const int cg = (int)(idx / ((long long)DCO * DCK * 16 * nchunks));
        const int cog = cg * DCO + co, cig = chunk * DCK + c;
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

__device__ __forceinline__ void sync_lds() {     // LDS hand-over only: does NOT wait for global loads / LDS-DMA in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void sync_all_keep() {   // barrier after an explicit (counted) vmcnt wait by the caller
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void sync_all() {     // + every wave's outstanding LDS-DMA has landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool PS>
__global__ void __launch_bounds__(kThreads, 2) conv3x3_winod_kernel(const float* __restrict__ xg,
                                                                    const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ res,
                                                                    float* __restrict__ out, WinoParams p) {
    // wave = (co block cbk, tile row tb, xi-row half ih) exactly as in conv3x3_wino.hip
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
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
    const float* __restrict__ xn = xg + (long long)n * p.Cin * HW;
    const float* __restrict__ wcg = wpk + (long long)cg * p.nchunks * DUF;

    // ---- LDS-DMA maps (the same for every chunk).  Raw tile, lane-linear: L = (2 wv + i) * 64 + lane  <->
    //      (segment = L / 18 = channel * 6 + row, q = L % 18 = float4 within the row); LDS offset = 4 L floats.
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
    // Every wave issues EXACTLY RINSTR raw and UINSTR weight DMA instructions per chunk (lanes outside the image read a
    // clamped in-bounds address; their LDS words are masked at transform time), so `s_waitcnt vmcnt(RINSTR)` means
    // "everything except the newest raw chunk has landed".
    auto issue_raw = [&](int chunk) {
        const bool cok = chunk < p.nchunks;
        const float* __restrict__ xc = xn + (long long)(cok ? chunk : 0) * DCK * HW;
        const int cin_left = p.Cin - (cok ? chunk : 0) * DCK;
        float* rb = lds + OFF_R + (chunk % NRAW) * RAWF;
#pragma unroll
        for (int i = 0; i < RINSTR; ++i) {
            const bool ok = roff[i] >= 0 && rch[i] < cin_left;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(xc + (ok ? roff[i] : 0)), (lds_void_t*)(rb + (RINSTR * wv + i) * 256), 16, 0, 0);
        }
    };
    auto issue_u = [&](int chunk) {
        const float* __restrict__ wc = wcg + (long long)chunk * DUF;
        float* ub = lds + OFF_U + (chunk & 1) * DUF;
#pragma unroll
        for (int i = 0; i < UINSTR; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wc + ((UINSTR * wv + i) * 64 + lane) * 4),
                                             (lds_void_t*)(ub + (UINSTR * wv + i) * 256), 16, 0, 0);
    };

    // ---- input transform: thread = (tile st = tid & 63, channel cs = tid >> 6 of the chunk)
    const int st = tid & 63, cs = tid >> 6, str = st >> 5, stc = st & 31;
    // patch rows r = 0..3 <-> raw rows 2 str + r; columns c-1 .. c+2 with c = tx0 + 2 stc <-> raw columns 2 stc + 3 .. + 6,
    // read as the three aligned pairs at raw columns 2 stc + 2, + 4, + 6
    unsigned pmask = 0;                 // bit r*3+k: pair k of row r is inside the image
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int y = ty0 + 2 * str - 1 + e / 3, x = tx0 + 2 * stc - 2 + 2 * (e % 3);
        if (y >= 0 && y < H && x >= 0 && x + 1 < W) pmask |= 1u << e;
    }
    const int rbase = (cs * RROWS + 2 * str) * RCOLS + 2 * stc + 2;
    // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    auto transform = [&](int chunk) {
        const float* rb = lds + OFF_R + (chunk % NRAW) * RAWF + rbase;
        float* vb = lds + OFF_V;
        const bool okc = chunk * DCK + cs < p.Cin;
        f32x2 pr[12];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) pr[r * 3 + k] = *reinterpret_cast<const f32x2*>(rb + r * RCOLS + 2 * k);
        float d[16], t[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool m0 = okc && ((pmask >> (r * 3 + 0)) & 1u), m1 = okc && ((pmask >> (r * 3 + 1)) & 1u),
                       m2 = okc && ((pmask >> (r * 3 + 2)) & 1u);
            d[r * 4 + 0] = m0 ? pr[r * 3 + 0].y : 0.f;
            d[r * 4 + 1] = m1 ? pr[r * 3 + 1].x : 0.f;
            d[r * 4 + 2] = m1 ? pr[r * 3 + 1].y : 0.f;
            d[r * 4 + 3] = m2 ? pr[r * 3 + 2].x : 0.f;
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
            vb[((i * 4 + 0) * DCK + cs) * NT + st] = t[i * 4 + 0] - t[i * 4 + 2];
            vb[((i * 4 + 1) * DCK + cs) * NT + st] = t[i * 4 + 1] + t[i * 4 + 2];
            vb[((i * 4 + 2) * DCK + cs) * NT + st] = t[i * 4 + 2] - t[i * 4 + 1];
            vb[((i * 4 + 3) * DCK + cs) * NT + st] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };

    f32x16 acc[8];      // local index il*4 + j  <->  xi = (2 ih + il)*4 + j
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int aoff = half * DCO + cbk * 32 + l32 + (8 * ih) * DCK * DCO;     // + (a*DCK + 2*kp) * DCO
    const int boff = half * NT + tb * 32 + l32 + (8 * ih) * DCK * NT;        // + (a*DCK + 2*kp) * NT
    const float* vb = lds + OFF_V;

    const int nch = p.nchunks;
    issue_raw(0);
    issue_u(0);
    if (nch > 1) { issue_raw(1); issue_u(1); }
    if (nch > 2) issue_raw(2);
    if (nch > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RINSTR + UINSTR) : "memory");   // chunk 0 (the oldest 6 instructions) has landed
    else if (nch > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RINSTR + UINSTR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sync_all_keep();
    transform(0);
    sync_lds();
    // residual (skip) values of the outputs this wave finalises, [4g + r] <-> channel co0(g) + r, pixel pair (ry, rx0..rx0+1):
    // fetched at the start of the LAST chunk's MFMA phase (no staging registers compete for the space in this kernel)
    f32x2 rres[16];
    const int ry = ty0 + 2 * tb + ih, rx0 = tx0 + 2 * l32;
    auto prefetch_residual = [&]() {
        if constexpr (!PS) {
            if (res) {                                       // block-uniform
                const long long plane = (long long)H * W;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co0 = cg * DCO + cbk * 32 + 8 * g + 4 * half;
                    const long long base = ((long long)n * p.Cout + co0) * plane + (long long)ry * W + rx0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = ry < H && rx0 < W && co0 + r < p.Cout;
                        rres[4 * g + r] = *reinterpret_cast<const f32x2*>(res + (ok ? base + r * plane : 0));   // masked at use
                    }
                }
            }
        }
    };
    for (int chunk = 0; chunk < nch; ++chunk) {
        const float* ub = lds + OFF_U + (chunk & 1) * DUF;
        if (chunk + 1 == nch) prefetch_residual();
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int kp = 0; kp < DCK / 2; ++kp) {
                const float av = ub[aoff + (a * DCK + 2 * kp) * DCO];
                const float bv = vb[boff + (a * DCK + 2 * kp) * NT];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
        // every wave has finished this chunk; U(chunk+1) and raw(chunk+1) have landed, raw(chunk+2) may still be in flight
        if (chunk + 2 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RINSTR) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sync_all_keep();
        if (chunk + 1 < nch) {
            if (chunk + 2 < nch) issue_u(chunk + 2);         // U buffer (chunk & 1) is free again
            if (chunk + 3 < nch) issue_raw(chunk + 3);       // raw buffer (chunk % 3) was consumed by this chunk's transform
            transform(chunk + 1);
            sync_lds();
        }
    }

    // ---- epilogue.  Own half of the column transform: MA[il][b], il = 0,1 (xi rows 2ih, 2ih+1), b = 0,1 (A^T = [[1,1,1,0],[0,1,-1,-1]]).
    //      Row transform: Y[0][b] = MA_0 + MA_1 + MA_2,  Y[1][b] = MA_1 - MA_2 - MA_3  -> per wave a "keep" part for the
    //      output row it finalises (row ih) and a "send" part for the partner's row.
    f32x2* xch = reinterpret_cast<f32x2*>(lds);              // [8 waves][16 regs][64 lanes] float2 = 64 KB (U/V are dead)
    float* sbias = lds + OFF_B;
    if (tid < DCO) {
        const int co = cg * DCO + tid;
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
        const int co0 = cg * DCO + cbk * 32 + 8 * g + 4 * half;              // channels co0 .. co0+3 <-> registers 4g .. 4g+3
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
            if (res) {   // block-uniform; fetched during the last chunk's MFMA phase (prefetch_residual)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = pok2 && co0 + r < p.Cout;
                    rr[r] = ok ? rres[4 * g + r] : f32x2{0.f, 0.f};
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

extern "C" size_t kmsr_conv3x3_winod_packed_weight_floats(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)ceil_div(Cout, DCO) * ceil_div(Cin, DCK) * DUF;
}

extern "C" int kmsr_conv3x3_winod_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, float* packed, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return fail(KMSR_E_BADARG, "winograd-dma pack: bad argument");
    const int nchunks = ceil_div(Cin, DCK);
    const long long total = (long long)ceil_div(Cout, DCO) * nchunks * DUF;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    winod_pack_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(w, Cin, Cout, packed, nchunks, total);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_conv3x3_winod_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                                          const float* bias, const float* residual, float* out, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (d->stride != 1) return fail(KMSR_E_UNSUPPORTED, "winograd conv needs stride 1");
    if (d->W % 4) return fail(KMSR_E_UNSUPPORTED, "winograd-dma conv needs W %% 4 == 0 (W=%d): rows are fetched as aligned 16-byte pieces", d->W);
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(packed_w) % 16 || reinterpret_cast<uintptr_t>(out) % 8 ||
        reinterpret_cast<uintptr_t>(residual) % 8)
        return fail(KMSR_E_BADARG, "winograd-dma conv needs 16-byte aligned input / weights and 8-byte aligned output");
    if (d->pixel_shuffle && (d->Cout % 4)) return fail(KMSR_E_SHAPE, "pixel_shuffle needs Cout %% 4 == 0 (Cout=%d)", d->Cout);
    if (d->pixel_shuffle && residual) return fail(KMSR_E_BADARG, "residual is not supported together with pixel_shuffle");
    if (!x || !packed_w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if ((long long)d->H * d->W * DCK >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "image plane exceeds 2^28 elements");
    WinoParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.tiles_x = ceil_div(d->W, 2 * TCOLS);
    p.tiles = p.tiles_x * ceil_div(d->H, 2 * TROWS);
    p.cgs = ceil_div(d->Cout, DCO);
    p.nchunks = ceil_div(d->Cin, DCK);
    p.relu = d->relu;
    const long long nblocks = (long long)p.tiles * p.cgs * d->N;
    if (nblocks > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_blocks = (int)nblocks;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->pixel_shuffle)
        conv3x3_winod_kernel<true><<<p.n_blocks, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    else
        conv3x3_winod_kernel<false><<<p.n_blocks, kThreads, 0, stream>>>(x, packed_w, bias, residual, out, p);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
