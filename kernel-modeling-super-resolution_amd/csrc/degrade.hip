// Fused degradation for gfx950: depthwise blur + xF downsample + noise in ONE pass over the HR batch.
//
// Reference semantics restated in include/kmsr.h (C_30apply_kernel_to_landsat.py:68-124,
// muti_kernel/train_gemini.py:117-138, E_make_train_data.py:65-74).
//
// Both downsample modes are executed as one strided correlation with an "effective tap table":
//     out[n,c,i,j] = sum_{a<TY,b<TX} taps[row(n),c,a,b] * hr[n,c,pad(F*i+a-ph),pad(F*j+b-pw)]
//   STRIDE : taps = K (TY=kh, TX=kw)
//   AVGPOOL: taps = K (*) box_F / F^2 (TY=kh+F-1, TX=kw+F-1): blur followed by the 2x2 avg-pool
//            cascade is linear, so it collapses to one table (SURVEY.md 8a' "fused form").
// The table is built on device by prepare_taps_kernel (normalisation included), then consumed by
//   - degrade_rowstream_kernel: the fast path.  One WAVE owns a (plane, column strip, row chunk)
//     unit and streams HR rows top to bottom: each row segment is loaded once with coalesced
//     16-byte loads, staged through a wave-private LDS row buffer so neighbouring lanes can read
//     each other's halo, and folded into TYB = ceil(TY/F) open accumulator rows held in registers.
//     HR bytes are read ~once (plus a (TY-F)/(F*rows_per_chunk) vertical halo that hits L2), taps
//     arrive through the scalar cache (wave-uniform), no block-level barriers exist.
//   - degrade_generic_kernel: one thread per output, any F / kernel size / alignment.
#include "kmsr_common.h"

namespace kmsr {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;

struct DegradeParams {
    const float* hr;
    long long sN, sC, sH;
    int N, C, H, W;
    int h, w;
    int F, TY, TX, TXP, ph, pw;
    const float* taps;
    int n_kernels;
    const int* kidx;
    const float* sigma;
    const float* noise;
    const long long* nidx;
    float* lr;
    unsigned long long seed;
    unsigned offset;
    int pad_zero;
    int noise_mode;
    int rows_per_chunk, n_chunks, n_strips, n_units, n_blocks;
};

// ------------------------------------------------------------------------------------------------
// Counter-based Gaussian stream (see oracle/kmsr_oracle.c: oracle_eps -- the two must stay in sync)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four normals for rows 4*iblk .. 4*iblk+3 of column j in `plane`
__device__ __forceinline__ void eps4(unsigned long long seed, unsigned offset, unsigned plane, unsigned iblk,
                                     unsigned j, float z[4]) {
    unsigned r[4];
    philox4x32_10(j, iblk, plane, offset, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const float u1 = ((float)(r[2 * pr] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float u2 = (float)(r[2 * pr + 1] >> 8) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1));
        float s, c;
        sincospif(2.0f * u2, &s, &c);
        z[2 * pr] = rad * c;
        z[2 * pr + 1] = rad * s;
    }
}

__device__ __forceinline__ float pick4(const float z[4], int k) {
    return k == 0 ? z[0] : (k == 1 ? z[1] : (k == 2 ? z[2] : z[3]));
}

// ------------------------------------------------------------------------------------------------
// Effective tap table.  One block per (bank row, band).
// ------------------------------------------------------------------------------------------------
__global__ void prepare_taps_kernel(const float* __restrict__ kernels, int kh, int kw, int F, int avgpool,
                                    int normalize, int TY, int TX, int TXP, float* __restrict__ taps) {
    __shared__ double kn[31 * 31];
    __shared__ double ksum;
    const float* k = kernels + (size_t)blockIdx.x * kh * kw;
    if (threadIdx.x == 0) {
        float s = 0.f;  // fp32 running sum like `kernel[i].sum()` (C_30:95); order differences are < 1 ulp of 1
        for (int i = 0; i < kh * kw; ++i) s += k[i];
        ksum = (double)s;
    }
    __syncthreads();
    const bool div = normalize && ksum > 0.0;
    for (int i = threadIdx.x; i < kh * kw; i += blockDim.x)
        kn[i] = div ? (double)(k[i] / (float)ksum) : (double)k[i];  // fp32 division as in C_30:97
    __syncthreads();
    float* t = taps + (size_t)blockIdx.x * TY * TXP;
    for (int e = threadIdx.x; e < TY * TXP; e += blockDim.x) {
        const int a = e / TXP, b = e % TXP;
        double v = 0.0;
        if (b < TX) {
            if (avgpool) {
                for (int dy = 0; dy < F; ++dy)
                    for (int dx = 0; dx < F; ++dx) {
                        const int u = a - dy, vv = b - dx;
                        if (u >= 0 && u < kh && vv >= 0 && vv < kw) v += kn[u * kw + vv];
                    }
                v /= (double)(F * F);
            } else {
                v = kn[a * kw + b];
            }
        }
        t[e] = (float)v;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: one thread per output element.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) degrade_generic_kernel(DegradeParams p) {
    const long long total = (long long)p.N * p.C * p.h * p.w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % p.w);
        const int i = (int)((idx / p.w) % p.h);
        const int plane = (int)(idx / ((long long)p.w * p.h));
        const int n = plane / p.C, c = plane % p.C;
        const int krow = p.kidx ? p.kidx[n] : (p.n_kernels == 1 ? 0 : n);
        const float* tp = p.taps + (size_t)(krow * p.C + c) * p.TY * p.TXP;
        const float* img = p.hr + n * p.sN + c * p.sC;
        float acc = 0.f;
        for (int a = 0; a < p.TY; ++a) {
            int y = p.F * i + a - p.ph;
            if (p.pad_zero) {
                if (y < 0 || y >= p.H) continue;
            } else {
                y = min(max(y, 0), p.H - 1);
            }
            const float* row = img + y * p.sH;
            for (int b = 0; b < p.TX; ++b) {
                int x = p.F * j + b - p.pw;
                if (p.pad_zero) {
                    if (x < 0 || x >= p.W) continue;
                } else {
                    x = min(max(x, 0), p.W - 1);
                }
                acc = fmaf(tp[a * p.TXP + b], row[x], acc);
            }
        }
        if (p.noise_mode == KMSR_NOISE_GAUSSIAN) {
            float z[4];
            eps4(p.seed, p.offset, (unsigned)plane, (unsigned)(i >> 2), (unsigned)j, z);
            acc += p.sigma[krow * p.C + c] * pick4(z, i & 3);
        } else if (p.noise_mode == KMSR_NOISE_POOL) {
            const long long pi = p.nidx ? p.nidx[n] : n;
            acc += p.noise[((pi * p.C + c) * p.h + i) * p.w + j];
        }
        p.lr[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Row-streaming fast kernel.
// ------------------------------------------------------------------------------------------------
template <int F, int TY, int TX, int PH, int PW, int NX>
struct RowCfg {
    static constexpr int TYB = (TY - 1) / F + 1;                  // open accumulator rows
    static constexpr int TXP = (TX + 3) / 4 * 4;
    static constexpr int G = (F % 4 == 0) ? 4 : ((F % 2 == 0) ? 2 : 1);  // LDS read vector width (floats)
    static constexpr int HL = (PW + 3) / 4 * 4;                   // left halo kept in the LDS row
    static constexpr int OFF = (HL - PW) % G;                     // first useful float inside the window
    static constexpr int NWG = (OFF + TX + G - 1) / G;            // vector reads per window
    static constexpr int SW = kWave * NX;                         // LR columns per strip
    static constexpr int OWN = F * SW;
    static constexpr int RH_ = NWG * G - F - PW - OFF;
    static constexpr int HRT = RH_ > 0 ? (RH_ + 3) / 4 * 4 : 0;
    static constexpr int ROWF = HL + OWN + HRT;                   // floats per staged row segment
    static constexpr int NV = ROWF / 4;                           // float4 per staged row segment
    static constexpr int NLD = (NV + kWave - 1) / kWave;          // float4 global loads per lane per row
    static_assert(ROWF % 4 == 0, "row buffer must be float4 granular");
    static_assert(OWN % 4 == 0, "strip must start 16-byte aligned");
};

// native clang vectors: one IR load/store of known alignment -> ds_read_b128 / global_load_dwordx4
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int G>
struct VecT;
template <> struct VecT<4> { using type = f32x4; };
template <> struct VecT<2> { using type = f32x2; };
template <> struct VecT<1> { using type = float; };

template <int F, int TY, int TX, int PH, int PW, int NX>
__global__ void __launch_bounds__(kWave* kWavesPerBlock)
    degrade_rowstream_kernel(const float* __restrict__ hr, const float* __restrict__ taps_all,
                             const float* __restrict__ sigma, const float* __restrict__ noise,
                             float* __restrict__ lr, DegradeParams p) {
    using Cfg = RowCfg<F, TY, TX, PH, PW, NX>;
    constexpr int TYB = Cfg::TYB, G = Cfg::G, NWG = Cfg::NWG, OFF = Cfg::OFF, HL = Cfg::HL;
    constexpr int ROWF = Cfg::ROWF, NV = Cfg::NV, NLD = Cfg::NLD, SW = Cfg::SW, TXP = Cfg::TXP;
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][ROWF];

    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware block remap (bijective form): blocks b, b+8, ... share an XCD, give them adjacent
    // units so vertically adjacent row chunks (which share TY-F halo rows) meet in one L2.
    const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    const int blk = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int unit = blk * kWavesPerBlock + wid;
    if (unit >= p.n_units) return;  // wave-uniform; no block barriers below

    const int chunk = unit % p.n_chunks;
    const int t1 = unit / p.n_chunks;
    const int strip = t1 % p.n_strips;
    const int plane = t1 / p.n_strips;
    const int n = plane / p.C, c = plane - n * p.C;
    const int krow = p.kidx ? __builtin_amdgcn_readfirstlane(p.kidx[n]) : (p.n_kernels == 1 ? 0 : n);
    const float* __restrict__ taps = taps_all + (size_t)(krow * p.C + c) * (TY * TXP);
    const float* __restrict__ img = hr + n * p.sN + c * p.sC;
    const int H = p.H, W = p.W;
    const int i0 = chunk * p.rows_per_chunk;
    const int i1 = min(i0 + p.rows_per_chunk, p.h);
    const int xs0 = strip * SW;
    const int col0 = F * xs0 - HL;  // HR column of LDS float 0 (multiple of 4)
    float* __restrict__ row = lds[wid];
    const bool pad_zero = p.pad_zero != 0;

    float acc[TYB][NX];
#pragma unroll
    for (int m = 0; m < TYB; ++m)
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[m][x] = 0.f;

    float sig = 0.f;
    if (p.noise_mode == KMSR_NOISE_GAUSSIAN) sig = sigma[krow * p.C + c];
    const float* __restrict__ npl = nullptr;
    if (p.noise_mode == KMSR_NOISE_POOL) {
        const long long pi = p.nidx ? p.nidx[n] : (long long)n;
        npl = noise + (pi * p.C + c) * (long long)p.h * p.w;
    }
    float z[NX][4];
#pragma unroll
    for (int x = 0; x < NX; ++x) z[x][0] = z[x][1] = z[x][2] = z[x][3] = 0.f;

    // shifted row index s = r + PH; output row i consumes s in [F*i, F*i+TY)
    const int s_begin = F * i0;
    const int s_last = F * (i1 - 1) + TY - 1;                  // last row that carries a tap
    const int s_end = F * (i1 - 1 + TYB - 1) + F - 1;          // end of the group that retires row i1-1

    // is HR row for shifted index s needed at all / which image row feeds it
    auto row_needed = [&](int s) -> bool {
        if (s > s_last) return false;
        const int r = s - PH;
        if (pad_zero && (r < 0 || r >= H)) return false;
        return true;
    };
    f32x4 pre[NLD];
    auto issue_loads = [&](int s) {
        const int r = min(max(s - PH, 0), H - 1);
        const float* __restrict__ rp = img + (long long)r * p.sH;
#pragma unroll
        for (int t = 0; t < NLD; ++t) {
            const int q = lane + kWave * t;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < NV) {
                const int c0 = col0 + 4 * q;
                if (c0 >= 0 && c0 + 3 < W) {
                    v = *reinterpret_cast<const f32x4*>(rp + c0);
                } else if (pad_zero) {
                    if (c0 + 0 >= 0 && c0 + 0 < W) v.x = rp[c0 + 0];
                    if (c0 + 1 >= 0 && c0 + 1 < W) v.y = rp[c0 + 1];
                    if (c0 + 2 >= 0 && c0 + 2 < W) v.z = rp[c0 + 2];
                    if (c0 + 3 >= 0 && c0 + 3 < W) v.w = rp[c0 + 3];
                } else {
                    v.x = rp[min(max(c0 + 0, 0), W - 1)];
                    v.y = rp[min(max(c0 + 1, 0), W - 1)];
                    v.z = rp[min(max(c0 + 2, 0), W - 1)];
                    v.w = rp[min(max(c0 + 3, 0), W - 1)];
                }
            }
            pre[t] = v;
        }
    };

    if (row_needed(s_begin)) issue_loads(s_begin);
    int g = i0, q = 0;
    for (int s = s_begin; s <= s_end; ++s) {
        if (row_needed(s)) {
            // ---- stage this row: registers -> wave-private LDS row (contiguous 16-B stores)
#pragma unroll
            for (int t = 0; t < NLD; ++t) {
                const int qq = lane + kWave * t;
                if (qq < NV) *reinterpret_cast<f32x4*>(row + 4 * qq) = pre[t];
            }
        }
        // ---- prefetch the next row while this one is consumed (also when this one was skipped)
        if (row_needed(s + 1)) issue_loads(s + 1);
        if (row_needed(s)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- each lane reads the windows of its NX outputs (neighbour lanes overlap: that is the halo share)
            float win[NX][NWG * G];
#pragma unroll
            for (int x = 0; x < NX; ++x) {
                const int base = F * (lane + kWave * x) + (HL - PW - OFF);
#pragma unroll
                for (int t = 0; t < NWG; ++t) {
                    using V = typename VecT<G>::type;
                    const V v = *reinterpret_cast<const V*>(row + base + G * t);
                    if constexpr (G == 4) {
                        win[x][4 * t + 0] = v.x; win[x][4 * t + 1] = v.y; win[x][4 * t + 2] = v.z; win[x][4 * t + 3] = v.w;
                    } else if constexpr (G == 2) {
                        win[x][2 * t + 0] = v.x; win[x][2 * t + 1] = v.y;
                    } else {
                        win[x][t] = v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- fold the row into every open output row (tap row a = F*m + q of slot m)
#pragma unroll
            for (int m = 0; m < TYB; ++m) {
                const int a = F * m + q;
                const int i = g - m;
                if (a < TY && i >= i0 && i < i1) {  // wave-uniform
                    const float* __restrict__ tp = taps + a * TXP;
#pragma unroll
                    for (int b = 0; b < TX; ++b) {
                        const float kv = tp[b];
#pragma unroll
                        for (int x = 0; x < NX; ++x) acc[m][x] = fmaf(kv, win[x][OFF + b], acc[m][x]);
                    }
                }
            }
        }
        if (++q == F) {
            // ---- group finished: slot TYB-1 holds output row i = g-(TYB-1), complete
            const int i = g - (TYB - 1);
            if (i >= i0 && i < i1) {
                if (p.noise_mode == KMSR_NOISE_GAUSSIAN && (i & 3) == 0) {
#pragma unroll
                    for (int x = 0; x < NX; ++x)
                        eps4(p.seed, p.offset, (unsigned)plane, (unsigned)(i >> 2), (unsigned)(xs0 + lane + kWave * x),
                             z[x]);
                }
#pragma unroll
                for (int x = 0; x < NX; ++x) {
                    const int xo = xs0 + lane + kWave * x;
                    if (xo < p.w) {
                        float v = acc[TYB - 1][x];
                        if (p.noise_mode == KMSR_NOISE_GAUSSIAN) v += sig * pick4(z[x], i & 3);
                        else if (p.noise_mode == KMSR_NOISE_POOL) v += npl[(long long)i * p.w + xo];
                        lr[((long long)plane * p.h + i) * p.w + xo] = v;
                    }
                }
            }
#pragma unroll
            for (int m = TYB - 1; m > 0; --m)
#pragma unroll
                for (int x = 0; x < NX; ++x) acc[m][x] = acc[m - 1][x];
#pragma unroll
            for (int x = 0; x < NX; ++x) acc[0][x] = 0.f;
            q = 0;
            ++g;
        }
    }
}

__global__ void gaussian_eps_kernel(float* out, int planes, int h, int w, unsigned long long seed, unsigned offset) {
    const long long total = (long long)planes * h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % w), i = (int)((idx / w) % h), pl = (int)(idx / ((long long)w * h));
        float z[4];
        eps4(seed, offset, (unsigned)pl, (unsigned)(i >> 2), (unsigned)j, z);
        out[idx] = pick4(z, i & 3);
    }
}

__global__ void noise_crops_kernel(const float* __restrict__ geo, const float* __restrict__ den, int C, int H, int W,
                                   const int* __restrict__ offs, int n, int cs, float* __restrict__ out) {
    const long long total = (long long)n * C * cs * cs;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % cs), y = (int)((idx / cs) % cs);
        const int c = (int)((idx / ((long long)cs * cs)) % C), t = (int)(idx / ((long long)cs * cs * C));
        const long long src = ((long long)c * H + offs[2 * t] + y) * W + offs[2 * t + 1] + x;
        out[idx] = geo[src] - den[src];  // D_build_noise_pool.py:88
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Geometry {
    int F, h, w, TY, TX, TXP, ph, pw, rows;
};

int geometry(const kmsr_degrade_desc* d, Geometry* g) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad HR shape");
    if (d->kh <= 0 || d->kw <= 0 || d->kh > 31 || d->kw > 31 || !(d->kh & 1) || !(d->kw & 1))
        return fail(KMSR_E_UNSUPPORTED, "kernel size %dx%d not supported (odd, <=31)", d->kh, d->kw);
    if (d->factor < 1) return fail(KMSR_E_BADARG, "downscale factor %d < 1", d->factor);
    if (d->n_kernels < 1) return fail(KMSR_E_BADARG, "n_kernels < 1");
    if (d->down_mode == KMSR_DOWN_AVGPOOL) {
        int s = 0;
        while ((2 << s) <= d->factor) ++s;  // s = floor(log2(factor)) == int(np.log2(f)), C_30:120
        g->F = 1 << s;
        g->h = d->H >> s;
        g->w = d->W >> s;
        g->TY = d->kh + g->F - 1;
        g->TX = d->kw + g->F - 1;
    } else if (d->down_mode == KMSR_DOWN_STRIDE) {
        g->F = d->factor;
        g->h = ceil_div(d->H, g->F);
        g->w = ceil_div(d->W, g->F);
        g->TY = d->kh;
        g->TX = d->kw;
    } else {
        return fail(KMSR_E_BADARG, "bad down_mode %d", d->down_mode);
    }
    if (d->pad_mode != KMSR_PAD_REPLICATE && d->pad_mode != KMSR_PAD_ZERO) return fail(KMSR_E_BADARG, "bad pad_mode");
    if (d->noise_mode < KMSR_NOISE_NONE || d->noise_mode > KMSR_NOISE_POOL) return fail(KMSR_E_BADARG, "bad noise_mode");
    g->TXP = round_up(g->TX, 4);
    g->ph = d->kh / 2;
    g->pw = d->kw / 2;
    g->rows = d->n_kernels * d->C;
    return KMSR_OK;
}

#define KMSR_ROWSTREAM_TABLE(X) \
    X(4, 13, 13, 6, 6)          \
    X(4, 16, 16, 6, 6)          \
    X(8, 20, 20, 6, 6)          \
    X(2, 14, 14, 6, 6)          \
    X(8, 13, 13, 6, 6)          \
    X(2, 13, 13, 6, 6)

bool fast_config(const Geometry& g) {
#define X(F_, TY_, TX_, PH_, PW_) \
    if (g.F == F_ && g.TY == TY_ && g.TX == TX_ && g.ph == PH_ && g.pw == PW_) return true;
    KMSR_ROWSTREAM_TABLE(X)
#undef X
    return false;
}

bool fast_eligible(const kmsr_degrade_desc* d, const Geometry& g, const void* hr) {
    if (!fast_config(g)) return false;
    if (d->W % 4 || d->hr_stride_h % 4 || d->hr_stride_c % 4 || d->hr_stride_n % 4) return false;
    if (reinterpret_cast<uintptr_t>(hr) % 16) return false;
    if (g.h < 1 || g.w < 16) return false;  // narrow LR rows would leave most lanes idle
    return true;
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_degrade_out_shape(const kmsr_degrade_desc* d, int32_t* h, int32_t* w) {
    Geometry g;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if (h) *h = g.h;
    if (w) *w = g.w;
    return KMSR_OK;
}

extern "C" size_t kmsr_degrade_workspace_bytes(const kmsr_degrade_desc* d) {
    Geometry g;
    if (geometry(d, &g)) return 0;
    return (size_t)g.rows * g.TY * g.TXP * sizeof(float);
}

extern "C" int kmsr_degrade_uses_fast_path(const kmsr_degrade_desc* d, const void* hr) {
    Geometry g;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if (d->flags & KMSR_FLAG_FORCE_GENERIC) return 0;
    return fast_eligible(d, g, hr) ? 1 : 0;
}

extern "C" int kmsr_degrade_f32(const kmsr_degrade_desc* d, const float* hr, const float* kernels,
                                const int32_t* kernel_idx, const float* sigma, const float* noise,
                                const int64_t* noise_idx, float* lr, void* workspace, size_t workspace_bytes,
                                void* stream_) {
    Geometry g;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if (!hr || !kernels) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (g.h == 0 || g.w == 0) return KMSR_OK;  // empty output (image smaller than the pooling factor)
    if (!lr) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (!kernel_idx && d->n_kernels != 1 && d->n_kernels != d->N)
        return fail(KMSR_E_SHAPE, "n_kernels=%d must be 1 or N=%d when kernel_idx is NULL", d->n_kernels, d->N);
    if (d->noise_mode == KMSR_NOISE_GAUSSIAN && !sigma) return fail(KMSR_E_BADARG, "GAUSSIAN noise needs sigma");
    if (d->noise_mode == KMSR_NOISE_POOL) {
        if (!noise) return fail(KMSR_E_BADARG, "POOL noise needs the noise tensor");
        if (!noise_idx && d->pool_size < d->N) return fail(KMSR_E_SHAPE, "pool_size=%d < N=%d with identity index", d->pool_size, d->N);
    }
    if (d->hr_stride_h < d->W) return fail(KMSR_E_BADARG, "hr_stride_h < W");
    const size_t need = (size_t)g.rows * g.TY * g.TXP * sizeof(float);
    if (!workspace || workspace_bytes < need)
        return fail(KMSR_E_NOMEM, "workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    if (!(d->flags & KMSR_FLAG_TAPS_READY)) {
        prepare_taps_kernel<<<g.rows, 256, 0, stream>>>(kernels, d->kh, d->kw, g.F, d->down_mode == KMSR_DOWN_AVGPOOL,
                                                         d->normalize, g.TY, g.TX, g.TXP, static_cast<float*>(workspace));
        KMSR_HIP_CHECK(hipGetLastError());
    }

    DegradeParams p{};
    p.hr = hr; p.sN = d->hr_stride_n; p.sC = d->hr_stride_c; p.sH = d->hr_stride_h;
    p.N = d->N; p.C = d->C; p.H = d->H; p.W = d->W; p.h = g.h; p.w = g.w;
    p.F = g.F; p.TY = g.TY; p.TX = g.TX; p.TXP = g.TXP; p.ph = g.ph; p.pw = g.pw;
    p.taps = static_cast<const float*>(workspace);
    p.n_kernels = d->n_kernels; p.kidx = kernel_idx; p.sigma = sigma; p.noise = noise;
    p.nidx = reinterpret_cast<const long long*>(noise_idx); p.lr = lr;
    p.seed = d->seed; p.offset = d->offset; p.pad_zero = d->pad_mode == KMSR_PAD_ZERO; p.noise_mode = d->noise_mode;

    const bool fast = !(d->flags & KMSR_FLAG_FORCE_GENERIC) && fast_eligible(d, g, hr);
    if (!fast && (d->flags & KMSR_FLAG_FORCE_FAST))
        return fail(KMSR_E_UNSUPPORTED, "fast path not available for F=%d taps=%dx%d W=%d", g.F, g.TY, g.TX, d->W);
    if (!fast) {
        const long long total = (long long)d->N * d->C * g.h * g.w;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
        degrade_generic_kernel<<<blocks, 256, 0, stream>>>(p);
        KMSR_HIP_CHECK(hipGetLastError());
        return KMSR_OK;
    }

    const int NX = g.w >= 128 ? 2 : 1;
    const int SW = kWave * NX;
    p.n_strips = ceil_div(g.w, SW);
    int rows = d->rows_per_chunk;
    if (rows <= 0) {
        // enough waves to fill 256 CUs x >= 8 waves, but chunks tall enough to amortise the (TY-F)-row halo
        rows = 64;
        const long long planes_strips = (long long)d->N * d->C * p.n_strips;
        while (rows > 8 && planes_strips * ceil_div(g.h, rows) < 8192) rows >>= 1;
    }
    rows = std::max(4, rows / 4 * 4);  // the noise stream is generated per 4 output rows
    p.rows_per_chunk = rows;
    p.n_chunks = ceil_div(g.h, rows);
    const long long units = (long long)d->N * d->C * p.n_strips * p.n_chunks;
    if (units > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "batch too large for one launch");
    p.n_units = (int)units;
    p.n_blocks = ceil_div(p.n_units, kWavesPerBlock);
    const dim3 grid(p.n_blocks), block(kWave * kWavesPerBlock);
#define X(F_, TY_, TX_, PH_, PW_)                                                                    \
    if (g.F == F_ && g.TY == TY_ && g.TX == TX_ && g.ph == PH_ && g.pw == PW_) {                     \
        if (NX == 2) degrade_rowstream_kernel<F_, TY_, TX_, PH_, PW_, 2><<<grid, block, 0, stream>>>(hr, p.taps, sigma, noise, lr, p); \
        else degrade_rowstream_kernel<F_, TY_, TX_, PH_, PW_, 1><<<grid, block, 0, stream>>>(hr, p.taps, sigma, noise, lr, p);     \
    } else
    KMSR_ROWSTREAM_TABLE(X) { return fail(KMSR_E_UNSUPPORTED, "internal: no instantiation"); }
#undef X
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_gaussian_eps_f32(float* out, int32_t planes, int32_t h, int32_t w, uint64_t seed, uint32_t offset,
                                     void* stream) {
    if (!out || planes < 0 || h < 0 || w < 0) return fail(KMSR_E_BADARG, "bad argument");
    const long long total = (long long)planes * h * w;
    if (!total) return KMSR_OK;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    gaussian_eps_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(out, planes, h, w, seed, offset);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_noise_crops_f32(const float* geo, const float* den, int32_t C, int32_t H, int32_t W,
                                    const int32_t* offs, int32_t n, int32_t crop, float* out, void* stream) {
    if (!geo || !den || !offs || !out) return fail(KMSR_E_BADARG, "null pointer");
    if (crop > H || crop > W) return fail(KMSR_E_SHAPE, "crop %d larger than image %dx%d", crop, H, W);  // D:44-45
    const long long total = (long long)n * C * crop * crop;
    if (total <= 0) return KMSR_OK;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    noise_crops_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(geo, den, C, H, W, offs, n, crop, out);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
