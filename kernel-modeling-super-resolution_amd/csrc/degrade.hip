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
#ifndef KMSR_DMA_AUX
#define KMSR_DMA_AUX 0  // cache policy of the HR row stream (2 = nt)
#endif

#include "kmsr_noise.h"

#include <algorithm>

namespace kmsr {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kRing = 3;  // LDS row slots per wave (2 rows in flight; measured best of {3,4,6}, profiles/r01)

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
    unsigned nz_base, nz_step;  // noise plane of local (n, c) = nz_base + n * nz_step + c  (global sample index * C + c)
    int pad_zero;
    int noise_mode;
    int rows_per_chunk, n_chunks, n_strips, n_units, n_blocks;
    int n_tap_rows;             // (bank row, band) entries of the tap table; its row-reversed copy starts n_tap_rows * TY * TXP floats in
};

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
    // second copy with the tap ROWS reversed, behind the table ([gridDim.x][TY][TXP] again): the row-streaming kernel's bottom-up
    // chunks read it, so that "tap row al" is base + a literal offset for both walking directions
    float* tr = taps + ((size_t)gridDim.x + blockIdx.x) * TY * TXP;
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
        tr[(TY - 1 - a) * TXP + b] = (float)v;
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
        float acc = 0.f, acc_odd = 0.f;   // taps at even / odd column index b: the row-streaming kernel's two partial sums (v_pk_fma_f32 pairs)
        // odd row chunks accumulate tap rows bottom-up (see degrade_rowstream_kernel): keeps the two kernels bit-identical
        const bool up = ((i / p.rows_per_chunk) & 1) != 0;
        for (int al = 0; al < p.TY; ++al) {
            const int a = up ? p.TY - 1 - al : al;
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
                if (b & 1) acc_odd = fmaf(tp[a * p.TXP + b], row[x], acc_odd);
                else acc = fmaf(tp[a * p.TXP + b], row[x], acc);
            }
        }
        acc += acc_odd;
        if (p.noise_mode == KMSR_NOISE_GAUSSIAN) {
            float z[4];
            eps4(p.seed, p.offset, p.nz_base + (unsigned)n * p.nz_step + (unsigned)c, (unsigned)(i >> 2), (unsigned)j, z);
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

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// 16 bytes per lane from (base + soff + voff) into LDS at lds_dst + 16 lane; voff >= records: the hardware writes zeros.  A plain
// function on purpose: with the builtin call inside the kernel template, hipcc's HOST pass silently dropped every instantiation and
// the library ended up without the kernel's launch stubs (undefined symbol at load time; no diagnostic).
__device__ __forceinline__ void dma16(const float* base, unsigned records, float* lds_dst, unsigned voff, unsigned soff) {
#ifndef KMSR_ABL_NO_DMA   // timing-only ablation builds (tools/diag): the compute side alone
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, records, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_dst, 16, voff, soff, 0, KMSR_DMA_AUX);
#endif
}

// Row-streaming kernel.  R = LDS ring slots per wave: R-1 HR rows are in flight (LDS-DMA, no staging VGPRs)
// while one is being folded into the accumulators.
template <int F, int TY, int TX, int PH, int PW, int NX, int R>
__global__ void __launch_bounds__(kWave* kWavesPerBlock)
    degrade_rowstream_kernel(const float* __restrict__ hr, const float* __restrict__ taps_all,
                             const float* __restrict__ sigma, const float* __restrict__ noise,
                             float* __restrict__ lr, DegradeParams p) {
    using Cfg = RowCfg<F, TY, TX, PH, PW, NX>;
    constexpr int TYB = Cfg::TYB, G = Cfg::G, NWG = Cfg::NWG, OFF = Cfg::OFF, HL = Cfg::HL;
    constexpr int ROWF = Cfg::ROWF, NV = Cfg::NV, NLD = Cfg::NLD, SW = Cfg::SW, TXP = Cfg::TXP;
#ifndef KMSR_ABL_LDS_PAD
#define KMSR_ABL_LDS_PAD 0      // timing-only builds: floats of padding per ring slot = fewer resident workgroups per CU
#endif
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][R][ROWF + KMSR_ABL_LDS_PAD];   // the ONLY LDS object (guide 4a)

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
    const unsigned nz_plane = p.nz_base + (unsigned)n * p.nz_step + (unsigned)c;
    const int krow = p.kidx ? __builtin_amdgcn_readfirstlane(p.kidx[n]) : (p.n_kernels == 1 ? 0 : n);
    const float* __restrict__ taps = taps_all + (size_t)(krow * p.C + c) * (TY * TXP);
    const float* __restrict__ img = hr + n * p.sN + c * p.sC;
    const int H = p.H, W = p.W;
    const int i0 = chunk * p.rows_per_chunk;
    const int i1 = min(i0 + p.rows_per_chunk, p.h);
    const int xs0 = strip * SW;
    const int col0 = F * xs0 - HL;  // HR column of LDS float 0 (multiple of 4)
    float* ring = &lds[wid][0][0];
    const bool pad_zero = p.pad_zero != 0;

    // ---- per-lane staging map.  Float4 q of a row segment covers HR columns col0+4q .. +3; W % 4 == 0 and col0 % 4 == 0, so it
    //      is either entirely inside the image or entirely outside.  Rows are staged by BUFFER LDS-DMA on one descriptor per plane
    //      (records = the plane's extent): the lane's column is the VGPR offset, the row position the scalar offset (which gfx950's
    //      range check INCLUDES -- a one-row descriptor zeroes every row but the first).  Float4s outside the image carry an offset
    //      of 2^31 (past any plane this path accepts, whatever the row), for which the hardware writes zeros
    //      -- zero padding costs nothing, every row is exactly NLD instructions under the full exec mask (no per-instruction mask
    //      juggling, no 64-bit per-lane addresses; tools/ubench/buffer_lds_oob.hip).  Replicate padding rewrites those float4s per
    //      row from the edge pixel.  Only the last instruction's lanes past the segment end are masked off (they would write the
    //      neighbouring slot).
    constexpr unsigned kOOB = 0x80000000u;
    unsigned voff[NLD];
    bool outside[NLD];
    bool any_outside = false;
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
        const int q = lane + kWave * t;
        const int c0 = col0 + 4 * q;
        const bool in = c0 >= 0 && c0 + 3 < W;
        voff[t] = in ? (unsigned)c0 * 4u : kOOB;
        outside[t] = q < NV && !in;
        any_outside = any_outside || (__ballot(outside[t]) != 0ull);
    }
    const unsigned row_bytes = (unsigned)p.sH * 4u;              // host: (H - 1) * sH + W < 2^29 elements on this path (row offset + 2^31 does not wrap)
    const unsigned plane_bytes = ((unsigned)(H - 1) * (unsigned)p.sH + (unsigned)W) * 4u;

    // accumulators as register PAIRS: .x collects the taps at even column index b, .y the odd ones (v_pk_fma_f32 does two taps per
    // instruction -- the fold loop is VALU-issue bound: 13 x NX x ~3.25 open rows scalar FMAs per HR row before); out = .x + .y.
    // degrade_generic_kernel keeps the same two partial sums, so the two kernels stay bit-identical.
    f32x2 acc[TYB][NX];
#pragma unroll
    for (int m = 0; m < TYB; ++m)
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[m][x] = f32x2{0.f, 0.f};

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

    // Shifted HR row index s = r + PH; output row i consumes s in [F*i, F*i+TY).  The chunk is walked in LOCAL
    // coordinates (il = local output row, sl = local shifted HR row) so that odd chunks can stream bottom-up:
    // vertically adjacent chunks then read their shared TY-F halo rows at the same moment (L2 / Infinity-Cache
    // hits instead of a second trip to HBM ~150 us later).  The mirrored walk is the same recurrence with the
    // tap rows reversed: a = TY-1-a_local.
    const int nrows = i1 - i0;
    const bool up = (chunk & 1) != 0;
    const int s_top = F * (i1 - 1) + TY - 1;                    // bottom-most shifted row of the chunk
    const int sl_last = F * (nrows - 1) + TY - 1;               // last local row that carries a tap
    // local rows that are actually staged form one contiguous range [sa, sb]: zero padding skips out-of-image rows
    int sa = 0, sb = sl_last;
    if (pad_zero) {
        if (!up) { sa = max(0, PH - F * i0); sb = min(sl_last, H - 1 + PH - F * i0); }
        else { sa = max(0, s_top - (H - 1 + PH)); sb = min(sl_last, s_top - PH); }
    }

    auto issue_row = [&](int sl, int slot_off) {   // sl wave-uniform, sa <= sl <= sb; slot_off = ((sl - sa) % R) * ROWF (kept as a running offset)
        const int s = up ? s_top - sl : F * i0 + sl;
        const int r = min(max(s - PH, 0), H - 1);               // replicate padding clamps the row
        const unsigned soff = (unsigned)r * row_bytes;
        float* slot = ring + slot_off;
#pragma unroll
        for (int t = 0; t < NLD; ++t) {
            if (t < NLD - 1 || NV % kWave == 0 || lane < NV - kWave * (NLD - 1))   // literal except for the last, partial instruction
                dma16(img, plane_bytes, slot + 4 * kWave * t, voff[t], soff);
        }
    };

#pragma unroll
    for (int k = 0; k < R - 1; ++k)
        if (sa + k <= sb) issue_row(sa + k, k * ROWF);

    // ---- main loop: one GROUP = F HR rows (local shifted rows F gl .. F gl + F - 1) opens local output row gl and retires row
    //      gl - (TYB - 1).  The F phases are unrolled, so the tap row al = F m + q of open row m is a literal: no `al < TY` tests, and
    //      which open rows exist (0 <= gl - m < nrows) is one bit mask per group instead of three compares per (row, m).  Round 4
    //      (tools/ubench/smem_prefetch.hip, part C): the memory side of this kernel alone runs 36.6 us at 48 planes / 115 us at the
    //      north-star unit -- the fold's ~25 scalar instructions and one exposed scalar-cache round trip per (row, m) were what the
    //      kernel waited for.  The tap row of open row m + 1 is requested behind the first FMA of open row m (every lgkmcnt wait is
    //      a wait for zero: scalar loads return out of order), the first one of an HR row in front of that row's DMA wait.
    //      Same sums in the same order as before (and as degrade_generic_kernel): bit-identical.
    const float* __restrict__ tbase = up ? taps + (size_t)p.n_tap_rows * (TY * TXP) : taps;   // mirrored walk: the row-reversed copy
    struct TapRow { float k[TX]; };
    auto load_taps = [&](int al) {                              // al literal after unrolling; the whole padded row exists for every al < TY
        TapRow t;
        const float* __restrict__ tp = tbase + al * TXP;
#pragma unroll
        for (int b = 0; b < TX; ++b) t.k[b] = tp[b];
        return t;
    };
    const int n_groups = nrows + TYB - 1;
    int rd_off = 0;                                             // ring slot (float offset) of the next row to consume: ((sl - sa) % R) * ROWF
    for (int gl = 0; gl < n_groups; ++gl) {
        unsigned vmask = 0;                                     // bit m: open row m (local output row gl - m) exists
#pragma unroll
        for (int m = 0; m < TYB; ++m) vmask |= (gl - m >= 0 && gl - m < nrows) ? (1u << m) : 0u;
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const int sl = F * gl + q;
            if (sl >= sa && sl <= sb) {
                TapRow tk = load_taps(q);                       // open row 0 (al = q < F <= TY): in flight across the DMA wait
                if (sl + R - 1 <= sb) issue_row(sl + R - 1, rd_off == 0 ? (R - 1) * ROWF : rd_off - ROWF);   // refills the slot consumed in the previous iteration
                // this row has landed (later rows may still be in flight); steady state = a literal count, no switch
                if (sb - sl >= R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * NLD) : "memory");
                else if (R > 2 && sb - sl == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the chunk's last row (or a deeper ring's tail: stricter than needed)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float* slot = ring + rd_off;
                rd_off = rd_off == (R - 1) * ROWF ? 0 : rd_off + ROWF;
                if (!pad_zero && any_outside) {
                    // replicate padding: out-of-image float4s take the nearest image pixel of this row (already in LDS)
                    int ln = lane;
                    asm volatile("" : "+v"(ln));                // lane-derived offsets live only here (not across the whole loop)
#pragma unroll
                    for (int t = 0; t < NLD; ++t) {
                        if (outside[t]) {
                            const int c0 = col0 + 4 * (ln + kWave * t);
                            const float e = slot[(c0 < 0 ? 0 : W - 1) - col0];
                            *reinterpret_cast<f32x4*>(slot + 4 * (ln + kWave * t)) = f32x4{e, e, e, e};
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                // ---- each lane reads the windows of its NX outputs (neighbour lanes overlap: that is the halo share)
                float win[NX][NWG * G];
#pragma unroll
                for (int x = 0; x < NX; ++x) {
                    const int base = F * (lane + kWave * x) + (HL - PW - OFF);
#pragma unroll
                    for (int t = 0; t < NWG; ++t) {
                        using V = typename VecT<G>::type;
                        const V v = *reinterpret_cast<const V*>(slot + base + G * t);
                        if constexpr (G == 4) {
                            win[x][4 * t + 0] = v.x; win[x][4 * t + 1] = v.y; win[x][4 * t + 2] = v.z; win[x][4 * t + 3] = v.w;
                        } else if constexpr (G == 2) {
                            win[x][2 * t + 0] = v.x; win[x][2 * t + 1] = v.y;
                        } else {
                            win[x][t] = v;
                        }
                    }
                }
                // ---- fold the row into every open output row (local tap row F*m + q of slot m)
                static_assert(OFF % 2 == 0, "window pairs (OFF + b, OFF + b + 1), b even, must be the aligned halves of the LDS reads");
#pragma unroll
                for (int m = 0; m < TYB; ++m) {
                    if (F * m + q < TY) {                       // literal
                        const bool more = F * (m + 1) + q < TY; // literal: open row m + 1 has a tap row in this phase
                        TapRow tn = tk;
                        auto fold_pair = [&](int b) {
#ifdef KMSR_ABL_NO_FOLD   // timing-only ablation builds (tools/diag): the memory side + tap loads + window reads alone
                            if (b > 0) return;
#endif
                            if (b + 1 < TX) {
                                const f32x2 kv = f32x2{tk.k[b], tk.k[b + 1]};
#pragma unroll
                                for (int x = 0; x < NX; ++x)
                                    acc[m][x] = __builtin_elementwise_fma(kv, f32x2{win[x][OFF + b], win[x][OFF + b + 1]}, acc[m][x]);
                            } else {        // odd TX: the last tap (even index) joins the even partial sum; never multiply a padded zero tap
                                const float kv = tk.k[b];       // with a pixel outside the window (0 x NaN)
#pragma unroll
                                for (int x = 0; x < NX; ++x) acc[m][x].x = fmaf(kv, win[x][OFF + b], acc[m][x].x);
                            }
                        };
                        if ((vmask >> m) & 1u) {               // wave-uniform
                            asm volatile("");                   // keep it a branch (if-converted, every accumulator gets a select per row)
                            fold_pair(0);                       // first use of tk: the wait for it (and for the window reads) sits here
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) tn = load_taps(F * (m + 1) + q);   // next open row's taps: requested under this row's FMAs
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int b = 2; b < TX; b += 2) fold_pair(b);
                        } else if (more) {
                            tn = load_taps(F * (m + 1) + q);
                        }
                        tk = tn;
                    }
                }
                // the LDS reads above must retire before a later iteration's DMA overwrites this slot
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        {
            // ---- group finished: slot TYB-1 holds local output row il = gl-(TYB-1), complete
            const int il = gl - (TYB - 1);
            if (il >= 0 && il < nrows) {
                const int i = up ? i1 - 1 - il : i0 + il;
#ifdef KMSR_ABL_NO_NOISE   // timing-only ablation build: the Gaussian stream is never generated (z stays 0)
                if (false) {
#else
                if (p.noise_mode == KMSR_NOISE_GAUSSIAN && (il == 0 || (i & 3) == (up ? 3 : 0))) {
#endif
                    unsigned long long sd = p.seed;
                    asm volatile("" : "+s"(sd));   // (same for the seed: its 20 Philox round keys were hoisted and spilled to VGPR lanes)
                    int ln = lane;
                    asm volatile("" : "+v"(ln));   // derived from the lane id HERE: hoisted out of the group loop, the Philox key products and
#pragma unroll                                     // Box-Muller constants took 25 VGPRs for the whole kernel (103 -> 4 instead of 6 waves per SIMD)
                    for (int x = 0; x < NX; ++x)
                        eps4(sd, p.offset, nz_plane, (unsigned)(i >> 2), (unsigned)(xs0 + ln + kWave * x), z[x]);
                }
                // wave-uniform row bases + a 32-bit lane index: scalar-base addressing, no 64-bit per-lane column kept across the loop
                float* __restrict__ orow = lr + ((long long)plane * p.h + i) * p.w + xs0;
                const float* __restrict__ nrow = p.noise_mode == KMSR_NOISE_POOL ? npl + (long long)i * p.w + xs0 : nullptr;
#pragma unroll
                for (int x = 0; x < NX; ++x) {
                    const unsigned xl = (unsigned)(lane + kWave * x);
                    if (xs0 + (int)xl < p.w) {
                        float v = acc[TYB - 1][x].x + acc[TYB - 1][x].y;
                        if (p.noise_mode == KMSR_NOISE_GAUSSIAN) v += sig * pick4(z[x], i & 3);
                        else if (p.noise_mode == KMSR_NOISE_POOL) v += nrow[xl];
                        orow[xl] = v;
                    }
                }
            }
#pragma unroll
            for (int m = TYB - 1; m > 0; --m)
#pragma unroll
                for (int x = 0; x < NX; ++x) acc[m][x] = acc[m - 1][x];
#pragma unroll
            for (int x = 0; x < NX; ++x) acc[0][x] = f32x2{0.f, 0.f};
        }
    }
}

__global__ void gaussian_eps_kernel(float* out, int planes, int h, int w, unsigned long long seed, unsigned offset,
                                    unsigned plane_base) {
    const long long total = (long long)planes * h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % w), i = (int)((idx / w) % h), pl = (int)(idx / ((long long)w * h));
        float z[4];
        eps4(seed, offset, plane_base + (unsigned)pl, (unsigned)(i >> 2), (unsigned)j, z);
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
    if ((long long)(d->H - 1) * d->hr_stride_h + d->W >= (1ll << 29)) return false;   // the row position is a 32-bit byte offset of the plane's buffer descriptor
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
    return 2 * (size_t)g.rows * g.TY * g.TXP * sizeof(float);    // tap table + its row-reversed copy
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
    const size_t need = 2 * (size_t)g.rows * g.TY * g.TXP * sizeof(float);
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
    p.n_tap_rows = g.rows;
    p.n_kernels = d->n_kernels; p.kidx = kernel_idx; p.sigma = sigma; p.noise = noise;
    p.nidx = reinterpret_cast<const long long*>(noise_idx); p.lr = lr;
    p.seed = d->seed; p.offset = d->offset;
    p.nz_step = (d->noise_sample_step ? d->noise_sample_step : 1u) * (unsigned)d->C;
    p.nz_base = d->noise_sample_base * (unsigned)d->C;
    p.pad_zero = d->pad_mode == KMSR_PAD_ZERO; p.noise_mode = d->noise_mode;

    // Row-chunk height.  It also fixes the summation direction per output row (odd chunks accumulate tap rows
    // bottom-up), so it is chosen identically for both kernels.
    const int NX_auto = g.w >= 128 ? 2 : 1;
    int rows = d->rows_per_chunk;
    if (rows <= 0) {
        // chunks tall enough to amortise the (TY-F)-row vertical halo, but enough waves to cover 256 CUs
        // several times over (measured: 16 rows is the sweet spot at >= 4k waves, profiles/r01)
        rows = 16;
        const long long planes_strips = (long long)d->N * d->C * ceil_div(g.w, kWave * NX_auto);
        while (rows > 4 && planes_strips * ceil_div(g.h, rows) < 4096) rows >>= 1;
    }
    rows = std::max(4, rows / 4 * 4);  // the noise stream is generated per 4 output rows
    p.rows_per_chunk = rows;
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

    const int NX = NX_auto;
    const int SW = kWave * NX;
    p.n_strips = ceil_div(g.w, SW);
    p.n_chunks = ceil_div(g.h, rows);
    const long long units = (long long)d->N * d->C * p.n_strips * p.n_chunks;
    if (units > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "batch too large for one launch");
    p.n_units = (int)units;
    p.n_blocks = ceil_div(p.n_units, kWavesPerBlock);
    const dim3 grid(p.n_blocks), block(kWave * kWavesPerBlock);
    // Resident workgroups per CU.  A grid that does not fill every CU to its LDS limit is packed unevenly by the dispatcher (the
    // north-star unit: 1280 workgroups on 256 CUs at 6 per CU = some CUs carry 24 waves and set the kernel's time, others idle).
    // Unreferenced dynamic LDS caps the residency at ceil(workgroups / CUs), which spreads them evenly: 151 -> 141 us on that unit
    // (profiles/r04_degrade_occupancy.log).
    auto spread_lds = [&](size_t static_bytes) -> size_t {
        const int n_cu = 256, lds_cu = 160 * 1024;
        const int target = ceil_div(p.n_blocks, n_cu);
        if (static_bytes == 0 || target >= (int)(lds_cu / static_bytes)) return 0;
        const size_t per = std::min<size_t>((size_t)(lds_cu / target) / 1024 * 1024 - 1024, 64 * 1024);   // 1 KB of slack: an exact fit (5 x 32 KB) left only 4 resident
        return per > static_bytes ? per - static_bytes : 0;
    };
#define KMSR_LAUNCH(F_, TY_, TX_, PH_, PW_, NX_, R_)                                                                        \
    degrade_rowstream_kernel<F_, TY_, TX_, PH_, PW_, NX_, R_>                                                                \
        <<<grid, block, spread_lds(sizeof(float) * kWavesPerBlock * R_ * RowCfg<F_, TY_, TX_, PH_, PW_, NX_>::ROWF), stream>>>( \
            hr, p.taps, sigma, noise, lr, p)
#define X(F_, TY_, TX_, PH_, PW_)                                                      \
    if (g.F == F_ && g.TY == TY_ && g.TX == TX_ && g.ph == PH_ && g.pw == PW_) {       \
        if (NX == 2) KMSR_LAUNCH(F_, TY_, TX_, PH_, PW_, 2, kRing);                    \
        else KMSR_LAUNCH(F_, TY_, TX_, PH_, PW_, 1, kRing);                            \
    } else
    KMSR_ROWSTREAM_TABLE(X) { return fail(KMSR_E_UNSUPPORTED, "internal: no instantiation"); }
#undef X
#undef KMSR_LAUNCH
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

extern "C" int kmsr_gaussian_eps_f32(float* out, int32_t planes, int32_t h, int32_t w, uint64_t seed, uint32_t offset,
                                     uint32_t plane_base, void* stream) {
    if (!out || planes < 0 || h < 0 || w < 0) return fail(KMSR_E_BADARG, "bad argument");
    const long long total = (long long)planes * h * w;
    if (!total) return KMSR_OK;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    gaussian_eps_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(out, planes, h, w, seed, offset, plane_base);
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
