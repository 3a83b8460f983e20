// Outer frame of the collapsed x4 upsampler (semantics: include/kmsr.h, kmsr_upsampler_frame_f32).
//
// Away from the border  up1 -> PixelShuffle -> up2 -> PixelShuffle -> tail  is one 5x5 convolution on the LR grid followed by
// PixelShuffle(4) (conv5x5_ps4.hip).  On the outermost LR row / column the intermediate layers' zero padding truncates some paths,
// which gives a DIFFERENT linear 5x5 map per border class (top, bottom, left, right, four corners; host side: sr.py reads the 8
// weight sets off the layers' impulse responses).  This kernel evaluates those maps for the 2W + 2(H-2) border LR pixels of
// every image and writes their 4x4 HR blocks into the output of the interior launch.
//
// Per class a GEMM  D[sub-pixel 16 x band][pixel] = W[k][.]^T x[pixel][k],  k = (input channel, dy, dx) = 25 F, on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32; round 2's VALU kernel spent 0.26 ms per step of config 2 on LDS reads: one 16-byte weight read per 3
// FMAs).  Workgroup = 64 border pixels of ONE class, wave = 16 of them x all bands; K streams through LDS in chunks of 160 rows
// (= 32 patch rows of 5 taps), register-staged one chunk ahead.  Both LDS arrays are row-major in k --
//   ws[k][co] (the global layout: 16-byte copies),   xs[k][pixel] at a pitch of 80 floats
// -- so every staging store is "per-thread base + immediate" (the first version's operand-shaped layouts cost ~6 address
// instructions per 4-byte store and its predicated global loads a branch each: 125 us), and the MFMA operand reads
// (lane (q, kk): ws[16 kb + 4 s + kk][16 band + q], xs[...][16 wave + q]) hit 64 different banks.
// D comes out with lane (pixel, a) holding the four HR columns of HR row a of its pixel: one 16-byte store per band.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int FPX = 64;            // border pixels per workgroup (16 per wave)
constexpr int FKC = 160;           // K rows (input channel, tap) per LDS chunk = 10 blocks of 16
constexpr int FMAXB = 8;           // bands

struct FrameParams {
    int N, F, H, W, bands;
};

// class c: 0 top, 1 bottom, 2 left, 3 right (corners excluded), 4 top-left, 5 top-right, 6 bottom-left, 7 bottom-right
__device__ __forceinline__ int class_pixels(int c, int H, int W) { return c < 2 ? W - 2 : (c < 4 ? H - 2 : 1); }
__device__ __forceinline__ void class_pixel(int c, int t, int H, int W, int& i, int& j) {
    switch (c) {
        case 0: i = 0; j = 1 + t; break;
        case 1: i = H - 1; j = 1 + t; break;
        case 2: i = 1 + t; j = 0; break;
        case 3: i = 1 + t; j = W - 1; break;
        case 4: i = 0; j = 0; break;
        case 5: i = 0; j = W - 1; break;
        case 6: i = H - 1; j = 0; break;
        default: i = H - 1; j = W - 1; break;
    }
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int FXP = 80;            // pitch of xs rows (floats): lanes (pixel q, kk) of an operand read fall into 64 different banks

template <int BANDS, bool VEC>   // VEC: W % 4 == 0 and a 16-byte aligned input -> patch rows arrive as two aligned 16-byte loads
__global__ void __launch_bounds__(256) upsampler_frame_kernel(const float* __restrict__ h, const float* __restrict__ wcls,
                                                              const float* __restrict__ bcls, float* __restrict__ y,
                                                              FrameParams p) {
    constexpr int CO = 16 * BANDS;
    __shared__ __attribute__((aligned(16))) float ws[FKC * CO];         // 160 x 48 floats = 30 KB at 3 bands
    __shared__ __attribute__((aligned(16))) float xs[FKC * FXP];        // 50 KB
    const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int per_img = class_pixels(c, H, W), total = per_img * p.N;
    const int first = blockIdx.x * FPX;
    if (first >= total) return;                    // block-uniform
    const int K = p.F * 25;
    const long long HW = (long long)H * W;
    // buffer descriptors: loads past the class's weights / outside the tensor return zero (no predication branches)
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wcls + (long long)c * K * CO), 0, (unsigned)(K * CO) * 4u, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // staging role: pixel slot sm = tid & 63 (coalesced along a top / bottom border), patch rows (tid >> 6) + 4 u
    const int sm = tid & 63, sme = first + sm, sg = tid >> 6;
    int sn = 0, si = 0, sj = 0;
    if (sme < total) {
        sn = sme / per_img;
        class_pixel(c, sme % per_img, H, W, si, sj);
    }
    // one descriptor per workgroup would need one image; pixels of a workgroup may straddle images, so the descriptor spans the
    // tensor from the first image of this workgroup's pixels and the per-lane offset carries the image (host checks the 2 GB span)
    const int n_first = min(first / per_img, p.N - 1);
    const float* hbase = h + (long long)n_first * p.F * HW;
    const long long span = ((long long)(p.N - n_first) * p.F * HW) * 4;
    const auto hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hbase), 0, (unsigned)(span > 0x7FFFFFFFll ? 0x7FFFFFFFll : span), 0x00020000);
    const unsigned img_off = (unsigned)((sn - n_first) * p.F) * (unsigned)HW * 4u;

    f32x4 acc[BANDS];
#pragma unroll
    for (int b = 0; b < BANDS; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NW4 = (FKC * CO / 4 + 255) / 256, NR = FKC / 5 / 4, NX = FKC / 4;   // per thread: weight quads, patch rows, patch values
    f32x4 wreg[NW4];
    f32x4 xq[VEC ? NR : 1][2];
    float xreg[VEC ? 1 : NX];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NW4; ++u) {
            const int e4 = tid + 256 * u;
            wreg[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, e4 < FKC * CO / 4 ? (unsigned)(k0 * CO + 4 * e4) * 4u : OOB, 0, 0));
        }
        if constexpr (VEC) {
            // a chunk is 32 patch rows (channel, dy) of 5 taps; thread (sm, sg) fetches rows sg + 4 u as the two aligned quads that
            // cover columns sj - 2 .. sj + 2
            int rg = k0 / 5 + sg;
            int ci = rg / 5, dy = rg - ci * 5;
            const int xb = (sj - 2) & ~3;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int yy = si + dy - 2;
                const bool rok = sme < total && ci < p.F && yy >= 0 && yy < H;
                const unsigned ro = img_off + (unsigned)(ci * (int)HW + yy * W + xb) * 4u;
                xq[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, rok && xb >= 0 ? ro : OOB, 0, 0));
                xq[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, rok && xb + 4 < W ? ro + 16u : OOB, 0, 0));
                dy += 4;
                if (dy >= 5) { dy -= 5; ++ci; }
            }
        } else {
            const int kg = k0 + sg;
            int ci = kg / 25, tap = kg - ci * 25;
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int yy = si + tap / 5 - 2, xx = sj + tap % 5 - 2;
                const bool ok = sme < total && ci < p.F && yy >= 0 && yy < H && xx >= 0 && xx < W;
                xreg[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, ok ? img_off + (unsigned)(ci * (int)HW + yy * W + xx) * 4u : OOB, 0, 0));
                tap += 4;
                if (tap >= 25) { tap -= 25; ++ci; }
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < NW4; ++u) {
            const int e4 = tid + 256 * u;
            if (NW4 * 256 == FKC * CO / 4 || e4 < FKC * CO / 4) *reinterpret_cast<f32x4*>(ws + 4 * e4) = wreg[u];
        }
        if constexpr (VEC) {
            const bool o1 = (sj - 2) & 1, o2 = (sj - 2) & 2;             // first tap's position inside the two quads = 2 o2 + o1
            float* xp = xs + sg * 5 * FXP + sm;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                // shift the 8 floats by o = 2 o2 + o1 in two select stages, on NAMED scalars: on an array the compiler turns
                // "o1 ? e[i + 1] : e[i]" into a dynamically indexed private array (LDS / scratch round trip), and the nested ternary of
                // the first version into one exec-mask branch per element
                const f32x4 A = xq[u][0], B = xq[u][1];
#define KMSR_SEL(C, X, Y) ((C) ? (X) : (Y))
                const float t0 = KMSR_SEL(o1, A.y, A.x), t1 = KMSR_SEL(o1, A.z, A.y), t2 = KMSR_SEL(o1, A.w, A.z), t3 = KMSR_SEL(o1, B.x, A.w),
                            t4 = KMSR_SEL(o1, B.y, B.x), t5 = KMSR_SEL(o1, B.z, B.y), t6 = KMSR_SEL(o1, B.w, B.z);
                xp[(20 * u + 0) * FXP] = KMSR_SEL(o2, t2, t0);
                xp[(20 * u + 1) * FXP] = KMSR_SEL(o2, t3, t1);
                xp[(20 * u + 2) * FXP] = KMSR_SEL(o2, t4, t2);
                xp[(20 * u + 3) * FXP] = KMSR_SEL(o2, t5, t3);
                xp[(20 * u + 4) * FXP] = KMSR_SEL(o2, t6, t4);
#undef KMSR_SEL
            }
        } else {
            float* xp = xs + sg * FXP + sm;
#pragma unroll
            for (int u = 0; u < NX; ++u) xp[4 * u * FXP] = xreg[u];
        }
    };
    load_chunk(0);
    const int q = lane & 15, kk = lane >> 4;
    const float* wrd = ws + kk * CO + q;
    const float* xrd = xs + kk * FXP + wv * 16 + q;
    for (int k0 = 0; k0 < K; k0 += FKC) {
        __syncthreads();                                                 // the previous chunk's MFMAs have read their operands
        store_chunk();
        __syncthreads();
        if (k0 + FKC < K) load_chunk(k0 + FKC);
        // ---- 10 blocks of 16 K rows, 4 MFMA steps each
#pragma unroll 2
        for (int kb = 0; kb < FKC / 16; ++kb) {
            float xv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) xv[s] = xrd[(16 * kb + 4 * s) * FXP];
#pragma unroll
            for (int b = 0; b < BANDS; ++b) {
                float wvv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) wvv[s] = wrd[(16 * kb + 4 * s) * CO + 16 * b];
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv[s], xv[s], acc[b], 0, 0, 0);
            }
        }
    }
    // lane (pixel q of this wave, HR row a = lane >> 4): acc[b][r] = sub-pixel (a, r) of band b
    const int me = first + wv * 16 + (lane & 15), a = lane >> 4;
    if (me < total) {
        const int n = me / per_img;
        int i, j;
        class_pixel(c, me % per_img, H, W, i, j);
        const long long H4 = 4ll * H, W4 = 4ll * W;
#pragma unroll
        for (int b = 0; b < BANDS; ++b) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bcls + c * CO + b * 16 + 4 * a);
            *reinterpret_cast<f32x4*>(y + (((long long)n * BANDS + b) * H4 + 4 * i + a) * W4 + 4 * j) = acc[b] + bv;
        }
    }
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_upsampler_frame_f32(const float* h, int32_t N, int32_t F, int32_t H, int32_t W, int32_t bands, const float* wcls,
                                        const float* bcls, float* y, void* stream) {
    if (!h || !wcls || !bcls || !y) return fail(KMSR_E_BADARG, "upsampler_frame: null pointer");
    if (N <= 0 || F <= 0 || H < 4 || W < 4) return fail(KMSR_E_BADARG, "upsampler_frame: bad shape (H, W >= 4)");
    if (bands < 1 || bands > FMAXB) return fail(KMSR_E_UNSUPPORTED, "upsampler_frame: 1..%d bands (got %d)", FMAXB, bands);
    if (reinterpret_cast<uintptr_t>(bcls) % 16 || reinterpret_cast<uintptr_t>(y) % 16)
        return fail(KMSR_E_BADARG, "upsampler_frame: class biases and output must be 16-byte aligned");
    // patch loads address the input through one buffer descriptor per workgroup whose 64 pixels may lie in up to 64 images (corner classes)
    if ((long long)F * H * W * 4 * std::min<long long>(N, FPX) >= (1ll << 31))
        return fail(KMSR_E_UNSUPPORTED, "upsampler_frame: min(N, 64) images of the input (%lld bytes each) exceed 2 GB", (long long)F * H * W * 4);
    FrameParams p{N, F, H, W, bands};
    const int longest = std::max(H, W) - 2;
    const dim3 grid((unsigned)ceil_div(longest * N, FPX), 8);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = W % 4 == 0 && reinterpret_cast<uintptr_t>(h) % 16 == 0;
    switch (bands) {
#define KMSR_FR(B_)                                                                                       \
    case B_:                                                                                              \
        if (vec) upsampler_frame_kernel<B_, true><<<grid, 256, 0, st>>>(h, wcls, bcls, y, p);             \
        else upsampler_frame_kernel<B_, false><<<grid, 256, 0, st>>>(h, wcls, bcls, y, p);                \
        break;
        KMSR_FR(1) KMSR_FR(2) KMSR_FR(3) KMSR_FR(4) KMSR_FR(5) KMSR_FR(6) KMSR_FR(7) KMSR_FR(8)
#undef KMSR_FR
    }
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
