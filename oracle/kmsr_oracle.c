/*
 * kmsr_oracle.c -- CPU restatement of the reference's hot-path algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker / the reported CPU baseline -- never as a fallback for the HIP path.
 *
 * Every function cites the reference file:line it restates (paths relative to the upstream
 * repo Zhiyyeah/Kernel-Modeling-Super-Resolution).  The restatement is pinned against golden
 * vectors produced by the reference itself (tests/golden/make_golden.py -> tests/golden/ npz files,
 * checked by tests/test_oracle_golden.py).  It deliberately follows the reference's ORDER of
 * operations (full-resolution blur, then the 2x2 pooling cascade) rather than the fused
 * formulation the HIP kernels use, so the two are independent implementations.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp; no third-party dependencies)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OK 0
#define ORACLE_E_BADARG (-1)
#define ORACLE_E_NOMEM (-4)

int oracle_version(void) { return 1; }

int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---------------------------------------------------------------------------------------------
 * Per-band kernel normalisation.  C_30apply_kernel_to_landsat.py:93-97 (== C_31:74-78):
 *   kernel_normalized[i] = kernel[i] / kernel[i].sum()   only if the sum is > 0
 * ------------------------------------------------------------------------------------------- */
void oracle_normalize_kernel(const float* k, int C, int kh, int kw, float* out) {
    for (int c = 0; c < C; ++c) {
        const float* kc = k + (size_t)c * kh * kw;
        float* oc = out + (size_t)c * kh * kw;
        float s = 0.f;
        for (int i = 0; i < kh * kw; ++i) s += kc[i];
        if (s > 0.f)
            for (int i = 0; i < kh * kw; ++i) oc[i] = kc[i] / s;
        else
            for (int i = 0; i < kh * kw; ++i) oc[i] = kc[i];
    }
}

/* ---------------------------------------------------------------------------------------------
 * V-A "apply-kernel" degradation.  C_30apply_kernel_to_landsat.py:68-124 (== C_31:59-97).
 *   img [C,H,W]; kernel [C,kh,kw] (caller has already broadcast a 2-D kernel, C_30:83-85)
 *   1. normalise per band (:93-97)
 *   2. replicate-pad kh//2, kw//2 (:107-109)
 *   3. depthwise cross-correlation at FULL resolution (:112-117)
 *   4. int(log2(f)) times avg_pool2d(2,2), each flooring odd sizes (:120-122)
 *   out must hold C * (H>>s) * (W>>s) floats, s = int(log2(f)).
 * ------------------------------------------------------------------------------------------- */
int oracle_apply_kernel_degradation(const float* img, int C, int H, int W, const float* kernel, int kh, int kw,
                                    int downscale_factor, float* out) {
    if (!img || !kernel || !out || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || downscale_factor < 1)
        return ORACLE_E_BADARG;
    int steps = (int)log2((double)downscale_factor); /* int(np.log2(f)) */
    float* kn = (float*)malloc(sizeof(float) * (size_t)C * kh * kw);
    float* cur = (float*)malloc(sizeof(float) * (size_t)C * H * W);
    float* nxt = (float*)malloc(sizeof(float) * (size_t)C * H * W);
    if (!kn || !cur || !nxt) { free(kn); free(cur); free(nxt); return ORACLE_E_NOMEM; }
    oracle_normalize_kernel(kernel, C, kh, kw, kn);
    const int ph = kh / 2, pw = kw / 2;
    /* F.pad(replicate) pads ph on both sides; conv2d(padding=0) then yields H+2ph-kh+1 rows (== H for odd kh) */
    const int Ho = H + 2 * ph - kh + 1, Wo = W + 2 * pw - kw + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Ho; ++y) {
            const float* ic = img + (size_t)c * H * W;
            const float* kc = kn + (size_t)c * kh * kw;
            float* oc = cur + (size_t)c * Ho * Wo + (size_t)y * Wo;
            for (int x = 0; x < Wo; ++x) {
                float acc = 0.f;
                for (int u = 0; u < kh; ++u) {
                    const float* row = ic + (size_t)clampi(y + u - ph, 0, H - 1) * W;
                    for (int v = 0; v < kw; ++v) acc += kc[u * kw + v] * row[clampi(x + v - pw, 0, W - 1)];
                }
                oc[x] = acc;
            }
        }
    int h = Ho, w = Wo;
    for (int s = 0; s < steps; ++s) {
        int h2 = h / 2, w2 = w / 2;
#pragma omp parallel for collapse(2) schedule(static)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h2; ++y) {
                const float* a = cur + (size_t)c * h * w + (size_t)(2 * y) * w;
                float* o = nxt + (size_t)c * h2 * w2 + (size_t)y * w2;
                for (int x = 0; x < w2; ++x)
                    o[x] = (a[2 * x] + a[2 * x + 1] + a[w + 2 * x] + a[w + 2 * x + 1]) * 0.25f;
            }
        float* t = cur; cur = nxt; nxt = t;
        h = h2; w = w2;
    }
    memcpy(out, cur, sizeof(float) * (size_t)C * h * w);
    free(kn); free(cur); free(nxt);
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * V-B deterministic part.  muti_kernel/train_gemini.py:124-134:
 *   out = F.conv2d(x.view(1,B*C,H,W), k.view(B*C,1,kh,kw), padding=k//2, groups=B*C)[..., ::f, ::f]
 *   zero padding, cross-correlation, NO kernel re-normalisation; output index i samples row f*i.
 *   x [B,C,H,W]; kernels [B,C,kh,kw]; out [B,C,ceil(H/f),ceil(W/f)]
 * ------------------------------------------------------------------------------------------- */
int oracle_vb_blur_stride(const float* x, int B, int C, int H, int W, const float* kernels, int kh, int kw, int f,
                          float* out) {
    if (!x || !kernels || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || f < 1) return ORACLE_E_BADARG;
    const int ph = kh / 2, pw = kw / 2;
    const int h = (H + f - 1) / f, w = (W + f - 1) / f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int p = 0; p < B * C; ++p)
        for (int i = 0; i < h; ++i) {
            const float* ic = x + (size_t)p * H * W;
            const float* kc = kernels + (size_t)p * kh * kw;
            float* o = out + (size_t)p * h * w + (size_t)i * w;
            for (int j = 0; j < w; ++j) {
                float acc = 0.f;
                for (int u = 0; u < kh; ++u) {
                    int yy = f * i + u - ph;
                    if (yy < 0 || yy >= H) continue;
                    for (int v = 0; v < kw; ++v) {
                        int xx = f * j + v - pw;
                        if (xx < 0 || xx >= W) continue;
                        acc += kc[u * kw + v] * ic[(size_t)yy * W + xx];
                    }
                }
                o[j] = acc;
            }
        }
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Gaussian noise stream (train_gemini.py:137-138 `randn_like(out) * sigma[b,c]`,
 * muti_kernel/networks.py:221-224).  torch's CPU generator cannot be reproduced on a device, so the
 * build DEFINES a counter-based stream (DESIGN.md "noise stream"): for output element
 * (plane p = b*C+c, row i, col j)
 *     (x0,x1,x2,x3) = Philox4x32-10(key=(seed_lo,seed_hi), ctr=(j, i>>2, p, offset))
 *     u_a = ((x_a >> 8) + 1) * 2^-24  in (0,1],  u_b = (x_b >> 8) * 2^-24 in [0,1)
 *     (z0,z1) = BoxMuller(u(x0),u(x1)),  (z2,z3) = BoxMuller(u(x2),u(x3));   eps = z[i & 3]
 * The integer stream is bit-exact between this file and the HIP kernel; the transcendental part
 * agrees to ~1e-6.
 * ------------------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox_u32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out4) {
    philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), out4);
}

static inline float oracle_eps(uint64_t seed, uint32_t offset, uint32_t plane, uint32_t i, uint32_t j) {
    uint32_t r[4];
    philox4x32_10(j, i >> 2, plane, offset, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    int pair = (i & 2) ? 2 : 0;
    double u1 = ((double)(r[pair] >> 8) + 1.0) * (1.0 / 16777216.0);
    double u2 = (double)(r[pair + 1] >> 8) * (1.0 / 16777216.0);
    double rad = sqrt(-2.0 * log(u1));
    double ang = 6.283185307179586476925 * u2;
    return (float)((i & 1) ? rad * sin(ang) : rad * cos(ang));
}

/* out[p,i,j] += sigma[p] * eps(plane_base + p,i,j); sigma is [planes].  plane_base = (global index of the first sample) * C:
 * a shard of a larger batch draws the noise of its GLOBAL planes (results invariant to the world size). */
int oracle_add_gaussian_noise(float* out, int planes, int h, int w, const float* sigma, uint64_t seed,
                              uint32_t offset, uint32_t plane_base) {
    if (!out || !sigma) return ORACLE_E_BADARG;
#pragma omp parallel for collapse(2) schedule(static)
    for (int p = 0; p < planes; ++p)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j)
                out[((size_t)p * h + i) * w + j] += sigma[p] * oracle_eps(seed, offset, plane_base + (uint32_t)p, (uint32_t)i, (uint32_t)j);
    return ORACLE_OK;
}

/* V-C noise-pool add.  E_make_train_data.py:65-74:  lr = blurred + noise_pool[idx]
 * blurred [N,C,h,w], pool [P,C,h,w], idx [N] (drawn by the caller from np.random.randint) */
int oracle_add_pool_noise(float* out, const float* blurred, int N, int C, int h, int w, const float* pool, int P,
                          const int64_t* idx) {
    if (!out || !blurred || !pool || !idx) return ORACLE_E_BADARG;
    size_t e = (size_t)C * h * w;
    for (int n = 0; n < N; ++n) {
        if (idx[n] < 0 || idx[n] >= P) return ORACLE_E_BADARG;
        const float* a = blurred + n * e;
        const float* b = pool + (size_t)idx[n] * e;
        for (size_t t = 0; t < e; ++t) out[n * e + t] = a[t] + b[t];
    }
    return ORACLE_OK;
}

/* D_build_noise_pool.py:88 + :49-51: noise = geo - denoised; crop [top:top+cs, left:left+cs] */
int oracle_noise_crop(const float* geo, const float* den, int C, int H, int W, int top, int left, int cs, float* out) {
    if (top < 0 || left < 0 || top + cs > H || left + cs > W) return ORACLE_E_BADARG;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < cs; ++y)
            for (int x = 0; x < cs; ++x) {
                size_t s = ((size_t)c * H + top + y) * W + left + x;
                out[((size_t)c * cs + y) * cs + x] = geo[s] - den[s];
            }
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Generic dense conv2d (cross-correlation), NCHW, zero padding, stride, optional bias / ReLU.
 * Restates the ATen call sites on the path: SelectorNet nn.Conv2d(k=3,s=2,p=1)
 * (train_gemini.py:18-33), ConditionEncoder (muti_kernel/networks.py:31-38), and the build-defined
 * SR network's 3x3 convs.  fp32 accumulation, (ci, u, v) order.
 * ------------------------------------------------------------------------------------------- */
int oracle_conv2d(const float* x, int N, int Ci, int H, int W, const float* wgt, const float* bias, int Co, int kh,
                  int kw, int stride, int pad, int relu, float* out) {
    if (!x || !wgt || !out || stride < 1) return ORACLE_E_BADARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co) {
            float* o = out + ((size_t)n * Co + co) * Ho * Wo;
            const float b = bias ? bias[co] : 0.f;
            for (size_t t = 0; t < (size_t)Ho * Wo; ++t) o[t] = b;
            for (int ci = 0; ci < Ci; ++ci) {
                const float* xi = x + ((size_t)n * Ci + ci) * H * W;
                const float* wk = wgt + ((size_t)co * Ci + ci) * kh * kw;
                for (int u = 0; u < kh; ++u)
                    for (int v = 0; v < kw; ++v) {
                        const float wv = wk[u * kw + v];
                        for (int y = 0; y < Ho; ++y) {
                            int yy = y * stride + u - pad;
                            if (yy < 0 || yy >= H) continue;
                            const float* row = xi + (size_t)yy * W;
                            float* orow = o + (size_t)y * Wo;
                            int x0 = 0, x1 = Wo;
                            while (x0 < Wo && x0 * stride + v - pad < 0) ++x0;
                            while (x1 > x0 && (x1 - 1) * stride + v - pad >= W) --x1;
                            if (stride == 1) {
                                const float* r = row + v - pad;
                                for (int xx = x0; xx < x1; ++xx) orow[xx] += wv * r[xx];
                            } else {
                                for (int xx = x0; xx < x1; ++xx) orow[xx] += wv * row[xx * stride + v - pad];
                            }
                        }
                    }
            }
            if (relu)
                for (size_t t = 0; t < (size_t)Ho * Wo; ++t) o[t] = o[t] > 0.f ? o[t] : 0.f;
        }
    return ORACLE_OK;
}

/* nn.PixelShuffle(r): out[n, c, h*r+i, w*r+j] = in[n, c*r*r + i*r + j, h, w] */
int oracle_pixel_shuffle(const float* x, int N, int Cin, int H, int W, int r, float* out) {
    if (Cin % (r * r)) return ORACLE_E_BADARG;
    int Co = Cin / (r * r);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < Co; ++c)
            for (int i = 0; i < r; ++i)
                for (int j = 0; j < r; ++j) {
                    const float* s = x + ((size_t)n * Cin + c * r * r + i * r + j) * H * W;
                    for (int h = 0; h < H; ++h)
                        for (int w = 0; w < W; ++w)
                            out[(((size_t)n * Co + c) * H * r + h * r + i) * W * r + w * r + j] = s[(size_t)h * W + w];
                }
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * KernelGAN kernel composition.  muti_kernel/networks.py:155-196 (== single_kernel/networks.py:125-157).
 *   conv_kernel(A,B) = full 2-D convolution (F.conv2d(A, flip(B), padding=b-1), :155-161)
 *   K0 = s0*W0 ; K_{l}[co,ci] = sum_cm fullconv((s_l*W_l)[co,cm], K_{l-1}[cm,ci])   (:171-188)
 *   k = mean_{co,ci}(K_last); k = max(k,0); k /= (sum(k) if sum(k) > 1e-12 else 1)   (:189-194)
 * One band of one sample per call.
 *   n_layers layers; layer l has weights w[l] of shape [co[l], ci[l], ks[l], ks[l]] and an optional
 *   per-output-channel scale s[l] (NULL = ones; muti variant :168-169 `w = w_base * s`).
 *   out: [K,K] with K = 1 + sum(ks[l]-1)  (13 for ks = 7,5,3,1,1,1)
 * ------------------------------------------------------------------------------------------- */
int oracle_compose_kernel(int n_layers, const float* const* w, const float* const* s, const int* co, const int* ci,
                          const int* ks, float* out) {
    if (n_layers < 1 || !w || !co || !ci || !ks || !out) return ORACLE_E_BADARG;
    int cin0 = ci[0];
    int size = ks[0];
    size_t n = (size_t)co[0] * cin0 * size * size;
    float* cur = (float*)malloc(n * sizeof(float));
    if (!cur) return ORACLE_E_NOMEM;
    for (int o = 0; o < co[0]; ++o)
        for (size_t t = 0; t < (size_t)cin0 * size * size; ++t)
            cur[(size_t)o * cin0 * size * size + t] = w[0][(size_t)o * cin0 * size * size + t] * (s && s[0] ? s[0][o] : 1.f);
    int cmid = co[0];
    for (int l = 1; l < n_layers; ++l) {
        if (ci[l] != cmid) { free(cur); return ORACLE_E_BADARG; }
        int k = ks[l], nsize = size + k - 1;
        float* nxt = (float*)calloc((size_t)co[l] * cin0 * nsize * nsize, sizeof(float));
        if (!nxt) { free(cur); return ORACLE_E_NOMEM; }
#pragma omp parallel for schedule(static)
        for (int o = 0; o < co[l]; ++o) {
            float sc = (s && s[l]) ? s[l][o] : 1.f;
            for (int i = 0; i < cin0; ++i) {
                float* dst = nxt + ((size_t)o * cin0 + i) * nsize * nsize;
                for (int m = 0; m < cmid; ++m) {
                    const float* A = w[l] + ((size_t)o * cmid + m) * k * k;      /* [k,k]       */
                    const float* Bm = cur + ((size_t)m * cin0 + i) * size * size; /* [size,size] */
                    for (int a = 0; a < k; ++a)
                        for (int b = 0; b < k; ++b) {
                            float av = A[a * k + b] * sc;
                            for (int y = 0; y < size; ++y)
                                for (int x = 0; x < size; ++x) dst[(a + y) * nsize + (b + x)] += av * Bm[y * size + x];
                        }
                }
            }
        }
        free(cur);
        cur = nxt; size = nsize; cmid = co[l];
    }
    /* mean over (co_last, ci0), clamp, normalise */
    int cnt = cmid * cin0;
    double total = 0.0;
    for (int t = 0; t < size * size; ++t) {
        float acc = 0.f;
        for (int q = 0; q < cnt; ++q) acc += cur[(size_t)q * size * size + t];
        acc /= (float)cnt;
        if (acc < 0.f) acc = 0.f;
        out[t] = acc;
    }
    float ssum = 0.f;
    for (int t = 0; t < size * size; ++t) ssum += out[t];
    (void)total;
    if (ssum <= 1e-12f) ssum = 1.f;
    for (int t = 0; t < size * size; ++t) out[t] /= ssum;
    free(cur);
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * KernelGAN generator forward, one band of one sample.
 * muti_kernel/networks.py:101-117 (_apply_conv_chain) / single_kernel/networks.py:96-104:
 *   for each layer: reflect-pad k//2, bias-free conv; then 3x AvgPool2d(2,2).
 *   x [H,W] -> out [H/8, W/8]
 * ------------------------------------------------------------------------------------------- */
static inline int reflecti(int v, int n) {
    if (v < 0) v = -v;
    if (v >= n) v = 2 * (n - 1) - v;
    return v;
}

int oracle_chain_forward(const float* x, int H, int W, int n_layers, const float* const* w, const float* const* s,
                         const int* co, const int* ci, const int* ks, int n_pools, float* out) {
    if (!x || !out || n_layers < 1) return ORACLE_E_BADARG;
    int cc = ci[0];
    float* cur = (float*)malloc(sizeof(float) * (size_t)cc * H * W);
    if (!cur) return ORACLE_E_NOMEM;
    memcpy(cur, x, sizeof(float) * (size_t)cc * H * W);
    for (int l = 0; l < n_layers; ++l) {
        int k = ks[l], p = k / 2;
        if (ci[l] != cc) { free(cur); return ORACLE_E_BADARG; }
        float* nxt = (float*)calloc((size_t)co[l] * H * W, sizeof(float));
        if (!nxt) { free(cur); return ORACLE_E_NOMEM; }
#pragma omp parallel for schedule(static)
        for (int o = 0; o < co[l]; ++o) {
            float sc = (s && s[l]) ? s[l][o] : 1.f;
            float* dst = nxt + (size_t)o * H * W;
            for (int m = 0; m < cc; ++m) {
                const float* src = cur + (size_t)m * H * W;
                const float* wk = w[l] + ((size_t)o * cc + m) * k * k;
                for (int u = 0; u < k; ++u)
                    for (int v = 0; v < k; ++v) {
                        float wv = wk[u * k + v] * sc;
                        for (int y = 0; y < H; ++y) {
                            const float* row = src + (size_t)reflecti(y + u - p, H) * W;
                            float* drow = dst + (size_t)y * W;
                            for (int xx = 0; xx < W; ++xx) drow[xx] += wv * row[reflecti(xx + v - p, W)];
                        }
                    }
            }
        }
        free(cur);
        cur = nxt; cc = co[l];
    }
    int h = H, wd = W;
    for (int t = 0; t < n_pools; ++t) {
        int h2 = h / 2, w2 = wd / 2;
        float* nxt = (float*)malloc(sizeof(float) * (size_t)cc * h2 * w2);
        for (int c = 0; c < cc; ++c)
            for (int y = 0; y < h2; ++y)
                for (int xx = 0; xx < w2; ++xx) {
                    const float* a = cur + (size_t)c * h * wd + (size_t)(2 * y) * wd + 2 * xx;
                    nxt[((size_t)c * h2 + y) * w2 + xx] = (a[0] + a[1] + a[wd] + a[wd + 1]) * 0.25f;
                }
        free(cur);
        cur = nxt; h = h2; wd = w2;
    }
    memcpy(out, cur, sizeof(float) * (size_t)cc * h * wd);
    free(cur);
    return ORACLE_OK;
}
