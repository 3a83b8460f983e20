/*
 * kmsr.h -- C ABI of libkmsr_hip.so, the MI355X (gfx950) implementation of the
 * degradation-synthesis + SR-CNN hot path of Zhiyyeah/Kernel-Modeling-Super-Resolution.
 *
 * The reference has no FFI: its boundary is a set of Python callables operating on torch / numpy
 * arrays.  Each entry point below is what a binding for one of those callables would call; the
 * reference interface it replaces is cited as file:line (paths relative to the upstream repo).
 * Python-side stubs a maintainer would add are shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - All tensor pointers are DEVICE pointers (HBM) unless a parameter says "host".
 *   - The caller owns every buffer, including the workspace (size it with the *_workspace_bytes query).
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  Calls are asynchronous with
 *     respect to the host and re-entrant; the library keeps no global mutable state.
 *   - Return value: KMSR_OK (0) or a negative KMSR_E_* code; kmsr_last_error() returns a
 *     thread-local message for the last failing call on this thread.  Nothing throws or aborts.
 */
#ifndef KMSR_H
#define KMSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMSR_VERSION 101 /* 0.1.1: noise_sample_base / noise_sample_step, kmsr_gaussian_eps_f32 plane_base */

#define KMSR_OK 0
#define KMSR_E_BADARG (-1)      /* null pointer / bad enum / misaligned         -> Python ValueError   */
#define KMSR_E_SHAPE (-2)       /* band / size mismatch (C_30:88 assert)         -> AssertionError      */
#define KMSR_E_HIP (-3)         /* a HIP runtime call failed                    -> RuntimeError        */
#define KMSR_E_RCCL (-4)        /* reserved for the collective layer                                   */
#define KMSR_E_NOMEM (-5)       /* workspace too small                          -> MemoryError         */
#define KMSR_E_UNSUPPORTED (-6) /* configuration outside the implemented range  -> NotImplementedError */

int kmsr_version(void);
const char* kmsr_last_error(void);
/* number of visible HIP devices, or a negative error code */
int kmsr_device_count(void);

/* ============================================================================================
 * Fused degradation: blur (depthwise cross-correlation) + downsample + noise, one launch per batch.
 *
 * Replaces
 *   apply_kernel_degradation(img, kernel, downscale_factor)   C_30apply_kernel_to_landsat.py:68-124
 *                                                             C_31apply_muti_kernel_to_landsat.py:59-97
 *       -> down=KMSR_DOWN_AVGPOOL, pad=KMSR_PAD_REPLICATE, normalize=1, noise=KMSR_NOISE_NONE
 *   ContentAdaptiveDegradation.forward steps 3-5              muti_kernel/train_gemini.py:117-138
 *       -> down=KMSR_DOWN_STRIDE,  pad=KMSR_PAD_ZERO,      normalize=0, noise=KMSR_NOISE_GAUSSIAN
 *   add_noise(blurred, noise_pool)                            E_make_train_data.py:65-74
 *       -> noise=KMSR_NOISE_POOL (fused into the epilogue of either of the above)
 *   NoiseEstimator.forward                                    muti_kernel/networks.py:221-224
 *       -> noise=KMSR_NOISE_GAUSSIAN with sigma already clamped by the caller
 *
 * Semantics (SURVEY.md section 8 a'):  p_h = kh/2, p_w = kw/2,
 *   AVGPOOL: s = floor(log2(factor)), F = 2^s, h = H>>s, w = W>>s,
 *            out[n,c,i,j] = mean_{dy,dx<F} blur[n,c,F*i+dy,F*j+dx]       (the cascade of s 2x2 avg-pools)
 *   STRIDE : F = factor, h = ceil(H/F), w = ceil(W/F), out[n,c,i,j] = blur[n,c,F*i,F*j]
 *   blur[n,c,y,x] = sum_{u,v} K[row(n),c,u,v] * hr[n,c,pad(y+u-p_h),pad(x+v-p_w)]
 *   REPLICATE: pad() clamps to the image; ZERO: out-of-image taps contribute 0.
 *   normalize=1: K[r,c] /= sum(K[r,c]) when that sum is > 0 (C_30:93-97).
 *   row(n) = kernel_idx[n] if kernel_idx else (n_kernels == 1 ? 0 : n).
 *   GAUSSIAN: out += sigma[row(n),c] * eps(g(n)*C+c, i, j) with the counter-based Philox stream
 *             documented in DESIGN.md ("noise stream"); (seed, offset) select the stream and
 *             g(n) = noise_sample_base + n * max(noise_sample_step, 1) is the GLOBAL index of local sample n,
 *             so a rank that holds a shard of a larger batch (contiguous block: base = first global sample, step 1;
 *             round-robin patch_ids[rank::world]: base = rank, step = world) draws exactly the noise a single
 *             process draws for the same global samples (results invariant to the world size).
 *   POOL    : out += noise[noise_idx ? noise_idx[n] : n, c, i, j]   (noise is [pool_size,C,h,w])
 * ========================================================================================== */
enum { KMSR_DOWN_AVGPOOL = 0, KMSR_DOWN_STRIDE = 1 };
enum { KMSR_PAD_REPLICATE = 0, KMSR_PAD_ZERO = 1 };
enum { KMSR_NOISE_NONE = 0, KMSR_NOISE_GAUSSIAN = 1, KMSR_NOISE_POOL = 2 };

#define KMSR_FLAG_FORCE_GENERIC 1u /* always use the generic (one thread per output) kernel        */
#define KMSR_FLAG_FORCE_FAST 2u    /* fail with KMSR_E_UNSUPPORTED instead of falling back to generic */
#define KMSR_FLAG_TAPS_READY 4u    /* workspace already holds the tap table of an earlier call with the
                                      same kernels / kh / kw / factor / down_mode / normalize: skip its rebuild */

typedef struct kmsr_degrade_desc {
    int32_t N, C, H, W;      /* HR batch [N,C,H,W], fp32, unit stride along W                      */
    int64_t hr_stride_n;     /* element strides of hr                                               */
    int64_t hr_stride_c;
    int64_t hr_stride_h;
    int32_t n_kernels;       /* kernels is [n_kernels,C,kh,kw] contiguous                            */
    int32_t kh, kw;          /* odd, <= 31                                                           */
    int32_t factor;          /* the caller's downscale_factor                                        */
    int32_t down_mode;       /* KMSR_DOWN_*                                                          */
    int32_t pad_mode;        /* KMSR_PAD_*                                                           */
    int32_t normalize;       /* 0/1                                                                  */
    int32_t noise_mode;      /* KMSR_NOISE_*                                                         */
    int32_t pool_size;       /* KMSR_NOISE_POOL: leading dim of `noise`                              */
    uint32_t flags;          /* KMSR_FLAG_*                                                          */
    uint64_t seed;           /* KMSR_NOISE_GAUSSIAN                                                  */
    uint32_t offset;         /* KMSR_NOISE_GAUSSIAN: stream offset (e.g. training step)             */
    int32_t rows_per_chunk;  /* tuning: LR rows per wave in the row-streaming kernel, 0 = auto      */
    uint32_t noise_sample_base; /* KMSR_NOISE_GAUSSIAN: global index of local sample 0 (0 for one process) */
    uint32_t noise_sample_step; /* KMSR_NOISE_GAUSSIAN: global index step between local samples, 0 == 1     */
} kmsr_degrade_desc;

/* LR size for a descriptor (pure host arithmetic). */
int kmsr_degrade_out_shape(const kmsr_degrade_desc* d, int32_t* h, int32_t* w);
/* Bytes of device workspace kmsr_degrade_f32 needs (the effective tap table and its row-reversed copy). */
size_t kmsr_degrade_workspace_bytes(const kmsr_degrade_desc* d);
/* 1 if this descriptor is served by the row-streaming fast kernel, 0 if by the generic kernel. */
int kmsr_degrade_uses_fast_path(const kmsr_degrade_desc* d, const void* hr);

int kmsr_degrade_f32(const kmsr_degrade_desc* d,
                     const float* hr,            /* [N,C,H,W] (strided)                              */
                     const float* kernels,       /* [n_kernels,C,kh,kw]                              */
                     const int32_t* kernel_idx,  /* [N] or NULL                                      */
                     const float* sigma,         /* [n_kernels,C]      (GAUSSIAN) else NULL          */
                     const float* noise,         /* [pool_size,C,h,w]  (POOL)     else NULL          */
                     const int64_t* noise_idx,   /* [N] or NULL        (POOL)                        */
                     float* lr,                  /* [N,C,h,w] contiguous                             */
                     void* workspace, size_t workspace_bytes, void* stream);

/* Backward of kmsr_degrade_f32 w.r.t. the blur kernels and noise sigmas, for the V-B configuration the reference
 * differentiates (ContentAdaptiveDegradation.forward muti_kernel/train_gemini.py:117-138 under loss_G.backward()
 * :219-222): down_mode STRIDE, normalize 0, pad ZERO or REPLICATE, factor <= 8, kh*kw <= 512.
 *   dkernels[n,c,u,v] = sum_{i,j} grad_lr[n,c,i,j] * x0[n,c, f*i+u-kh/2, f*j+v-kw/2]        [N,C,kh,kw]  (or NULL)
 *   dsigma[n,c]       = sum_{i,j} grad_lr[n,c,i,j] * eps[n,c,i,j]   (same (seed, offset) stream)  [N,C]  (or NULL)
 * Per-SAMPLE gradients: a caller that mixed / gathered a kernel bank (kernel_idx, einsum weights) finishes the chain
 * rule itself.  Deterministic (per-tile partial sums in the workspace, added in a fixed order).  The HR batch is read
 * once: HBM-bound like the forward.  The gradient w.r.t. the HR image is not provided (patches are data). */
size_t kmsr_degrade_bwd_workspace_bytes(const kmsr_degrade_desc* d);
int kmsr_degrade_bwd_f32(const kmsr_degrade_desc* d, const float* hr, const float* grad_lr /* [N,C,h,w] */,
                         float* dkernels, float* dsigma, void* workspace, size_t workspace_bytes, void* stream);

/* eps stream only: out[p,i,j] = eps(plane_base + p, i, j)  (tests; p < planes) */
int kmsr_gaussian_eps_f32(float* out, int32_t planes, int32_t h, int32_t w, uint64_t seed, uint32_t offset,
                          uint32_t plane_base, void* stream);

/* ============================================================================================
 * Noise-pool construction.  D_build_noise_pool.py:41-53,88: out[t] = (geo - den)[:, top:top+cs, left:left+cs]
 *   geo, den [C,H,W]; offs DEVICE int32 array [n][2] = (top,left) drawn by the caller with Python's `random`
 *   (D:49-50, inclusive bounds) so the integer stream stays bit-identical to the reference.
 * ========================================================================================== */
int kmsr_noise_crops_f32(const float* geo, const float* den, int32_t C, int32_t H, int32_t W,
                         const int32_t* offs, int32_t n, int32_t crop, float* out, void* stream);

/* ============================================================================================
 * 3x3 convolution, NCHW fp32, zero padding 1, stride 1 or 2 -- the SR network's conv stack
 * (head / residual body / upsampler / tail; the reference itself ships no SR model,
 * the module is build-defined and its oracle is the same module in stock torch ops) and the
 * reference's own 3x3 stacks: SelectorNet (muti_kernel/train_gemini.py:18-33, stride 2, BN folded by the
 * caller) and ConditionEncoder (muti_kernel/networks.py:31-38).
 *
 * Implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain):
 *   out[n,co,y,x] = epilogue( sum_{ci,u,v} w[co,ci,u,v] * x[n,ci,y*stride+u-1,x*stride+v-1] )
 *   epilogue(v)   = v + bias[co];  if relu: max(v,0);  if residual: v + residual[n,co,y,x]
 *   pixel_shuffle=1 (nn.PixelShuffle(2) fused into the store; Cout % 4 == 0):
 *       out[n, co/4, 2y + (co/2)%2, 2x + co%2] = epilogue(...)        out is [N,Cout/4,2Ho,2Wo]
 * Weights are consumed in a packed layout produced once per weight update by kmsr_conv3x3_pack_weights_f32
 * (a pure permutation + zero padding: [ceil(Cout/64)][ceil(Cin/8)][9][8][64]; for Cout <= 8, which is served by a
 * direct VALU kernel instead of MFMA tiles that would be mostly padding, [ceil(Cin/4)*4][9][Cout]).
 * ========================================================================================== */
typedef struct kmsr_conv3x3_desc {
    int32_t N, Cin, H, W;    /* input  [N,Cin,H,W] contiguous                                        */
    int32_t Cout;            /* output [N,Cout,Ho,Wo], Ho = (H-1)/stride+1, Wo = (W-1)/stride+1       */
    int32_t stride;          /* 1 or 2                                                               */
    int32_t relu;            /* 0/1                                                                  */
    int32_t pixel_shuffle;   /* 0/1 (stride must be 1)                                               */
    uint32_t flags;          /* KMSR_CONV_FLAG_* (0 for the plain forward)                           */
} kmsr_conv3x3_desc;
/* `residual` is not added but used as a ReLU mask: out = residual > 0 ? conv : 0 -- the data gradient through
 * y = relu(conv(x)) computed in the gradient convolution's own epilogue (mask source = the saved y).  Honoured by
 * kmsr_conv3x3_winov_fwd[_ex]_f32; the other entry points reject it with KMSR_E_UNSUPPORTED.               */
#define KMSR_CONV_FLAG_RELU_MASK 1u
/* kmsr_conv3x3_wgrad_f32 only: take the direct (36 products per tap set) kernel even where the Winograd-domain one applies */
#define KMSR_CONV_FLAG_WGRAD_DIRECT 2u
/* kmsr_conv3x3_wgrad_f32 only: skip the F(4x4,3x3)-domain kernel (round 4, the default where it applies) and take the
 * F(2x2,3x3)-domain one where that applies -- kept as the cross-check of the new kernel (tests) and for A/B timing */
#define KMSR_CONV_FLAG_WGRAD_WINO2 4u

/* number of floats in the packed weight buffer for (Cin, Cout) */
size_t kmsr_conv3x3_packed_weight_floats(int32_t Cin, int32_t Cout);
/* w [Cout,Cin,3,3] -> packed.  transpose_flip=1 packs the weights of the data-gradient convolution instead
 * (w'[ci,co,u,v] = w[co,ci,2-u,2-v]; then Cin/Cout of the *packed* conv are swapped by the caller).          */
int kmsr_conv3x3_pack_weights_f32(const float* w, int32_t Cin, int32_t Cout, int32_t transpose_flip, float* packed,
                                  void* stream);
int kmsr_conv3x3_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w,
                         const float* bias,      /* [Cout] or NULL                                  */
                         const float* residual,  /* out-shaped or NULL (not with pixel_shuffle)      */
                         float* out, void* stream);
/* Limits: stride 1 or 2; per image Cin * H * W < 2^31 elements and 8 * H * W * 4 < 2^32 bytes (round 4: input tiles are staged by buffer
 * loads whose 32-bit byte offsets span an 8-channel chunk; padding and channels past Cin are out-of-range offsets = zeros) -> otherwise
 * KMSR_E_UNSUPPORTED.  Stride 2 with W % 4 == 0 and a 16-byte aligned x takes the instance that stages 16-byte pieces; same results. */

/* L1 loss of the SR training step (build-defined, the reference has no SR model or loss): *loss = mean |a - b| over n floats and,
 * when grad != NULL, grad[i] = sign(a[i] - b[i]) / n -- the gradient w.r.t. a for an upstream gradient of 1 -- in ONE pass over a and b
 * (torch's l1_loss + autograd: five passes).  Fixed-order reduction (deterministic).  workspace: kmsr_l1_loss_workspace_floats() floats.
 * a, b, grad 16-byte aligned.  kmsr_scale_by_device_scalar_f32: x *= *scale in place unless *scale == 1 (the upstream gradient of
 * loss.backward() lives on the device; no host sync). */
size_t kmsr_l1_loss_workspace_floats(void);
int kmsr_l1_loss_f32(const float* a, const float* b, int64_t n, float* loss, float* grad, float* workspace, void* stream);
int kmsr_scale_by_device_scalar_f32(float* x, int64_t n, const float* scale, void* stream);

/* Backward of the fused PixelShuffle(2) store: g [N,C,2H,2W] (gradient in the shuffled layout) -> gu [N,4C,H,W] with
 * gu[n,4c+2i+j,y,x] = g[n,c,2y+i,2x+j], plus per-channel partial sums partials[N * ceil(H/32)][4C] (add them over the first
 * axis for the bias gradient sum_{n,y,x} gu; fixed order).  One pass instead of permute-copy + reduction.  W % 4 == 0. */
size_t kmsr_pixel_unshuffle2_sum_partials(int32_t N, int32_t C, int32_t H);    /* number of floats in `partials` */
int kmsr_pixel_unshuffle2_sum_f32(const float* g, int32_t N, int32_t C, int32_t H, int32_t W, float* gu, float* partials,
                                  void* stream);

/* 3x3 convolution with few INPUT channels (Cin <= 8: the SR head, bands -> F, and the data gradient of the tail), stride 1,
 * zero padding, bias / ReLU epilogue only (no residual, no pixel_shuffle, flags 0); W % 4 == 0, x 16-byte aligned.
 * Takes the UNPACKED weights w [Cout,Cin,3,3] (they live in registers for the whole persistent kernel; for a data gradient pass
 * w'[ci,co,u,v] = w[co,ci,2-u,2-v] made contiguous).  Same arithmetic as kmsr_conv3x3_fwd_f32 (fp32 MFMA, K ordered
 * (tap, channel pair)); it exists because these layers are output-write bound and the general kernel's 8-channel chunks and
 * per-tile weight reloads left them at a quarter of the HBM rate. */
int kmsr_conv3x3_smallci_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* w, const float* bias, float* out,
                                 void* stream);
/* The same kernel storing PIXEL-UNSHUFFLED (round 4): out is [N, 4 Cout, H/2, W/2] with out[n, 4 co + 2 (y & 1) + (x & 1), y >> 1, x >> 1] =
 * conv(x, w)[n, co, y, x] (+ bias, ReLU), and chan_sums[4 Cout] receives sum_{n,y,x} of every output channel (fixed order).  It is the data
 * gradient of the SR tail (bands -> F at HR size) delivered in the layout the F -> 4F + PixelShuffle(2) layer below takes as its output
 * gradient, together with that layer's bias gradient: replaces kmsr_pixel_unshuffle2_sum_f32's extra pass over the HR-sized map.
 * Cin <= 8, Cout <= 64, even H, W % 4 == 0.  workspace: kmsr_conv3x3_smallci_unshuffle_workspace_bytes(d).
 * chan_sums may be NULL (then workspace may be NULL too): store only -- the 64 per-lane sum registers push the kernel over its two-workgroups-
 * per-CU register budget (4.7 vs 2.5 ms at config 3), and a caller that knows the input's parity / border sums has the channel sums in closed
 * form (sum_p conv(x, w)[co, p] = sum_{ci,tap} w[co,ci,tap] * (sum of x[ci] over the pixels tap reaches): kmsr_amd/sr.py, UpTailFunction). */
size_t kmsr_conv3x3_smallci_unshuffle_workspace_bytes(const kmsr_conv3x3_desc* d);
int kmsr_conv3x3_smallci_unshuffle_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* w, const float* bias, float* out,
                                           float* chan_sums, void* workspace, size_t workspace_bytes, void* stream);

/* ============================================================================================
 * KernelGAN kernel composition (BASELINE config 5: batched per-patch kernel estimation).
 * Replaces the nested Python loops of
 *   DynamicMultiBandLinearGenerator.extract_effective_kernels   muti_kernel/networks.py:135-202
 *   MultiBandLinearGenerator.extract_effective_kernels          single_kernel/networks.py:108-157
 * (about 20 000 tiny F.conv2d calls per sample there) by ONE launch, one workgroup per (sample, band):
 *   K_0 = s_0 (.) W_0;  K_l[co] = sum_cm fullconv((s_l (.) W_l)[co,cm], K_{l-1}[cm])   (true 2-D convolution, :155-188)
 *   k = mean over the last layer's output maps; k = max(k,0); k /= (sum(k) if sum(k) > 1e-12 else 1)   (:189-194)
 * The chain has one input channel per band (ci[0] == 1), <= 8 layers, <= 64 channels, composed size <= 31.
 *   weights  [C][sum_l co_l*ci_l*ks_l^2]  one band after the other, layers concatenated in order
 *   scales   [B][C][sum_l co_l] per-output-channel scale factors (ConditionEncoder output) or NULL (= ones)
 *   co, ci, ks  HOST int32 arrays [n_layers]
 *   out      [B][C][K][K],  K = 1 + sum_l (ks_l - 1)
 * ========================================================================================== */
int kmsr_compose_kernels_f32(const float* weights, const float* scales, int32_t B, int32_t C, int32_t n_layers,
                             const int32_t* co, const int32_t* ci, const int32_t* ks, float* out, void* stream);

/* ============================================================================================
 * Backward of the 3x3 convolution (SR training step, SURVEY.md 8a S2 -- build-defined, stride 1).
 *   data gradient  : dx = conv3x3(dy, w')  with w'[ci,co,u,v] = w[co,ci,2-u,2-v]  -> kmsr_conv3x3_fwd_f32 on weights
 *                    packed with transpose_flip = 1 (Cin/Cout swapped in the descriptor).
 *   weight gradient: dw[co,ci,u,v] = sum_{n,y,x} dy[n,co,y,x] * x[n,ci,y+u-1,x+v-1]      (zero padded x)
 *                    fp32 MFMA, persistent workgroups accumulate slabs in registers, partial slabs are written to the
 *                    workspace and summed by a second kernel in a FIXED order (bit-reproducible, no float atomics).
 *                    W % 4 == 0, H >= 4 with Cout > 32 are computed in the Winograd F(4x4,3x3) domain (round 4):
 *                    dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G -- 36 products per (co, ci, 4x4 tile) instead of 144,
 *                    slabs of 36 x 64 x 32 (x / dy 16-byte aligned); with KMSR_CONV_FLAG_WGRAD_WINO2, even H, W % 4 == 0 and
 *                    Cout > 32 take the F(2x2,3x3) form (16 products per 2x2 tile instead of 36, slabs of 16 x 64 x 64);
 *                    other shapes (and d->flags & KMSR_CONV_FLAG_WGRAD_DIRECT) accumulate 64 x 64 x 9 slabs tap by tap.
 *                    Same result within fp32 rounding (tests: 2e-5 of max|dw| vs float64).
 *   bias gradient  : a plain reduction of dy; left to the caller (torch.sum).
 * d describes the FORWARD conv (N,Cin,H,W,Cout; stride must be 1; relu/pixel_shuffle ignored).
 * ========================================================================================== */
/* Named forms of two legs (SURVEY 8b's export list); both are kmsr_conv3x3_fwd_f32 with the descriptor rewritten:
 *   dgrad: d describes the FORWARD conv; packed_wT = kmsr_conv3x3_pack_weights_f32(w, Cin, Cout, transpose_flip = 1);
 *          dy [N,Cout,H,W] -> dx [N,Cin,H,W].  (The Winograd kernels serve the same purpose with their own packs: pack
 *          w' = flip(w).transpose(0,1) with kmsr_conv3x3_winov_pack_weights_f32 and call kmsr_conv3x3_winov_fwd[_ex]_f32.)
 *   pixelshuffle_fwd: d->pixel_shuffle forced to 1 (conv + PixelShuffle(2) in the store). */
int kmsr_conv3x3_dgrad_f32(const kmsr_conv3x3_desc* d, const float* dy, const float* packed_wT, float* dx, void* stream);
int kmsr_conv3x3_pixelshuffle_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                                      float* out, void* stream);
size_t kmsr_conv3x3_wgrad_workspace_bytes(const kmsr_conv3x3_desc* d);
int kmsr_conv3x3_wgrad_f32(const kmsr_conv3x3_desc* d, const float* x, const float* dy, float* dw /* [Cout,Cin,3,3] */,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ============================================================================================
 * Winograd F(2x2,3x3) variant of the stride-1 forward convolution (same semantics and epilogues as
 * kmsr_conv3x3_fwd_f32; exact fp32 arithmetic re-associated: Y = A^T[(G g G^T) .* (B^T d B)]A, 2.25x fewer MFMAs).
 * The 16 transform-domain products are 16 independent [Cout x Cin] x [Cin x tiles] GEMMs on the fp32 matrix cores;
 * a workgroup of 8 waves (one per CU) computes 64 channels x 64 2x2-tiles; the two waves of a SIMD split the 16
 * transform points and exchange partial output transforms through LDS.
 * Weights are consumed pre-transformed: packed layout [ceil(Cout/64)][ceil(Cin/16)][16][16][64] = G g G^T.
 * Intended for Cin >= 16 and Cout >= 32 (the SR body / upsampler convs); numerically within ~1e-6 of the direct kernel.
 * ========================================================================================== */
size_t kmsr_conv3x3_wino_packed_weight_floats(int32_t Cin, int32_t Cout);
int kmsr_conv3x3_wino_pack_weights_f32(const float* w /* [Cout,Cin,3,3] */, int32_t Cin, int32_t Cout, float* packed,
                                       void* stream);
int kmsr_conv3x3_wino_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                              const float* residual, float* out, void* stream);

/* Persistent "V in registers" Winograd kernel (the one the SR net runs; same math, epilogues and tolerances as above).
 * Needs W % 4 == 0 and 16-byte aligned x / packed_w (rows and weights arrive by 16-byte LDS-DMA pieces); every other stride-1
 * shape is served by kmsr_conv3x3_wino_fwd_f32.  One persistent 8-wave workgroup per CU walks a contiguous range of
 * (image, cout group, 4x64-pixel tile) items as ONE chunk stream (LDS-DMA rings for the pre-transformed weights and the raw input
 * rows, one barrier per 8-channel chunk).  MFMA tile v_mfma_f32_16x16x4_f32 with all 16 transform points of a (channel, tile) pair
 * in one lane: each lane computes its own B operand (the 4x4 input transform of its patch) from the raw rows in LDS, and the
 * output transform is lane-local -- the transformed inputs never go through LDS and the epilogue needs no exchange.
 * Packed layout [ceil(Cout/64)][ceil(Cin/8)][2 steps][4 channel blocks of 16][4][64 lanes][4].  Cout <= 1024 when bias is given
 * (the bias vector is staged in LDS). */
/* CROSS-CHECK entry points since round 3 (kmsr_conv3x3_winov_*): the F(2x2,3x3) kernel with the transformed inputs in registers.  No default
 * path takes it any more (the F(4x4,3x3) kernel below serves the same shapes); it stays exported because the tests compare the product kernel
 * with it on every shape and the host layer reaches it with algo = "winograd_vreg" / use_winograd4 = False. */
size_t kmsr_conv3x3_winov_packed_weight_floats(int32_t Cin, int32_t Cout);
int kmsr_conv3x3_winov_pack_weights_f32(const float* w /* [Cout,Cin,3,3] */, int32_t Cin, int32_t Cout, float* packed,
                                        void* stream);
int kmsr_conv3x3_winov_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                               const float* residual, float* out, void* stream);
/* Winograd F(4x4,3x3) kernel (the SR net's F -> F and F -> 4F layers; stride 1; this entry point: bias / ReLU / residual /
 * PixelShuffle epilogues, flags = 0).  36 multiplies per 4x4 output block and (ci, co) pair instead of 64 for F(2x2,3x3) and 144 direct:
 * 4x fewer MFMA FLOPs issued than the direct form.  Arithmetic stays fp32; the transforms with the points {0, +-1, +-2, inf}
 * re-associate the sums: <= 2e-5 of the output scale vs float64 in the tests (measured 5e-6 .. 8e-6 for a 64 -> 64 layer).
 * One persistent 4-wave workgroup per CU (one 512-register wave per SIMD) walks (image, cout group, 8x64-pixel tile) items as one
 * stream of 4-channel steps; each lane computes the 6x6 input transform of its own patch and the output transform of its own
 * 4x4 blocks; pre-transformed weights are loaded straight from L2 into the MFMA operand registers, raw rows arrive by buffer
 * LDS-DMA (zero padding = out-of-range offsets).  Needs W % 4 == 0, 16-byte aligned x / packed_w / out / residual,
 * Cin*H*W*4 < 2 GB per image, Cout <= 1024 (LDS bias table; with or without a bias pointer).  Packed layout [ceil(Cout/64)][ceil(Cin/4)][2 halves][9 groups][2 blocks of 16][64 lanes][4]. */
size_t kmsr_conv3x3_wino4_packed_weight_floats(int32_t Cin, int32_t Cout);
int kmsr_conv3x3_wino4_pack_weights_f32(const float* w /* [Cout,Cin,3,3] */, int32_t Cin, int32_t Cout, float* packed,
                                        void* stream);
/* transpose_flip = 1 (round 4): pack the DATA-GRADIENT convolution of the forward layer w [Cout,Cin,3,3] (w'[ci,co,u,v] = w[co,ci,2-u,2-v]; the
 * packed convolution runs Cout -> Cin channels: size the buffer with kmsr_conv3x3_wino4_packed_weight_floats(Cout, Cin)) in the pack launch
 * itself -- same convention as kmsr_conv3x3_pack_weights_f32.  transpose_flip = 0 is kmsr_conv3x3_wino4_pack_weights_f32. */
int kmsr_conv3x3_wino4_pack_weights_ex_f32(const float* w, int32_t Cin, int32_t Cout, int32_t transpose_flip, float* packed, void* stream);
int kmsr_conv3x3_wino4_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                               const float* residual, float* out, void* stream);
/* The same kernel with the F(2x2) kernel's other epilogues: d->pixel_shuffle (fused PixelShuffle(2), no residual), and the training
 * step's extras exactly as kmsr_conv3x3_winov_fwd_ex_f32 defines them: KMSR_CONV_FLAG_RELU_MASK (out = residual > 0 ? conv : 0) and,
 * when chan_sum != NULL, per-channel partial sums of the stored tensor: chan_sum[item][2 tile rows][64] with
 * item = (n * ceil(Cout/64) + cg) * tiles + tile (kmsr_conv3x3_wino4_chan_sum_floats(d) floats, every entry written, padding
 * channels 0); summing over (n, tile, tile row) gives sum_{n,y,x} out[n,c,y,x].  Fixed order: run-to-run identical. */
size_t kmsr_conv3x3_wino4_chan_sum_floats(const kmsr_conv3x3_desc* d);
int kmsr_conv3x3_wino4_fwd_ex_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                                  const float* residual, float* out, float* chan_sum, void* stream);
/* Training-step variant: honours KMSR_CONV_FLAG_RELU_MASK and, when chan_sum != NULL (not with pixel_shuffle), also writes
 * per-channel partial sums of the tensor it stores: chan_sum[item][4][64] with item = (n * ceil(Cout/64) + cg) * tiles + tile
 * (kmsr_conv3x3_winov_chan_sum_floats(d) floats, every entry written, padding channels 0).  Summing over (n, tile, 4) gives
 * sum_{n,y,x} out[n,c,y,x] -- the bias gradient of the layer whose output gradient this launch produces -- without re-reading
 * the tensor; the partials are formed in a fixed order (run-to-run identical). */
size_t kmsr_conv3x3_winov_chan_sum_floats(const kmsr_conv3x3_desc* d);
int kmsr_conv3x3_winov_fwd_ex_f32(const kmsr_conv3x3_desc* d, const float* x, const float* packed_w, const float* bias,
                                  const float* residual, float* out, float* chan_sum, void* stream);

/* ============================================================================================
 * 5x5 convolution (zero padding 2) with PixelShuffle(4) fused into the store -- the COLLAPSED SR upsampler.
 * up1 (3x3 F->4F) -> PixelShuffle(2) -> up2 (3x3 F->4F) -> PixelShuffle(2) -> tail (3x3 F->bands) contains no
 * non-linearity, so away from the image border it equals one 5x5 convolution F -> 16*bands on the LR grid followed
 * by PixelShuffle(4): ~10x fewer flops and no HR-sized feature maps.  The caller composes the 5x5 weights/bias from
 * the three layers and recomputes the outer 4-pixel HR frame (up2's and tail's own zero padding reach 3 pixels) with
 * the layer-by-layer kernels (kmsr_amd/sr.py: SRNet.fuse_upsampler).
 *   out[n, co>>4, 4y + ((co>>2)&3), 4x + (co&3)] = bias[co] + sum_{ci,u,v} w[co,ci,u,v] * x[n,ci,y+u-2,x+v-2]
 *   x [N,Cin,H,W], w [Cout,Cin,5,5] with Cout % 16 == 0, out [N,Cout/16,4H,4W].  fp32 MFMA implicit GEMM.
 * ========================================================================================== */
size_t kmsr_conv5x5_ps4_packed_weight_floats(int32_t Cin, int32_t Cout);
int kmsr_conv5x5_ps4_pack_weights_f32(const float* w /* [Cout,Cin,5,5] */, int32_t Cin, int32_t Cout, float* packed,
                                      void* stream);
int kmsr_conv5x5_ps4_fwd_f32(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, const float* x,
                             const float* packed_w, const float* bias, float* out, void* stream);

/* bf16 form of the collapsed x4 upsampler for the channels-last bf16 path (BASELINE config 4): x [N,H,W,Cin_p] bf16 (Cin_p % 16 == 0),
 * w the composed 5x5 weights [Cout,Cin,5,5] fp32 (rounded to bf16 by the pack kernel), bias fp32 [Cout] or NULL, Cout = 16 * bands;
 * out bf16 PLANES [N, bands, 4H, 4W] with out[n, co>>4, 4y+((co>>2)&3), 4x+(co&3)] = bias[co] + conv5x5(x, w)[co] (zero pad 2, fp32
 * accumulation on v_mfma_f32_32x32x16_bf16, one rounding to bf16 at the store).  Replaces three bf16 launches (up1 / up2 / tail) and
 * their two HR-sized intermediates; the outer HR frame, where the intermediate layers' zero padding differs from the composed
 * operator's, is recomputed by the caller layer by layer on four border strips (kmsr_amd/sr_bf16.py).
 * Packed layout [ceil(Cout/64)][Cin_p/16][25 taps][64][16]. */
size_t kmsr_conv5x5_ps4_bf16_packed_weight_elems(int32_t Cin_p, int32_t Cout);
int kmsr_conv5x5_ps4_bf16_pack_weights(const float* w /* [Cout,Cin,5,5] */, int32_t Cin, int32_t Cout, int32_t Cin_p, void* packed,
                                       void* stream);
int kmsr_conv5x5_ps4_fwd_bf16(int32_t N, int32_t Cin_p, int32_t H, int32_t W, int32_t Cout, const void* x, const void* packed_w,
                              const float* bias, void* out, void* stream);

/* Outer frame of the collapsed upsampler: on the outermost LR row / column the chain's intermediate zero padding turns the
 * interior 5x5 map into 8 border-class maps (0 top, 1 bottom, 2 left, 3 right -- corners excluded --, 4..7 the corners tl, tr,
 * bl, br).  wcls [8][Cin][25][16*bands] (tap = dy*5+dx, out channel = band*16 + 4a + b), bcls [8][16*bands]; for every border LR
 * pixel (i,j) of every image:  y[n, band, 4i+a, 4j+b] = bcls[c] + sum wcls[c][ci][dy][dx] * h[n, ci, i+dy-2, j+dx-2]  (zero
 * outside the image) is written into y [N,bands,4H,4W] (the interior launch's output).  H, W >= 4, 1..8 bands; bcls and y 16-byte
 * aligned (KMSR_E_BADARG otherwise); min(N, 64) images of h below 2 GB (32-bit buffer offsets; KMSR_E_UNSUPPORTED otherwise --
 * callers then run the three layers one by one).  Any W; W % 4 == 0 with a 16-byte aligned h takes the 16-byte patch-row loads. */
int kmsr_upsampler_frame_f32(const float* h, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t bands, const float* wcls,
                             const float* bcls, float* y, void* stream);

/* ============================================================================================
 * Small-channel convolution with REFLECT padding and per-unit weights -- the KernelGAN generators' forward
 * (DynamicMultiBandLinearGenerator._apply_conv_chain  muti_kernel/networks.py:101-117,
 *  MultiBandLinearGenerator.forward                    single_kernel/networks.py:83-106):
 * every (sample, band) "unit" runs its own bias-free chain of k x k convs with F.pad(mode='reflect') per layer.
 *   out[u,co,y,x] = sum_{ci,a,b} w[u,ci,a*k+b,co] * x[u,ci,refl(y+a-k/2),refl(x+b-k/2)],  refl(-i) = i, refl(H-1+i) = H-1-i
 *   x [U,Cin,H,W], w [U][Cin][k*k][Cout] (already scaled / folded by the caller), out [U,Cout,H,W]
 *   k in {1,3,5,7}, Cin <= 64, Cout in {1, 5, 32, 64}; H, W > k/2.
 * ========================================================================================== */
int kmsr_conv_reflect_f32(const float* x, const float* w, int32_t U, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                          int32_t k, float* out, void* stream);

/* ============================================================================================
 * The same small-channel k x k convolution as a family with gradient legs -- what the KernelGAN trainers
 * (muti_kernel/train.py:234-323, single_kernel/train.py:250-347) differentiate through: the generators' reflect-padded
 * chains (per-unit weights) and the PatchDiscriminator's zero-padded 7x7 / 1x1 convs (muti_kernel/networks.py:250-268,
 * one shared weight set, bias).
 *   fwd  : out[u,co,y,x] = bias[co] + sum_{ci,a,b} w[u|0,ci,a*k+b,co] * xpad[u,ci,y+a-k/2,x+b-k/2]
 *          (xpad = reflect or zero extension of x by k/2); bias may be NULL.
 *   dgrad: dx = (d out / d x)^T dy.  dy [U,Cout,H,W]; wT [U|1][Cout][k*k][Cin] = w with its channel axes exchanged
 *          (taps NOT flipped -- the kernel walks them backwards); dx [U,Cin,H,W].  Reflect padding folds the gradient
 *          of every mirrored position back onto the interior pixel it copies.
 *   wgrad: dw[u|0,ci,a*k+b,co] = sum_{(u,)y,x} dy[u,co,y,x] * xpad[u,ci,y+a-k/2,x+b-k/2]  (summed over units when the
 *          weights are shared); fixed-order reduction, run-to-run identical.  workspace from
 *          kmsr_convk_wgrad_workspace_bytes(d) (0 = unsupported configuration).
 *   k odd in 1..7, Cin, Cout <= 64; instantiated: fwd Cout in {1,5,32,64}, dgrad Cin in {1,5,32,64}, wgrad
 *   Cin*k*k*Cout <= 2048 * (16 if Cout%16==0, 4 if Cout%4==0, else 1).  H, W > k/2.
 * ========================================================================================== */
enum { KMSR_CONVK_REFLECT = 0, KMSR_CONVK_ZERO = 1 };
typedef struct kmsr_convk_desc {
    int32_t U, Cin, Cout, H, W, k;
    int32_t pad_mode;        /* KMSR_CONVK_*                                                     */
    int32_t shared_weights;  /* 0: one weight set per unit; 1: one weight set for all units      */
} kmsr_convk_desc;
int kmsr_convk_fwd_f32(const kmsr_convk_desc* d, const float* x, const float* w, const float* bias, float* out, void* stream);
int kmsr_convk_dgrad_f32(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, void* stream);
/* The same data gradient with scratch memory: the reflect-padded 32-channel layers then run as a zero-padded pass on the padded
 * domain plus a fold of the mirrored ring (same sums, different order; about 2x faster).  kmsr_convk_dgrad_workspace_bytes(d) = 0
 * means the configuration has no such path (workspace may then be NULL). */
size_t kmsr_convk_dgrad_workspace_bytes(const kmsr_convk_desc* d);
int kmsr_convk_dgrad_ws_f32(const kmsr_convk_desc* d, const float* dy, const float* wT, float* dx, void* workspace,
                            size_t workspace_bytes, void* stream);
size_t kmsr_convk_wgrad_workspace_bytes(const kmsr_convk_desc* d);
int kmsr_convk_wgrad_f32(const kmsr_convk_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ============================================================================================
 * bf16 3x3 convolution (BASELINE config 4: 7-band x4 SR on 1024x1024 LR tiles -- the HBM-bound large-tile path).
 * Activations are CHANNELS-LAST bf16 (element (n,c,y,x) at ((n*H + y)*W + x)*Cp + c, Cp = channel count padded to a
 * multiple of 16, pad channels zero); products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 bias.
 * Same epilogue semantics as kmsr_conv3x3_fwd_f32 (bias, ReLU, residual, fused PixelShuffle(2)), stride 1.
 * Persistent workgroups keep the whole layer's packed weights in LDS and stream 8x32-pixel tiles.
 *   d->Cin / d->Cout are the PADDED channel counts of the input / conv output (multiples of 16; Cout % 64 == 0 with
 *   pixel_shuffle).  out is [N,H,W,Cout] (or [N,2H,2W,Cout/4] with pixel_shuffle), residual like out.
 *   packed weights: bf16 [ceil(Cout/64)][9][64][Cin] (zero padded), from kmsr_conv3x3_bf16_pack_weights.
 * ========================================================================================== */
size_t kmsr_conv3x3_bf16_packed_weight_elems(int32_t Cin_p, int32_t Cout_p);
/* w fp32 [Cout,Cin,3,3] (unpadded sizes) -> packed bf16 for padded sizes Cin_p, Cout_p */
int kmsr_conv3x3_bf16_pack_weights(const float* w, int32_t Cin, int32_t Cout, int32_t Cin_p, int32_t Cout_p,
                                   void* packed_bf16, void* stream);
int kmsr_conv3x3_fwd_bf16(const kmsr_conv3x3_desc* d, const void* x_bf16, const void* packed_w_bf16,
                          const float* bias /* [Cout] fp32 (padded) or NULL */, const void* residual_bf16,
                          void* out_bf16, void* stream);
/* Fused residual block of the bf16 path (round 4): out = x + conv2(relu(conv1(x) + bias1)) + bias2 on channels-last bf16 tensors with
 * exactly 64 channels ([N,H,W,64]); packed_w1 / packed_w2 from kmsr_conv3x3_bf16_pack_weights(w, 64, 64, 64, 64), biases fp32 [64] or
 * NULL.  ONE launch instead of two kmsr_conv3x3_fwd_bf16 calls: the intermediate map stays in LDS (conv1 is recomputed on the 1-pixel
 * halo ring of every 8 x 16 tile), so a block reads x once and writes out once.  Every rounding of the two-launch path is kept
 * (intermediate rounded to bf16, zero outside the image; conv2 + bias rounded to bf16 before the skip add): the result is
 * bit-identical to it.  out must not alias x; 16-byte aligned pointers; H*W*128 B < 4 GiB per image. */
int kmsr_resblock_fwd_bf16(int32_t N, int32_t H, int32_t W, const void* x_bf16, const void* packed_w1_bf16, const float* bias1,
                           const void* packed_w2_bf16, const float* bias2, void* out_bf16, void* stream);
int kmsr_conv3x3_pixelshuffle_fwd_bf16(const kmsr_conv3x3_desc* d, const void* x_bf16, const void* packed_w_bf16, const float* bias,
                                       void* out_bf16, void* stream);   /* kmsr_conv3x3_fwd_bf16 with pixel_shuffle = 1 */
/* Backward legs on the same channels-last bf16 tensors; d describes the FORWARD conv (padded channel counts).
 *   dgrad: dx[N,H,W,Cin] = conv3x3(dy[N,H,W,Cout], w') with w'[ci,co,u,v] = w[co,ci,2-u,2-v] -- the forward kernel on weights
 *          packed by kmsr_conv3x3_bf16_pack_weights_dgrad (same fp32 master weights; needs padded Cout <= 64, i.e. not the
 *          F->4F layers, and no pixel_shuffle: un-shuffle the gradient first).
 *   wgrad: dw fp32 [Cout][Cin][3][3] (padded counts; rows / columns of padding channels are zero)
 *          = sum_{n,y,x} dy * x.  Every MFMA shape needs the reduction index (pixels) contiguous per lane, which channels-last
 *          bf16 is not, so the two operands are first expanded to fp32 planes in the workspace (exact) and the sum runs on the
 *          fp32 weight-gradient kernels (Winograd domain where eligible): exact products of the bf16 values, fp32 accumulation,
 *          fixed order. */
int kmsr_conv3x3_bf16_pack_weights_dgrad(const float* w, int32_t Cin, int32_t Cout, int32_t Cin_p, int32_t Cout_p,
                                         void* packed_bf16, void* stream);
int kmsr_conv3x3_dgrad_bf16(const kmsr_conv3x3_desc* d, const void* dy_bf16, const void* packed_wT_bf16, void* dx_bf16,
                            void* stream);
size_t kmsr_conv3x3_wgrad_bf16_workspace_bytes(const kmsr_conv3x3_desc* d);
int kmsr_conv3x3_wgrad_bf16(const kmsr_conv3x3_desc* d, const void* x_bf16, const void* dy_bf16, float* dw, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ============================================================================================
 * Patch cutting with a water mask -- the step that FEEDS the path.
 * Replaces apply_water_mask / create_patches_nc   kernel_from_lr_gan/A_00_patch_cutter_universal.py:89-197
 *   water mask : v == invalid_value -> NaN (every band);  water = tmin <= NIR <= tmax (NaN compares false);
 *                masked[c] = water ? v[c] : NaN;   counts[0] = #(NIR not NaN), counts[1] = #water   (:102-121)
 *   patch grid : stride = int(ps * stride_ratio) (host), hp = (H-ps)/stride+1, wp = (W-ps)/stride+1 (:149-153);
 *                nan_counts[i*wp+j] = number of NaN in data[:, i*stride:+ps, j*stride:+ps]  (:171) -- an INTEGER, so the
 *                caller's `count / patch.size > nan_threshold` test reproduces the reference's decision exactly
 *   gather     : out[t] = data[:, tl[t][0]:+ps, tl[t][1]:+ps]
 * All counts are exact integer reductions (integer atomics only).
 * ========================================================================================== */
int kmsr_water_mask_f32(const float* data, int32_t C, int32_t H, int32_t W, int32_t nir_index, float invalid_value,
                        float tmin, float tmax, float* masked, uint64_t* counts /* device [2], zeroed by the call */,
                        void* stream);
int kmsr_patch_nan_counts_f32(const float* data, int32_t C, int32_t H, int32_t W, int32_t ps, int32_t stride,
                              int32_t* nan_counts /* device [hp*wp] */, void* stream);
int kmsr_gather_patches_f32(const float* data, int32_t C, int32_t H, int32_t W, int32_t ps,
                            const int32_t* top_left /* device [n][2] */, int32_t n, float* out /* [n,C,ps,ps] */,
                            void* stream);

/* ============================================================================================
 * Non-local-means denoise of one band (the denoise/ stage).  PARITY UNPINNED: the reference
 * calls scikit-image `denoise_nl_means(fast_mode=True, patch_size=7, patch_distance=11)` (denoise/denoise.py:55-62),
 * which is neither vendored, versioned nor installable here; this entry point implements the published fast-mode
 * algorithm as restated in oracle/nlm.py (the only checker available):
 *   v = reflect-padded image; for every shift t in [-d,d]^2:
 *   D_t(p) = sum over the box rows (r-3, r+3], cols (c-3, c+3] of (v[x] - v[x+t])^2 - 2 sigma^2;  dist = max(D,0) / (h^2 * 49)
 *   out[p] = sum_t w v[p+t] / sum_t w,   w = exp(-dist) if dist <= 5 else 0.
 *   img [B,H,W] (NaN-free), h [B], sigma [B] device arrays (one denoising strength per band); min(H,W) > d + 4.
 * ========================================================================================== */
int kmsr_nlm_f32(const float* img, int32_t B, int32_t H, int32_t W, int32_t patch_size, int32_t patch_distance,
                 const float* h, const float* sigma, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KMSR_H */
