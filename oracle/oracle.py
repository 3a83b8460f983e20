"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see kmsr_oracle.c).

numpy in / numpy out.  Function names follow the reference's own names so that parity tests
read like tests of the reference.  Citations: paths relative to the upstream repository.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkmsr_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)


def build(force=False):
    """Compile oracle/libkmsr_oracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "kmsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libkmsr_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_set_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(f32p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def set_threads(n):
    return lib().oracle_set_threads(int(n))


# ----------------------------------------------------------------------------------------------
# V-A  C_30apply_kernel_to_landsat.py:68-124 / C_31apply_muti_kernel_to_landsat.py:59-97
# ----------------------------------------------------------------------------------------------
def apply_kernel_degradation(img, kernel, downscale_factor=8):
    img = _f32(img)
    kernel = _f32(kernel)
    Cc, H, W = img.shape
    if kernel.ndim == 2:                       # C_30:83-85 broadcast
        kernel = np.ascontiguousarray(np.broadcast_to(kernel, (Cc,) + kernel.shape))
    elif kernel.ndim == 3:
        assert kernel.shape[0] == Cc           # C_30:88
    else:
        raise ValueError("unsupported kernel rank")   # C_31:69
    s = int(np.log2(downscale_factor))
    h, w = H, W
    for _ in range(s):
        h, w = h // 2, w // 2
    out = np.empty((Cc, h, w), np.float32)
    _check(lib().oracle_apply_kernel_degradation(_p(img), Cc, H, W, _p(kernel), kernel.shape[1], kernel.shape[2],
                                                 int(downscale_factor), _p(out)), "apply_kernel_degradation")
    return out


def normalize_kernel(kernel):
    kernel = _f32(kernel)
    out = np.empty_like(kernel)
    lib().oracle_normalize_kernel(_p(kernel), kernel.shape[0], kernel.shape[1], kernel.shape[2], _p(out))
    return out


# ----------------------------------------------------------------------------------------------
# V-B  muti_kernel/train_gemini.py:124-138
# ----------------------------------------------------------------------------------------------
def vb_blur_stride(x, kernels, f=4):
    """x [B,C,H,W]; kernels [B,C,kh,kw] (per sample, per band); zero pad, cross-correlation, [::f, ::f]."""
    x = _f32(x)
    kernels = _f32(kernels)
    B, Cc, H, W = x.shape
    assert kernels.shape[:2] == (B, Cc)
    h, w = -(-H // f), -(-W // f)
    out = np.empty((B, Cc, h, w), np.float32)
    _check(lib().oracle_vb_blur_stride(_p(x), B, Cc, H, W, _p(kernels), kernels.shape[2], kernels.shape[3], int(f),
                                       _p(out)), "vb_blur_stride")
    return out


def vb_backward(x, g, ksize=13, f=4, pad="zero"):
    """d/dK of vb_blur_stride (train_gemini.py:117-134 under autograd): per (sample, band) plane
    dK[u,v] = sum_{i,j} g[i,j] * x0[f*i+u-p, f*j+v-p]  -- float64 accumulation, returned as float64 [N,C,k,k].
    Test infrastructure only."""
    x = np.asarray(x, np.float64)
    g = np.asarray(g, np.float64)
    N, C, H, W = x.shape
    p = ksize // 2
    h, w = -(-H // f), -(-W // f)
    assert g.shape == (N, C, h, w)
    ext = (h - 1) * f + ksize, (w - 1) * f + ksize            # padded extent the taps can reach
    xp = np.pad(x, ((0, 0), (0, 0), (p, max(ext[0] - H - p, 0)), (p, max(ext[1] - W - p, 0))),
                mode="constant" if pad == "zero" else "edge")
    dk = np.empty((N, C, ksize, ksize), np.float64)
    for u in range(ksize):
        for v in range(ksize):
            dk[:, :, u, v] = (g * xp[:, :, u:u + f * h:f, v:v + f * w:f]).sum(axis=(2, 3))
    return dk


def vb_backward_sigma(g, seed, offset=0, plane_base=0):
    """d/dsigma of add_gaussian_noise: sum_{i,j} g * eps with the same counter-based stream -> float64 [N,C]."""
    g = np.asarray(g, np.float64)
    N, C, h, w = g.shape
    eps = gaussian_eps(N * C, h, w, seed, offset, plane_base).reshape(N, C, h, w).astype(np.float64)
    return (g * eps).sum(axis=(2, 3))


def philox_u32(seed, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox_u32(C.c_uint64(seed), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), out)
    return np.array(list(out), np.uint32)


def add_gaussian_noise(x, sigma, seed, offset=0, plane_base=0):
    """x [...,h,w] with prod(leading dims) planes; sigma one value per plane.  Returns x + sigma*eps, eps addressed by the
    GLOBAL plane index plane_base + p (plane_base = first global sample * C for a shard of a larger batch)."""
    x = _f32(x).copy()
    h, w = x.shape[-2:]
    planes = x.size // (h * w)
    sigma = _f32(np.broadcast_to(np.asarray(sigma, np.float32).reshape(-1), (planes,)))
    _check(lib().oracle_add_gaussian_noise(_p(x), planes, h, w, _p(sigma), C.c_uint64(seed), C.c_uint32(offset),
                                            C.c_uint32(plane_base)), "add_gaussian_noise")
    return x


def gaussian_eps(planes, h, w, seed, offset=0, plane_base=0):
    return add_gaussian_noise(np.zeros((planes, h, w), np.float32), np.ones(planes, np.float32), seed, offset, plane_base)


# ----------------------------------------------------------------------------------------------
# V-C  E_make_train_data.py:65-74, D_build_noise_pool.py:41-53,88
# ----------------------------------------------------------------------------------------------
def add_pool_noise(blurred, pool, idx):
    blurred = _f32(blurred)
    pool = _f32(pool)
    idx = np.ascontiguousarray(idx, np.int64)
    N, Cc, h, w = blurred.shape
    out = np.empty_like(blurred)
    _check(lib().oracle_add_pool_noise(_p(out), _p(blurred), N, Cc, h, w, _p(pool), pool.shape[0],
                                       idx.ctypes.data_as(C.POINTER(C.c_int64))), "add_pool_noise")
    return out


def noise_crop(geo, den, top, left, cs):
    geo = _f32(geo)
    den = _f32(den)
    Cc, H, W = geo.shape
    out = np.empty((Cc, cs, cs), np.float32)
    _check(lib().oracle_noise_crop(_p(geo), _p(den), Cc, H, W, int(top), int(left), int(cs), _p(out)), "noise_crop")
    return out


# ----------------------------------------------------------------------------------------------
# dense conv / pixel shuffle
# ----------------------------------------------------------------------------------------------
def conv2d(x, weight, bias=None, stride=1, pad=0, relu=False):
    x = _f32(x)
    weight = _f32(weight)
    N, Ci, H, W = x.shape
    Co, Ci2, kh, kw = weight.shape
    assert Ci == Ci2
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = np.empty((N, Co, Ho, Wo), np.float32)
    b = _f32(bias) if bias is not None else None
    _check(lib().oracle_conv2d(_p(x), N, Ci, H, W, _p(weight), _p(b) if b is not None else None, Co, kh, kw,
                               int(stride), int(pad), int(bool(relu)), _p(out)), "conv2d")
    return out


def pixel_shuffle(x, r):
    x = _f32(x)
    N, Cin, H, W = x.shape
    out = np.empty((N, Cin // (r * r), H * r, W * r), np.float32)
    _check(lib().oracle_pixel_shuffle(_p(x), N, Cin, H, W, int(r), _p(out)), "pixel_shuffle")
    return out


# ----------------------------------------------------------------------------------------------
# SelectorNet  muti_kernel/train_gemini.py:14-39 (eval-mode BatchNorm, running stats from moe_model.pth)
# ----------------------------------------------------------------------------------------------
def _bn_eval(x, w, b, mean, var, eps=1e-5):
    inv = (w / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return (x - mean[None, :, None, None]) * inv[None, :, None, None] + b[None, :, None, None]


def selector_logits(sd, x):
    """sd: dict of numpy arrays with the reference's state_dict keys (selector.*)."""
    h = _f32(x)
    for conv, bn in ((0, 1), (3, 4), (6, 7)):
        h = conv2d(h, sd[f"selector.features.{conv}.weight"], sd[f"selector.features.{conv}.bias"], stride=2, pad=1)
        h = _bn_eval(h, sd[f"selector.features.{bn}.weight"], sd[f"selector.features.{bn}.bias"],
                     sd[f"selector.features.{bn}.running_mean"], sd[f"selector.features.{bn}.running_var"])
        h = np.maximum(h, 0).astype(np.float32)
    feat = h.mean(axis=(2, 3), dtype=np.float32)                 # AdaptiveAvgPool2d(1)
    return feat @ sd["selector.classifier.weight"].T + sd["selector.classifier.bias"]


def effective_kernels(kernel_bank):
    """train_gemini.py:68-80: softmax over the 169 taps of each [n,c] kernel."""
    kb = np.asarray(kernel_bank, np.float32)
    flat = kb.reshape(kb.shape[0], kb.shape[1], -1)
    e = np.exp(flat - flat.max(axis=-1, keepdims=True))
    return (e / e.sum(axis=-1, keepdims=True)).reshape(kb.shape).astype(np.float32)


def effective_sigmas(sigma_bank):
    """train_gemini.py:82-86: softplus."""
    s = np.asarray(sigma_bank, np.float64)
    return np.where(s > 20, s, np.log1p(np.exp(np.minimum(s, 20)))).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# KernelGAN  muti_kernel/networks.py:10-202, single_kernel/networks.py:83-157
# ----------------------------------------------------------------------------------------------
KS = (7, 5, 3, 1, 1, 1)
LAYER_OUT = (32, 32, 32, 32, 32, 1)


def _ptr_array(arrs):
    arr_t = f32p * len(arrs)
    return arr_t(*[(_p(a) if a is not None else C.cast(None, f32p)) for a in arrs])


def _int_array(v):
    return (C.c_int * len(v))(*[int(t) for t in v])


def compose_kernel(weights, scales=None):
    """weights: list of [co,ci,k,k]; scales: list of [co] or None.  -> [K,K] (one band, one sample)."""
    weights = [_f32(w) for w in weights]
    scales = [(_f32(s) if s is not None else None) for s in (scales or [None] * len(weights))]
    co = [w.shape[0] for w in weights]
    ci = [w.shape[1] for w in weights]
    ks = [w.shape[2] for w in weights]
    K = 1 + sum(k - 1 for k in ks)
    out = np.empty((K, K), np.float32)
    _check(lib().oracle_compose_kernel(len(weights), _ptr_array(weights), _ptr_array(scales), _int_array(co),
                                       _int_array(ci), _int_array(ks), _p(out)), "compose_kernel")
    return out


def chain_forward(x2d, weights, scales=None, n_pools=3):
    """x2d [H,W] (one band of one sample) -> [H/8, W/8]"""
    x2d = _f32(x2d)
    H, W = x2d.shape
    weights = [_f32(w) for w in weights]
    scales = [(_f32(s) if s is not None else None) for s in (scales or [None] * len(weights))]
    co = [w.shape[0] for w in weights]
    ci = [w.shape[1] for w in weights]
    ks = [w.shape[2] for w in weights]
    out = np.empty((co[-1], H >> n_pools, W >> n_pools), np.float32)
    _check(lib().oracle_chain_forward(_p(x2d), H, W, len(weights), _ptr_array(weights), _ptr_array(scales),
                                      _int_array(co), _int_array(ci), _int_array(ks), int(n_pools), _p(out)),
           "chain_forward")
    return out[0]


def condition_scales(sd, x, prefix="generator.condition_encoder.", scale_gain=0.1, in_ch=5):
    """ConditionEncoder.forward, muti_kernel/networks.py:41-58.  Returns [B, in_ch, 161]
    (per band the 6 layers' scales concatenated: 32,32,32,32,32,1)."""
    h = conv2d(x, sd[prefix + "encoder.0.weight"], sd[prefix + "encoder.0.bias"], stride=1, pad=1, relu=True)
    h = conv2d(h, sd[prefix + "encoder.2.weight"], sd[prefix + "encoder.2.bias"], stride=2, pad=1, relu=True)
    h = conv2d(h, sd[prefix + "encoder.4.weight"], sd[prefix + "encoder.4.bias"], stride=2, pad=1, relu=True)
    feat = h.mean(axis=(2, 3), dtype=np.float32)
    raw = feat @ sd[prefix + "fc.weight"].T + sd[prefix + "fc.bias"]
    s = (1.0 + scale_gain * np.tanh(raw)).astype(np.float32)
    return s.reshape(x.shape[0], in_ch, sum(LAYER_OUT))


def _split_scales(s161):
    out, st = [], 0
    for oc in LAYER_OUT:
        out.append(np.ascontiguousarray(s161[st:st + oc]))
        st += oc
    return out


def muti_band_weights(sd, band, prefix="generator."):
    return [sd[f"{prefix}weight_bands.{band}.{l}"] for l in range(6)]


def muti_extract_effective_kernels(sd, x=None, reduce_batch=True, in_ch=5):
    """DynamicMultiBandLinearGenerator.extract_effective_kernels, muti_kernel/networks.py:135-202."""
    if x is None:
        B = 1
        scales = np.ones((1, in_ch, sum(LAYER_OUT)), np.float32)
    else:
        B = x.shape[0]
        scales = condition_scales(sd, x, in_ch=in_ch)
    ks = np.stack([np.stack([compose_kernel(muti_band_weights(sd, c), _split_scales(scales[b, c]))
                             for c in range(in_ch)]) for b in range(B)])
    return ks.mean(axis=0) if reduce_batch else ks


def muti_generator_forward(sd, x, in_ch=5):
    """DynamicMultiBandLinearGenerator.forward, muti_kernel/networks.py:119-133."""
    scales = condition_scales(sd, x, in_ch=in_ch)
    B = x.shape[0]
    return np.stack([np.stack([chain_forward(x[b, c], muti_band_weights(sd, c), _split_scales(scales[b, c]))
                               for c in range(in_ch)]) for b in range(B)])


def single_band_weights(sd, band):
    return [sd[f"chains.{band}.{l}.weight"] for l in range(6)]


def single_extract_effective_kernels(sd, in_ch=5):
    """MultiBandLinearGenerator.extract_effective_kernels, single_kernel/networks.py:108-157."""
    return np.stack([compose_kernel(single_band_weights(sd, c)) for c in range(in_ch)])


def single_generator_forward(sd, x, in_ch=5):
    """MultiBandLinearGenerator.forward, single_kernel/networks.py:83-106."""
    return np.stack([np.stack([chain_forward(x[b, c], single_band_weights(sd, c)) for c in range(in_ch)])
                     for b in range(x.shape[0])])


def noise_estimator_sigma(log_sigma, sigma_max):
    """NoiseEstimator.forward, muti_kernel/networks.py:222: clamp(exp(log_sigma), 1e-5, sigma_max)."""
    return np.clip(np.exp(np.asarray(log_sigma, np.float32)), 1e-5, sigma_max).astype(np.float32)
