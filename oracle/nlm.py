"""Oracle of the NLM denoise row (denoise/denoise.py:34-67) -- TEST INFRASTRUCTURE ONLY, and **PARITY UNPINNED**.

The reference's arithmetic lives in third-party code that is neither vendored nor versioned by the reference and is not
installed here: scikit-image `restoration.denoise_nl_means(fast_mode=True)` / `estimate_sigma`, and PyWavelets (`db2`).
The reference ships no fixture for this path.  This file therefore restates the PUBLISHED algorithms from their
documentation / source as remembered (specification risk):

  estimate_sigma : sigma = median(|HH| over non-zero coefficients) / Phi^-1(0.75), HH = level-1 `db2` diagonal detail
                   coefficients, 'symmetric' signal extension, cD[k] = sum_j dec_hi[j] * x_ext[2k + 1 - j].
  fast NLM (2-D) : image reflect-padded by offset + d + 1 (offset = s // 2); for every shift t in [-d, d]^2 the patch
                   distance is the sum over the box rows (r-offset, r+offset], cols (c-offset, c+offset] of
                   (v[x] - v[x+t])^2 - 2 sigma^2, clipped at 0, divided by h^2 s^2; weights exp(-dist) with dist > 5
                   dropped; shifts with t_col == 0 are visited from both ends with alpha = 0.5, which for the cropped
                   output is the same as one full-weight visit.  Net effect per output pixel p:
                       out[p] = sum_t w_t(p) v[p+t] / sum_t w_t(p),   w_t(p) = [D_t(p) <= 5] exp(-D_t(p)).
  (scikit-image evaluates the box sums through float integral images; this oracle sums them directly in float64.)
"""
import numpy as np

DEC_HI_DB2 = np.array([-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145])
NORM_PPF_075 = 0.6744897501960817
DISTANCE_CUTOFF = 5.0


def _dwt_hi_symmetric(x, axis):
    """cD[k] = sum_j dec_hi[j] * x_ext[2k+1-j], k < (N+3)//2, half-sample symmetric extension."""
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    N = x.shape[-1]
    ext = np.concatenate([x[..., 2::-1], x, x[..., :-4:-1]], axis=-1)        # 3 samples each side
    nout = (N + 3) // 2
    out = np.zeros(x.shape[:-1] + (nout,))
    for j in range(4):
        idx = 2 * np.arange(nout) + 1 - j + 3                                  # +3: offset of x[0] inside ext
        out += DEC_HI_DB2[j] * ext[..., idx]
    return np.moveaxis(out, -1, axis)


def estimate_sigma(img):
    """skimage.restoration.estimate_sigma(img, average_sigmas=True) for one 2-D band."""
    hh = _dwt_hi_symmetric(_dwt_hi_symmetric(img, 0), 1)
    c = np.abs(hh[hh != 0])
    return float(np.median(c) / NORM_PPF_075)


def nl_means_fast(img, patch_size=7, patch_distance=11, h=0.1, sigma=0.0):
    img = np.asarray(img, np.float64)
    s = patch_size + (1 - patch_size % 2)
    off, d = s // 2, patch_distance
    pad = off + d + 1
    v = np.pad(img, pad, mode="reflect")
    H, W = img.shape
    h2s2 = h * h * s * s
    var2 = 2.0 * sigma * sigma
    num = np.zeros((H, W))
    den = np.zeros((H, W))
    rr = slice(pad, pad + H)
    cc = slice(pad, pad + W)
    for tr in range(-d, d + 1):
        for tc in range(-d, d + 1):
            sh = np.roll(v, (-tr, -tc), axis=(0, 1))                           # sh[x] = v[x + t]  (wrap never reaches the crop)
            dlt = (v - sh) ** 2 - var2
            ii = np.pad(np.cumsum(np.cumsum(dlt, 0), 1), ((1, 0), (1, 0)))     # ii[a,b] = sum dlt[:a,:b]
            r = np.arange(pad, pad + H)[:, None]
            c = np.arange(pad, pad + W)[None, :]
            box = (ii[r + off + 1, c + off + 1] - ii[r - off + 1, c + off + 1] - ii[r + off + 1, c - off + 1]
                   + ii[r - off + 1, c - off + 1])                             # rows (r-off, r+off], cols (c-off, c+off]
            dist = np.maximum(box, 0.0) / h2s2
            w = np.where(dist > DISTANCE_CUTOFF, 0.0, np.exp(-dist))
            num += w * sh[rr, cc]
            den += w
    return (num / den).astype(np.float32)


def denoise_band_float_nlm(img, h_factor=1.15, patch_size=7, patch_distance=11):
    """denoise/denoise.py:34-67"""
    img = np.asarray(img, np.float32)
    valid = ~np.isnan(img)
    if not valid.any():
        return img, 0.0
    filled = np.nan_to_num(img, nan=np.nanmean(img)).astype(np.float32)
    sig = estimate_sigma(filled)
    den = nl_means_fast(filled, patch_size, patch_distance, h_factor * sig, sig)
    return np.where(valid, den, np.nan).astype(np.float32), sig
