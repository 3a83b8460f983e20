"""NLM denoise stage on the GPU -- **parity unpinned** (see include/kmsr.h: kmsr_nlm_f32 and oracle/nlm.py).

Mirrors denoise/denoise.py:34-67 `denoise_band_float_nlm(img_float, h_factor, patch_size, patch_distance)`:
NaN -> band mean, sigma-hat by the wavelet MAD estimator, NLM with h = h_factor * sigma-hat, NaN mask restored.
README.MD:17 of the reference: h_factor 1.8 for GOCI-2, 1.0 for Landsat-8/9.  The scikit-image / PyWavelets arithmetic
the reference relies on is absent here, so the numbers are those of the restated published algorithm.
"""
import numpy as np
import torch

from . import _lib as L

_DEC_HI_DB2 = (-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145)


def _dwt_hi(x):
    """level-1 db2 high-pass along the last axis, 'symmetric' extension: cD[k] = sum_j dec_hi[j] x_ext[2k+1-j]."""
    ext = torch.cat([x[..., :3].flip(-1), x, x[..., -3:].flip(-1)], dim=-1)
    w = torch.tensor(_DEC_HI_DB2[::-1], device=x.device, dtype=x.dtype).view(1, 1, 4)      # window ext[2k+1 .. 2k+4]
    shp = ext.shape
    y = torch.nn.functional.conv1d(ext.reshape(-1, 1, shp[-1])[..., 1:], w, stride=2)
    return y.reshape(*shp[:-1], -1)[..., :(x.shape[-1] + 3) // 2]


def estimate_sigma(img):
    """[.., H, W] CUDA tensor -> sigma-hat per band: median(|HH != 0|) / 0.6745 (skimage estimate_sigma, db2)."""
    x = img.double()
    hh = _dwt_hi(_dwt_hi(x.transpose(-1, -2)).transpose(-1, -2))
    flat = hh.reshape(-1, hh.shape[-2] * hh.shape[-1]).abs()
    out = []
    for row in flat:
        nz = row[row != 0]
        # np.median averages the two middle values for even counts; torch.median returns the lower one
        s = torch.sort(nz).values
        n = s.numel()
        out.append(((s[(n - 1) // 2] + s[n // 2]) * 0.5 / 0.6744897501960817) if n else torch.zeros((), device=x.device, dtype=x.dtype))
    return torch.stack(out).reshape(img.shape[:-2]).float()


def nl_means(img, h, sigma, patch_size=7, patch_distance=11):
    """img [B,H,W] float32 CUDA (NaN-free), h / sigma [B] -> denoised [B,H,W]."""
    lib = L.load()
    img = img.contiguous()
    B, H, W = img.shape
    out = torch.empty_like(img)
    h = h.to(img.device, torch.float32).contiguous()
    sigma = sigma.to(img.device, torch.float32).contiguous()
    with torch.cuda.device(img.device):
        L.check(lib.kmsr_nlm_f32(L.ptr(img), B, H, W, int(patch_size), int(patch_distance), L.ptr(h), L.ptr(sigma), L.ptr(out),
                                 L.current_stream()))
    return out


def denoise_band_float_nlm(img_float, h_factor=1.15, patch_size=7, patch_distance=11, verbose=True):
    """Drop-in signature of denoise/denoise.py:34: one 2-D band (numpy) -> (denoised with the NaN mask restored, sigma-hat).
    Also accepts [B,H,W] to denoise several bands in one launch (returns a list of sigmas then)."""
    dev = L.require_gpu()
    arr = np.asarray(img_float, np.float32)
    single = arr.ndim == 2
    bands = arr[None] if single else arr
    valid = ~np.isnan(bands)
    if not valid.any():                                                     # :38-39
        return img_float, 0.0
    fill = np.array([np.nanmean(b) if v.any() else 0.0 for b, v in zip(bands, valid)], np.float32)   # :41-42
    filled = np.where(valid, bands, fill[:, None, None]).astype(np.float32)
    x = torch.from_numpy(filled).to(dev)
    sig = estimate_sigma(x)                                                 # :47
    hval = h_factor * sig                                                   # :50
    if verbose:
        for s_, h_ in zip(sig.tolist(), hval.tolist()):
            print(f"    -> Sigma: {s_:.6f} | h: {h_:.6f}")
    den = nl_means(x, hval, sig, patch_size, patch_distance).cpu().numpy()  # :55-62
    out = np.where(valid, den, np.nan).astype(np.float32)                   # :65
    if single:
        return out[0], float(sig[0])
    return out, [float(s) for s in sig]
