"""The degradation expressed with STOCK torch CPU ops -- TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the product).

SURVEY.md 8(d) asks for the reference's CPU path timed beside the HIP kernel in two forms: the C restatement
(kmsr_oracle.c) and "the same math expressed with stock torch CPU ops ... (F.pad + F.conv2d + F.avg_pool2d as in
C_30:109-122)", i.e. the way the reference itself spends its time on a CPU (ATen / oneDNN grouped convolution at full
resolution followed by the discard / pooling).  bench.py's `cpu_baseline` leg times these two functions; the CPU tests
pin them to the same reference-generated fixtures as the C oracle (tests/test_oracle_golden.py).
"""
import math

import torch
import torch.nn.functional as F


def va_torch(img, kernel, downscale_factor=8):
    """V-A, C_30apply_kernel_to_landsat.py:83-122 step by step: broadcast a 2-D kernel to all bands, divide each band's
    kernel by its sum when positive, replicate-pad by k//2, depthwise cross-correlation, int(log2 f) 2x2 average pools."""
    img = torch.as_tensor(img, dtype=torch.float32)
    k = torch.as_tensor(kernel, dtype=torch.float32)
    bands = img.shape[0]
    if k.ndim == 2:
        k = k[None].expand(bands, -1, -1)
    assert k.shape[0] == bands, f"kernel bands ({k.shape[0]}) != image bands ({bands})"
    sums = k.sum(dim=(1, 2), keepdim=True)
    k = torch.where(sums > 0, k / sums, k)
    ph, pw = k.shape[-2] // 2, k.shape[-1] // 2
    y = F.conv2d(F.pad(img[None], (pw, pw, ph, ph), mode="replicate"), k[:, None], groups=bands)
    for _ in range(int(math.log2(downscale_factor))):
        y = F.avg_pool2d(y, 2, 2)
    return y[0]


def vb_torch(x, kernels, stride=4):
    """V-B clean part, muti_kernel/train_gemini.py:124-134: every (sample, band) plane is one group of a zero-padded
    grouped convolution at full resolution, then every `stride`-th row / column is kept."""
    x = torch.as_tensor(x, dtype=torch.float32)
    k = torch.as_tensor(kernels, dtype=torch.float32)
    B, Cb, H, W = x.shape
    ks = k.shape[-1]
    y = F.conv2d(x.reshape(1, B * Cb, H, W), k.reshape(B * Cb, 1, ks, ks), padding=ks // 2, groups=B * Cb)
    return y.reshape(B, Cb, H, W)[:, :, ::stride, ::stride]


def convk_torch(x, w, k, pad="reflect", bias=None, dtype=torch.float64):
    """The KernelGAN layer (muti_kernel/networks.py:108-112: F.pad(mode='reflect') + F.conv2d with per-unit weights) and the
    critic's zero-padded conv (muti_kernel/networks.py:257-264), in the kmsr_convk_* weight layout:
    x [U,Cin,H,W]; w [U,Cin,k*k,Cout] (per unit) or [Cin,k*k,Cout] (shared) -> [U,Cout,H,W].  Differentiable (torch
    autograd), float64 by default: the checker of kmsr_convk_fwd / dgrad / wgrad_f32."""
    x = x.to(dtype)
    w = w.to(dtype)
    U, Cin, H, W = x.shape
    p = k // 2
    xp = F.pad(x, (p, p, p, p), mode="reflect") if (pad == "reflect" and p) else F.pad(x, (p, p, p, p))
    if w.ndim == 3:
        wt = w.permute(2, 0, 1).reshape(w.shape[2], Cin, k, k)
        return F.conv2d(xp, wt, None if bias is None else bias.to(dtype))
    outs = [F.conv2d(xp[u:u + 1], w[u].permute(2, 0, 1).reshape(w.shape[3], Cin, k, k)) for u in range(U)]
    return torch.cat(outs)
