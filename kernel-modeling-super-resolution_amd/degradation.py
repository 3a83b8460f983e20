"""Host-side mirror of the reference's degradation call surface, executing on MI355X through the
C ABI (include/kmsr.h).  Same names, argument meaning and error behaviour as the reference:

    load_kernel                 C_30apply_kernel_to_landsat.py:18-33, C_31apply_muti_kernel_to_landsat.py:22-37
    apply_kernel_degradation    C_30:68-124 == C_31:59-97                       (V-A)
    add_noise                   E_make_train_data.py:65-74                       (V-C)
    random_crop                 D_build_noise_pool.py:41-53
    build_noise_pool_from_arrays  D_build_noise_pool.py:56-132 (array-level; NetCDF I/O is out of scope)

plus the batched forms the reference's per-file loops (C_30:152, E:223) collapse into:

    degrade_batch               one fused launch for a whole [N,C,H,W] batch (V-A, V-B and V-C semantics by flags)
    add_noise_batch             E:223-250 for N files at once (same np.random stream as N sequential add_noise calls)

No CPU fallback exists: without libkmsr_hip.so or a GPU these functions raise.
"""
import os
import random
import weakref

import numpy as np
import torch

from . import _lib as L

BAND_NAMES = ['L_TOA_443', 'L_TOA_490', 'L_TOA_555', 'L_TOA_660', 'L_TOA_865']   # C_30:47, D:23, E:28


def load_kernel(kernel_path: str, verbose: bool = True) -> torch.Tensor:
    """C_31:22-37 (superset of C_30:18-33): np.load -> float32; [B,C,kH,kW] -> mean over B;
    [kH,kW] -> [1,kH,kW].  Returns a CPU tensor like the reference."""
    kernel = np.load(kernel_path)
    kernel_tensor = torch.from_numpy(kernel.astype(np.float32))
    if kernel_tensor.ndim == 4:
        kernel_tensor = kernel_tensor.mean(dim=0)
    if kernel_tensor.ndim == 2:
        kernel_tensor = kernel_tensor.unsqueeze(0)
    if verbose:
        print(f"loaded blur kernel: {os.path.basename(kernel_path)}")
        print(f"  shape: {tuple(kernel_tensor.shape)}")
        print(f"  sum: {kernel_tensor.sum().item():.6f}")
    return kernel_tensor


def _as_dev_f32(t, dev):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.to(device=dev, dtype=torch.float32)


_CHECKED_IDX = {}     # what -> (weakref to the tensor OBJECT, version, numel, bound) of the device index tensor last range-checked


def _checked_index(idx, n, bound, what, device, dtype):
    """Index tensors are range-checked on the host (the reference's np.random.randint / einsum gather cannot leave
    [0, bound); an out-of-range index here would be an out-of-bounds DEVICE read of taps, sigma or the noise pool).
    CPU tensors are checked before the copy (no device sync); a device tensor costs one min/max read-back the first time
    it is seen and nothing on later calls with the same unmodified tensor OBJECT (the steady state of a training loop).  The
    cache entry is bound to the object by a weak reference: a fresh tensor that lands on a recycled allocator block (same
    data_ptr, version 0) is a different object and is checked again."""
    if idx.numel() != n:
        raise AssertionError(f"{what} must hold one index per sample ({n}), got {idx.numel()}")
    if idx.device.type != "cuda":
        if n and (int(idx.min()) < 0 or int(idx.max()) >= bound):
            raise IndexError(f"{what} out of range [0, {bound}): min {int(idx.min())}, max {int(idx.max())}")
        return idx.to(device=device, dtype=dtype).contiguous()
    ent = _CHECKED_IDX.get(what)
    if ent is None or ent[0]() is not idx or ent[1:] != (idx._version, n, bound):
        if n:
            lo, hi = (int(v) for v in torch.aminmax(idx))
            if lo < 0 or hi >= bound:
                _CHECKED_IDX.pop(what, None)
                raise IndexError(f"{what} out of range [0, {bound}): min {lo}, max {hi}")
        _CHECKED_IDX[what] = (weakref.ref(idx), idx._version, n, bound)
    return idx.to(device=device, dtype=dtype).contiguous()


def _sample_addr(sample_base, sample_step, N, Cc):
    """Validate the global-sample addressing of the noise stream: the plane counter (global sample * C + band) is 32 bits."""
    base, step = int(sample_base), int(sample_step)
    if base < 0 or step < 1 or (base + max(N - 1, 0) * step + 1) * Cc > 2 ** 32:
        raise ValueError(f"noise stream addressing out of range: sample_base={base}, sample_step={step}, N={N}, C={Cc}")
    return base, step


def degrade_batch(hr, kernels, factor, *, down="avgpool", pad="replicate", normalize=True,
                  kernel_idx=None, sigma=None, noise="none", seed=0, offset=0, sample_base=0, sample_step=1,
                  noise_pool=None, noise_idx=None, out=None, flags=0, rows_per_chunk=0, workspace=None,
                  taps_ready=False, validate_idx=True):
    """One fused blur + downsample + noise launch over a batch.

    hr       [N,C,H,W] float32 CUDA tensor (any N/C/H strides, unit W stride)
    kernels  [K,C,kh,kw] float32 CUDA tensor (K == 1, K == N, or any K with kernel_idx [N] int32)
    down     "avgpool" (C_30:120-122) | "stride" (train_gemini.py:134)
    pad      "replicate" (C_30:107-109) | "zero" (train_gemini.py:128)
    noise    "none" | "gaussian" (sigma [K,C], train_gemini.py:137-138) | "pool" (noise_pool [P,C,h,w], E:72-74)
    seed / offset / sample_base / sample_step: the counter-based Gaussian stream is addressed by (seed, offset, GLOBAL sample
             index, band, row, col); local sample n is global sample sample_base + n * sample_step.  A rank that holds a
             shard of a larger batch passes its shard's position (dist.sample_base(N) for contiguous blocks; base=rank,
             step=world for patch_ids[rank::world]) and draws exactly the noise one process draws for those samples.
    workspace / taps_ready: pass the same float32 CUDA tensor to successive calls that use the SAME kernels and
             geometry and set taps_ready=True from the second call on to skip rebuilding the tap table.
    validate_idx: range-check kernel_idx / noise_idx on the host (IndexError); pass False only for indices that are in
             range by construction (an argmax over K logits), to keep the call free of a device read-back.
    Returns lr [N,C,h,w] float32 on the same device.
    """
    dev = L.require_gpu()
    lib = L.load()
    if hr.device.type != "cuda" or hr.dtype != torch.float32 or hr.ndim != 4:
        raise ValueError("hr must be a 4-D float32 CUDA tensor")
    if hr.stride(3) != 1:
        hr = hr.contiguous()
    kernels = kernels.to(device=hr.device, dtype=torch.float32).contiguous()
    if kernels.ndim != 4:
        raise ValueError(f"kernels must be [K,C,kh,kw], got {tuple(kernels.shape)}")
    N, Cc, H, W = hr.shape
    assert kernels.shape[1] == Cc, f"kernel bands ({kernels.shape[1]}) != image bands ({Cc})"   # C_30:88
    d = L.DegradeDesc()
    d.N, d.C, d.H, d.W = N, Cc, H, W
    d.hr_stride_n, d.hr_stride_c, d.hr_stride_h = hr.stride(0), hr.stride(1), hr.stride(2)
    d.n_kernels, d.kh, d.kw = kernels.shape[0], kernels.shape[2], kernels.shape[3]
    d.factor = int(factor)
    d.down_mode = {"avgpool": L.DOWN_AVGPOOL, "stride": L.DOWN_STRIDE}[down]
    d.pad_mode = {"replicate": L.PAD_REPLICATE, "zero": L.PAD_ZERO}[pad]
    d.normalize = int(bool(normalize))
    d.noise_mode = {"none": L.NOISE_NONE, "gaussian": L.NOISE_GAUSSIAN, "pool": L.NOISE_POOL}[noise]
    d.flags = int(flags)
    d.seed, d.offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 32 - 1)
    d.noise_sample_base, d.noise_sample_step = _sample_addr(sample_base, sample_step, N, Cc)
    d.rows_per_chunk = int(rows_per_chunk)
    h, w = L.C.c_int32(), L.C.c_int32()
    L.check(lib.kmsr_degrade_out_shape(d, h, w))
    h, w = h.value, w.value
    if kernel_idx is not None:
        if validate_idx:
            kernel_idx = _checked_index(kernel_idx, N, kernels.shape[0], "kernel_idx", hr.device, torch.int32)
        else:
            kernel_idx = kernel_idx.to(device=hr.device, dtype=torch.int32).contiguous()
            assert kernel_idx.numel() == N
    if d.noise_mode == L.NOISE_GAUSSIAN:
        sigma = sigma.to(device=hr.device, dtype=torch.float32).contiguous()
        assert sigma.shape == (kernels.shape[0], Cc), f"sigma must be [K,C], got {tuple(sigma.shape)}"
    else:
        sigma = None
    if d.noise_mode == L.NOISE_POOL:
        noise_pool = noise_pool.to(device=hr.device, dtype=torch.float32).contiguous()
        assert noise_pool.shape[1:] == (Cc, h, w), \
            f"noise pool entries {tuple(noise_pool.shape[1:])} != LR shape {(Cc, h, w)}"          # E:239-247 guard
        d.pool_size = noise_pool.shape[0]
        if noise_idx is not None:
            if validate_idx:
                noise_idx = _checked_index(noise_idx, N, noise_pool.shape[0], "noise_idx", hr.device, torch.int64)
            else:
                noise_idx = noise_idx.to(device=hr.device, dtype=torch.int64).contiguous()
                assert noise_idx.numel() == N
    else:
        noise_pool, noise_idx = None, None
    if out is None:
        out = torch.empty((N, Cc, h, w), device=hr.device, dtype=torch.float32)
    else:
        assert out.shape == (N, Cc, h, w) and out.is_contiguous() and out.dtype == torch.float32
    ws_bytes = lib.kmsr_degrade_workspace_bytes(d)
    if workspace is None:
        assert not taps_ready, "taps_ready needs the workspace of the earlier call"
        ws = torch.empty(max(ws_bytes, 4) // 4, device=hr.device, dtype=torch.float32)
    else:
        ws = workspace
        assert ws.is_cuda and ws.dtype == torch.float32 and ws.is_contiguous() and ws.numel() * 4 >= ws_bytes
        ws_bytes = ws.numel() * 4
    if taps_ready:
        d.flags |= L.FLAG_TAPS_READY
    with torch.cuda.device(hr.device):
        L.check(lib.kmsr_degrade_f32(d, L.ptr(hr), L.ptr(kernels), L.ptr(kernel_idx), L.ptr(sigma), L.ptr(noise_pool),
                                     L.ptr(noise_idx), L.ptr(out), L.ptr(ws), ws_bytes, L.current_stream()))
    return out



def degrade_backward(hr, grad_lr, ksize, factor, *, pad="zero", seed=0, offset=0, sample_base=0, sample_step=1,
                     want_kernels=True, want_sigma=True):
    """Gradients of the V-B degradation (down="stride", normalize=False) w.r.t. per-sample kernels and sigmas
    (kmsr_degrade_bwd_f32; the chain the reference differentiates at train_gemini.py:117-138 / :219-222).
    hr [N,C,H,W], grad_lr [N,C,ceil(H/f),ceil(W/f)], ksize (kh, kw) -> (dkernels [N,C,kh,kw] | None, dsigma [N,C] | None)."""
    L.require_gpu()
    lib = L.load()
    if hr.device.type != "cuda" or hr.dtype != torch.float32 or hr.ndim != 4:
        raise ValueError("hr must be a 4-D float32 CUDA tensor")
    if hr.stride(3) != 1:
        hr = hr.contiguous()
    N, Cc, H, W = hr.shape
    kh, kw = (ksize, ksize) if isinstance(ksize, int) else ksize
    d = L.DegradeDesc()
    d.N, d.C, d.H, d.W = N, Cc, H, W
    d.hr_stride_n, d.hr_stride_c, d.hr_stride_h = hr.stride(0), hr.stride(1), hr.stride(2)
    d.n_kernels, d.kh, d.kw, d.factor = N, int(kh), int(kw), int(factor)
    d.down_mode, d.normalize = L.DOWN_STRIDE, 0
    d.pad_mode = {"replicate": L.PAD_REPLICATE, "zero": L.PAD_ZERO}[pad]
    d.noise_mode = L.NOISE_GAUSSIAN if want_sigma else L.NOISE_NONE
    d.seed, d.offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 32 - 1)
    d.noise_sample_base, d.noise_sample_step = _sample_addr(sample_base, sample_step, N, Cc)
    h, w = -(-H // int(factor)), -(-W // int(factor))
    grad_lr = grad_lr.to(torch.float32).contiguous()
    assert grad_lr.is_cuda and tuple(grad_lr.shape) == (N, Cc, h, w), \
        f"grad_lr {tuple(grad_lr.shape)} != LR shape {(N, Cc, h, w)}"
    nbytes = lib.kmsr_degrade_bwd_workspace_bytes(d)
    ws = torch.empty(max(nbytes, 4) // 4, device=hr.device, dtype=torch.float32)   # nbytes == 0: the call reports why
    dk = torch.empty((N, Cc, kh, kw), device=hr.device, dtype=torch.float32) if want_kernels else None
    ds = torch.empty((N, Cc), device=hr.device, dtype=torch.float32) if want_sigma else None
    with torch.cuda.device(hr.device):
        L.check(lib.kmsr_degrade_bwd_f32(d, L.ptr(hr), L.ptr(grad_lr), L.ptr(dk), L.ptr(ds), L.ptr(ws), nbytes,
                                         L.current_stream()))
    return dk, ds


class DegradeFunction(torch.autograd.Function):
    """Differentiable V-B degradation: out = blur_stride(x, kernels) + sigma * eps, HIP kernels in both directions.
    kernels [N,C,kh,kw] and sigma [N,C] are per sample (mix / gather a bank before calling so autograd finishes the
    chain rule); x is treated as data (no gradient)."""

    @staticmethod
    def forward(ctx, x, kernels, sigma, factor, pad, seed, offset, sample_base=0, sample_step=1):
        out = degrade_batch(x, kernels.detach(), factor, down="stride", pad=pad, normalize=False,
                            sigma=None if sigma is None else sigma.detach(),
                            noise="none" if sigma is None else "gaussian", seed=seed, offset=offset,
                            sample_base=sample_base, sample_step=sample_step)
        ctx.save_for_backward(x)
        ctx.meta = (tuple(kernels.shape[2:]), factor, pad, seed, offset, sigma is not None, sample_base, sample_step)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        ksize, factor, pad, seed, offset, has_sigma, sample_base, sample_step = ctx.meta
        want_k, want_s = ctx.needs_input_grad[1], has_sigma and ctx.needs_input_grad[2]
        dk = ds = None
        if want_k or want_s:
            dk, ds = degrade_backward(x, g, ksize, factor, pad=pad, seed=seed, offset=offset, sample_base=sample_base,
                                      sample_step=sample_step, want_kernels=want_k, want_sigma=want_s)
        return None, dk, ds, None, None, None, None, None, None


def workspace_for(hr_shape, kernels_shape, factor, down="avgpool", device=None):
    """A workspace tensor sized for degrade_batch on this geometry."""
    lib = L.load()
    d = L.DegradeDesc()
    d.N, d.C, d.H, d.W = hr_shape
    d.n_kernels, d.kh, d.kw = kernels_shape[0], kernels_shape[2], kernels_shape[3]
    d.factor = int(factor)
    d.down_mode = {"avgpool": L.DOWN_AVGPOOL, "stride": L.DOWN_STRIDE}[down]
    nbytes = lib.kmsr_degrade_workspace_bytes(d)
    return torch.empty(max(nbytes, 4) // 4, device=device or L.require_gpu(), dtype=torch.float32)


def uses_fast_path(hr, kernels, factor, down="avgpool", flags=0):
    """True when degrade_batch will run the row-streaming kernel for this configuration."""
    lib = L.load()
    d = L.DegradeDesc()
    d.N, d.C, d.H, d.W = hr.shape
    d.hr_stride_n, d.hr_stride_c, d.hr_stride_h = hr.stride(0), hr.stride(1), hr.stride(2)
    d.n_kernels, d.kh, d.kw = kernels.shape[0], kernels.shape[2], kernels.shape[3]
    d.factor = int(factor)
    d.down_mode = {"avgpool": L.DOWN_AVGPOOL, "stride": L.DOWN_STRIDE}[down]
    d.flags = flags
    return lib.kmsr_degrade_uses_fast_path(d, L.ptr(hr)) == 1


def apply_kernel_degradation(img: torch.Tensor, kernel: torch.Tensor, downscale_factor: int = 8) -> torch.Tensor:
    """Drop-in for C_30apply_kernel_to_landsat.py:68-124 / C_31apply_muti_kernel_to_landsat.py:59-97.

    img [C,H,W], kernel [C,kH,kW] or [kH,kW]; returns [C, H>>s, W>>s], s = int(log2(downscale_factor)),
    on img's device (a CPU input is moved to the GPU and the result moved back, like the reference
    returns a fresh CPU tensor).  AssertionError on band mismatch (C_30:88), ValueError on a bad
    kernel rank (C_31:69).
    """
    dev = L.require_gpu()
    C, H, W = img.shape
    if kernel.ndim == 2:
        kernel = kernel.unsqueeze(0).repeat(C, 1, 1)                      # C_30:83-85
    elif kernel.ndim == 3:
        assert kernel.shape[0] == C, f"kernel bands ({kernel.shape[0]}) != image bands ({C})"   # C_30:88
    else:
        raise ValueError(f"unsupported kernel shape: {tuple(kernel.shape)}")   # C_31:69
    src_dev = img.device
    hr = img.to(device=dev, dtype=torch.float32).unsqueeze(0)
    out = degrade_batch(hr, kernel.to(dev).unsqueeze(0), downscale_factor, down="avgpool", pad="replicate",
                        normalize=True)
    return out.squeeze(0).to(src_dev)


def add_noise(blurred: np.ndarray, noise_pool: np.ndarray) -> np.ndarray:
    """Drop-in for E_make_train_data.py:65-74: one np.random.randint draw (global NumPy RNG, high exclusive),
    then blurred + pool[idx], computed on the GPU."""
    idx = np.random.randint(0, len(noise_pool))
    return add_noise_batch(blurred[None], noise_pool, idx=np.array([idx]))[0]


def add_noise_batch(blurred, noise_pool, idx=None):
    """N files at once.  idx=None draws N indices from the global NumPy RNG in file order, which yields the
    same stream as N sequential add_noise calls (E:223-250).  Arrays may be numpy or CUDA tensors; numpy in ->
    numpy out."""
    dev = L.require_gpu()
    N = blurred.shape[0]
    if idx is None:
        idx = np.array([np.random.randint(0, len(noise_pool)) for _ in range(N)], np.int64)
    was_np = isinstance(blurred, np.ndarray)
    b = _as_dev_f32(blurred, dev)
    pool = _as_dev_f32(noise_pool, dev)
    assert pool.shape[1:] == b.shape[1:], f"pool entry {tuple(pool.shape[1:])} != blurred {tuple(b.shape[1:])}"
    # identity blur through the same fused kernel: 1x1 kernel, stride 1, noise=pool
    Cc = b.shape[1]
    one = torch.ones((1, Cc, 1, 1), device=dev)
    out = degrade_batch(b, one, 1, down="stride", pad="zero", normalize=False, noise="pool", noise_pool=pool,
                        noise_idx=torch.as_tensor(np.asarray(idx, np.int64)))
    return out.cpu().numpy() if was_np else out


def random_crop(data: np.ndarray, crop_size: int, n_samples: int):
    """Drop-in for D_build_noise_pool.py:41-53 (host slicing; Python `random`, inclusive bounds)."""
    _, H, W = data.shape
    if H < crop_size or W < crop_size:
        raise ValueError(f"image size {H}x{W} smaller than crop size {crop_size}")       # D:44-45
    patches = []
    for _ in range(n_samples):
        top = random.randint(0, H - crop_size)
        left = random.randint(0, W - crop_size)
        patches.append(data[:, top:top + crop_size, left:left + crop_size])
    return patches


def noise_crops(geo, denoised, offsets, crop_size):
    """(geo - denoised)[:, top:top+cs, left:left+cs] for every (top,left) in offsets, on the GPU
    (D_build_noise_pool.py:88 + :49-51).  geo/denoised [C,H,W] numpy or CUDA; returns [n,C,cs,cs]."""
    dev = L.require_gpu()
    lib = L.load()
    was_np = isinstance(geo, np.ndarray)
    g = _as_dev_f32(geo, dev).contiguous()
    d = _as_dev_f32(denoised, dev).contiguous()
    assert g.shape == d.shape and g.ndim == 3
    Cc, H, W = g.shape
    offs = np.ascontiguousarray(offsets, np.int32).reshape(-1, 2)
    n = offs.shape[0]
    if H < crop_size or W < crop_size:
        raise ValueError(f"image size {H}x{W} smaller than crop size {crop_size}")
    if n and (offs.min() < 0 or (offs[:, 0] + crop_size).max() > H or (offs[:, 1] + crop_size).max() > W):
        raise ValueError("crop offset out of range")
    out = torch.empty((n, Cc, crop_size, crop_size), device=dev)
    offs_d = torch.from_numpy(offs).to(dev)
    L.check(lib.kmsr_noise_crops_f32(L.ptr(g), L.ptr(d), Cc, H, W, L.ptr(offs_d), n, crop_size, L.ptr(out),
                                     L.current_stream()))
    return out.cpu().numpy() if was_np else out


def build_noise_pool_from_arrays(pairs, samples_per_file=1, patch_size=32, seed=42):
    """Array-level D_build_noise_pool.build_noise_pool (D:56-132): `pairs` yields (geo, denoised) [C,H,W]
    arrays in file order; seeds Python `random` and NumPy like D:65-66, draws (top,left) per sample exactly
    like random_crop (D:49-50), computes the crops of geo-denoised on the GPU and stacks them (D:110).
    Pairs that raise are skipped like D:102-104."""
    random.seed(seed)
    np.random.seed(seed)
    chunks = []
    for geo, den in pairs:
        try:
            _, H, W = geo.shape
            if H < patch_size or W < patch_size:
                raise ValueError(f"image size {H}x{W} smaller than crop size {patch_size}")
            offs = [(random.randint(0, H - patch_size), random.randint(0, W - patch_size))
                    for _ in range(samples_per_file)]
            chunks.append(noise_crops(geo, den, offs, patch_size))
        except Exception as e:  # noqa: BLE001  (reference: print and continue)
            print(f"skipping pair: {e}")
            continue
    if not chunks:
        raise RuntimeError("no noise patches extracted")                                   # D:106-107
    if isinstance(chunks[0], np.ndarray):
        return np.concatenate(chunks, axis=0)
    return torch.cat(chunks, dim=0)
