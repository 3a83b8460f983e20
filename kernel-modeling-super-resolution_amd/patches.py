"""Water mask + patch cutting on the GPU -- the step that feeds the degradation path.

Mirrors kernel_from_lr_gan/A_00_patch_cutter_universal.py:89-197 (same names / arguments / defaults):
    apply_water_mask(data, threshold_min, threshold_max)                        :89-123
    create_patches(data, patch_size, stride_ratio, nan_threshold)               :126-197  (array-level: returns the kept
        patches and their grid positions instead of writing one NetCDF file per patch -- NetCDF I/O is out of scope, N4)
NaN counting and the keep/drop decision are exact integer arithmetic, identical to the reference's.
"""
import numpy as np
import torch

from . import _lib as L

PATCH_SIZE = 256          # A_00_patch_cutter_universal.py:29-36
STRIDE_RATIO = 0.5
NAN_THRESHOLD = 0.0
THRESHOLD_MIN = 0.000001
THRESHOLD_MAX = 7.0
NIR_BAND_INDEX = 4
INVALID_VALUE = -9999.0


def apply_water_mask(data, threshold_min=THRESHOLD_MIN, threshold_max=THRESHOLD_MAX, verbose=False):
    """data [C,H,W] (numpy or CUDA tensor) -> masked array of the same kind: invalid marker -> NaN, pixels whose NIR band
    is outside [threshold_min, threshold_max] -> NaN in every band (:102-114).  Unlike the reference, the input is not
    modified in place.  Returns (masked, total_valid, water_pixels) -- the reference prints the two counts (:116-121)."""
    dev = L.require_gpu()
    lib = L.load()
    was_np = isinstance(data, np.ndarray)
    d = (torch.from_numpy(data) if was_np else data).to(dev, torch.float32).contiguous()
    C, H, W = d.shape
    out = torch.empty_like(d)
    counts = torch.zeros(2, device=dev, dtype=torch.int64)
    L.check(lib.kmsr_water_mask_f32(L.ptr(d), C, H, W, NIR_BAND_INDEX, INVALID_VALUE, float(threshold_min),
                                    float(threshold_max), L.ptr(out), L.ptr(counts), L.current_stream()))
    total_valid, water = (int(v) for v in counts.cpu())
    if verbose:
        ratio = water / total_valid * 100 if total_valid > 0 else 0
        print(f"  valid pixels: {total_valid:,}\n  water pixels: {water:,} ({ratio:.2f}%)")
    return (out.cpu().numpy() if was_np else out), total_valid, water


def create_patches(data, patch_size=PATCH_SIZE, stride_ratio=STRIDE_RATIO, nan_threshold=NAN_THRESHOLD):
    """data [C,H,W] -> dict(total, kept, index [K,4] = (i, j, h_start, w_start), nan_counts [K], patches [K,C,ps,ps]).
    Grid, stride and the `nan_ratio > nan_threshold` rejection follow :149-180 exactly (float64 ratio of integer counts)."""
    dev = L.require_gpu()
    lib = L.load()
    was_np = isinstance(data, np.ndarray)
    d = (torch.from_numpy(data) if was_np else data).to(dev, torch.float32).contiguous()
    C, H, W = d.shape
    stride = int(patch_size * stride_ratio)                                   # :150
    if H < patch_size or W < patch_size or stride <= 0:
        return dict(total=0, kept=0, index=np.zeros((0, 4), np.int64), nan_counts=np.zeros(0, np.int64),
                    patches=(np.zeros((0, C, patch_size, patch_size), np.float32) if was_np else
                             torch.zeros((0, C, patch_size, patch_size), device=dev)))
    hp, wp = (H - patch_size) // stride + 1, (W - patch_size) // stride + 1   # :152-153
    cnt = torch.empty(hp * wp, device=dev, dtype=torch.int32)
    L.check(lib.kmsr_patch_nan_counts_f32(L.ptr(d), C, H, W, patch_size, stride, L.ptr(cnt), L.current_stream()))
    counts = cnt.cpu().numpy().astype(np.int64)
    size = C * patch_size * patch_size
    keep = [(t // wp, t % wp) for t in range(hp * wp) if not (counts[t] / size > nan_threshold)]     # :171-175
    index = np.array([(i, j, i * stride, j * stride) for i, j in keep], np.int64).reshape(-1, 4)
    tl = torch.from_numpy(index[:, 2:4].astype(np.int32)).contiguous().to(dev)
    patches = torch.empty((len(keep), C, patch_size, patch_size), device=dev, dtype=torch.float32)
    L.check(lib.kmsr_gather_patches_f32(L.ptr(d), C, H, W, patch_size, L.ptr(tl), len(keep), L.ptr(patches),
                                        L.current_stream()))
    return dict(total=hp * wp, kept=len(keep), index=index,
                nan_counts=np.array([counts[i * wp + j] for i, j in keep], np.int64),
                patches=patches.cpu().numpy() if was_np else patches)
