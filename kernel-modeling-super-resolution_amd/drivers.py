"""Batched counterparts of the reference's per-file L4 loops, at .npy level.

The reference walks a folder of NetCDF patches one file at a time (C_30apply_kernel_to_landsat.py:127-213,
C_31apply_muti_kernel_to_landsat.py:136-195, E_make_train_data.py:187-272).  NetCDF I/O is out of scope here
(`netCDF4` is not installable in this environment), so these drivers read / write the same arrays as
.npy files -- [C,H,W] float32 patches, the reference's own kernel .npy files -- and keep the loops' contracts:
sorted file order (C_30:140), skip-and-continue on a failing file with a printed message (C_30:205-209, E:264-267),
shape guards (E:239-247), one global-NumPy-RNG draw per file in file order (E:223-250).  The math is ONE fused launch
per batch of files instead of one call per file.
"""
import glob
import os

import numpy as np
import torch

from . import _lib as L
from .degradation import add_noise_batch, degrade_batch, load_kernel


def process_landsat_folder(input_dir, kernel_path, output_dir, downscale_factor=8, batch_size=64, suffix="_lr"):
    """C_30:127-213 / C_31:136-195 on .npy patches: every `*.npy` [C,H,W] in input_dir -> output_dir/<name><suffix>.npy
    holding apply_kernel_degradation(patch, kernel, downscale_factor).  Returns (n_success, n_failed)."""
    dev = L.require_gpu()
    kernel = load_kernel(kernel_path, verbose=False)
    files = sorted(glob.glob(os.path.join(input_dir, "*.npy")))             # C_30:140
    if not files:
        print(f"no .npy files in {input_dir}")                               # C_30:142-144
        return 0, 0
    os.makedirs(output_dir, exist_ok=True)
    ok = bad = 0
    pending = []                                                             # (path, array) of one shape

    def flush():
        nonlocal ok, bad, pending
        if not pending:
            return
        try:
            hr = torch.from_numpy(np.stack([a for _, a in pending])).to(dev)
            C = hr.shape[1]
            k = kernel
            if k.shape[0] == 1 and C != 1:
                k = k.repeat(C, 1, 1)                                        # 2-D kernel broadcast (C_30:83-85)
            assert k.shape[0] == C, f"kernel bands ({k.shape[0]}) != image bands ({C})"   # C_30:88
            lr = degrade_batch(hr, k.to(dev)[None], downscale_factor).cpu().numpy()
            for (path, _), out in zip(pending, lr):
                name = os.path.splitext(os.path.basename(path))[0]
                np.save(os.path.join(output_dir, name + suffix + ".npy"), out)
                ok += 1
        except Exception as e:  # noqa: BLE001 -- the reference prints and continues (C_30:205-209)
            print(f"failed on a batch of {len(pending)} file(s) starting at {os.path.basename(pending[0][0])}: {e}")
            bad += len(pending)
        pending = []

    for path in files:
        try:
            a = np.load(path).astype(np.float32)
            if a.ndim != 3:
                raise ValueError(f"expected [C,H,W], got {a.shape}")
        except Exception as e:  # noqa: BLE001
            print(f"failed to read {os.path.basename(path)}: {e}")
            bad += 1
            continue
        if pending and (pending[0][1].shape != a.shape or len(pending) >= batch_size):
            flush()
        pending.append((path, a))
    flush()
    print(f"done: {ok} ok, {bad} failed")                                    # C_30:211-213
    return ok, bad


def process_files(hr_dir, blurred_dir, noise_pool_path, output_dir, seed=42, hr_shape=(5, 256, 256), lr_shape=(5, 32, 32),
                  batch_size=256):
    """E_make_train_data.py:187-272 on .npy: for every name present in both folders, lr = blurred + noise_pool[idx] with
    idx drawn per file from the global NumPy RNG seeded like E:190, in sorted file order; writes output_dir/<name>_train.npz
    with arrays `hr` and `lr` (the reference writes the same two groups to NetCDF, E:77-117).  Files whose shapes fail the
    guards (E:239-247) are skipped.  At most `batch_size` files are held in host memory at a time (one fused launch per
    batch; the RNG stream is the same as one draw per file in file order whatever the batch size).  Returns (n_success, n_skipped)."""
    L.require_gpu()
    np.random.seed(seed)                                                     # E:190
    if not os.path.isdir(hr_dir) or not os.path.isdir(blurred_dir):
        raise FileNotFoundError(f"input directory missing: {hr_dir} / {blurred_dir}")     # E:194
    if not os.path.isfile(noise_pool_path):
        raise FileNotFoundError(f"noise pool missing: {noise_pool_path}")                # E:198
    pool = np.load(noise_pool_path).astype(np.float32)                       # E:201  (N,5,32,32)
    names = sorted(f for f in os.listdir(blurred_dir) if f.endswith(".npy"))
    if not names:
        raise FileNotFoundError(f"no .npy files in {blurred_dir}")                       # E:210
    os.makedirs(output_dir, exist_ok=True)
    done = skipped = 0
    pending = []

    def flush():
        nonlocal done, pending
        if pending:
            # one draw per surviving file, in file order, exactly like sequential add_noise calls (E:250 -> E:72)
            lr = add_noise_batch(np.stack([b for _, _, b in pending]), pool)
            for (f, hr, _), l in zip(pending, lr):
                np.savez(os.path.join(output_dir, os.path.splitext(f)[0] + "_train.npz"), hr=hr, lr=l)
            done += len(pending)
            pending = []

    for f in names:
        try:
            hr = np.load(os.path.join(hr_dir, f)).astype(np.float32)
            bl = np.load(os.path.join(blurred_dir, f)).astype(np.float32)
            if hr.shape != tuple(hr_shape) or bl.shape != tuple(lr_shape):               # E:239-247
                print(f"skip {f}: HR {hr.shape} / blurred {bl.shape}")
                skipped += 1
                continue
            pending.append((f, hr, bl))
            if len(pending) >= batch_size:
                flush()
        except Exception as e:  # noqa: BLE001  (E:264-267)
            print(f"failed on {f}: {e}")
            skipped += 1
    flush()
    print(f"done: {done} ok, {skipped} skipped")
    return done, skipped
