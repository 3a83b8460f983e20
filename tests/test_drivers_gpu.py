"""GPU tests of the batched .npy drivers (counterparts of C_30:127-213 / C_31:136-195 and E_make_train_data.py:187-272)."""
import os

import numpy as np
import pytest

import kmsr_amd
import oracle
from conftest import assert_close, golden, landsat_like

pytestmark = pytest.mark.gpu
K = golden("kernels.npz")


def test_process_landsat_folder(tmp_path, gpu_device):
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    patches = {f"p{i:02d}": landsat_like((5, 64, 64), 200 + i) for i in range(5)}
    patches["q_other_shape"] = landsat_like((5, 32, 48), 300)
    for k, v in patches.items():
        np.save(ind / f"{k}.npy", v)
    np.save(ind / "broken.npy", np.zeros((4, 4), np.float32))                # wrong rank -> skipped, loop continues
    kp = tmp_path / "kernel_per_band.npy"
    np.save(kp, K["kgan_per_band"])
    ok, bad = kmsr_amd.process_landsat_folder(str(ind), str(kp), str(outd), downscale_factor=8, batch_size=2)
    assert (ok, bad) == (6, 1)
    for k, v in patches.items():
        got = np.load(outd / f"{k}_lr.npy")
        assert_close(got, oracle.apply_kernel_degradation(v, K["kgan_per_band"], 8), rtol=2e-5, what=k)
    # a 2-D merged kernel is broadcast to all bands (C_30:83-85)
    np.save(kp, K["kgan_merged"])
    ok, bad = kmsr_amd.process_landsat_folder(str(ind), str(kp), str(tmp_path / "out2"), downscale_factor=4)
    assert ok == 6
    assert_close(np.load(tmp_path / "out2" / "p00_lr.npy"), oracle.apply_kernel_degradation(patches["p00"], K["kgan_merged"], 4),
                 rtol=2e-5, what="2-D kernel")
    # band mismatch: the whole batch is reported as failed, nothing raises
    np.save(kp, K["kgan_per_band"][:3])
    ok, bad = kmsr_amd.process_landsat_folder(str(ind), str(kp), str(tmp_path / "out3"))
    assert ok == 0 and bad == 7


def test_process_files(tmp_path, gpu_device):
    hrd, bld, outd = tmp_path / "hr", tmp_path / "blurred", tmp_path / "train"
    hrd.mkdir()
    bld.mkdir()
    rs = np.random.RandomState(0)
    pool = rs.randn(100, 5, 32, 32).astype(np.float32)
    np.save(tmp_path / "pool.npy", pool)
    names = [f"s{i}.npy" for i in range(4)]
    data = {}
    for n in names:
        data[n] = (rs.rand(5, 256, 256).astype(np.float32), rs.rand(5, 32, 32).astype(np.float32))
        np.save(hrd / n, data[n][0])
        np.save(bld / n, data[n][1])
    np.save(hrd / "bad.npy", rs.rand(5, 128, 128).astype(np.float32))       # fails the shape guard (E:239-247)
    np.save(bld / "bad.npy", rs.rand(5, 32, 32).astype(np.float32))
    ok, skipped = kmsr_amd.process_files(str(hrd), str(bld), str(tmp_path / "pool.npy"), str(outd), seed=42)
    assert (ok, skipped) == (4, 1)
    np.random.seed(42)
    want_idx = [np.random.randint(0, 100) for _ in range(4)]                 # one draw per surviving file, sorted order
    for n, idx in zip(sorted(names), want_idx):
        z = np.load(outd / (n[:-4] + "_train.npz"))
        assert np.array_equal(z["hr"], data[n][0])
        assert np.array_equal(z["lr"], data[n][1] + pool[idx])
    # bounded host buffering: batches of 3 files give byte-identical outputs (same RNG stream, one draw per file in order)
    out2 = tmp_path / "train_b3"
    assert kmsr_amd.process_files(str(hrd), str(bld), str(tmp_path / "pool.npy"), str(out2), seed=42, batch_size=3) == (4, 1)
    for n in names:
        a, b = np.load(outd / (n[:-4] + "_train.npz")), np.load(out2 / (n[:-4] + "_train.npz"))
        assert np.array_equal(a["lr"], b["lr"]) and np.array_equal(a["hr"], b["hr"])
    with pytest.raises(FileNotFoundError):
        kmsr_amd.process_files(str(hrd), str(bld), str(tmp_path / "nope.npy"), str(outd))
