"""GPU tests of the water mask + patch cutter against golden vectors recorded from the reference's own
apply_water_mask / create_patches_nc (kernel_from_lr_gan/A_00_patch_cutter_universal.py:89-197).  Integer work: bit-exact."""
import os
import sys

import numpy as np
import pytest

import kmsr_amd
from conftest import golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
pytestmark = pytest.mark.gpu
G = golden("patch_cutter.npz")


def _scene():
    rs = np.random.RandomState(0)          # == tests/golden/make_golden.py: patch_cutter_scene(seed 0)
    data = (rs.rand(5, 700, 900) * 6.0 + 0.5).astype(np.float32)
    data[4, 300:380, 100:400] = 50.0
    data[4, 20:25, 600:640] = 0.0
    data[:, 100:140, 200:260] = -9999.0
    data[2, 600:601, 850:851] = -9999.0
    return data


def test_water_mask_matches_reference(gpu_device):
    data = _scene()
    keep = data.copy()
    masked, total_valid, water = kmsr_amd.apply_water_mask(data)
    assert np.array_equal(data, keep)                                        # input untouched
    assert int(np.isnan(masked).sum()) == int(G["masked_nan_count"][0])
    assert np.array_equal(np.isnan(masked).sum(axis=(0, 2)), G["masked_nan_rowsum"])
    assert abs(float(np.nansum(masked.astype(np.float64))) - float(G["masked_sum"][0])) < 1e-6 * abs(float(G["masked_sum"][0]))
    nir = data[4].copy()
    nir[nir == -9999.0] = np.nan
    assert total_valid == int((~np.isnan(nir)).sum()) and water == int(((nir >= 1e-6) & (nir <= 7.0)).sum())
    ok = ~np.isnan(masked)
    assert np.array_equal(masked[ok], data[ok])                              # surviving values are bit-identical


@pytest.mark.parametrize("tag", ["thr0", "thr002", "ps128"])
def test_create_patches_matches_reference(tag, gpu_device):
    thr, ps, sr = G[tag + "__args"]
    masked, _, _ = kmsr_amd.apply_water_mask(_scene())
    r = kmsr_amd.create_patches(masked, int(ps), float(sr), float(thr))
    tot, kept = (int(v) for v in G[tag + "__total_kept"])
    assert (r["total"], r["kept"]) == (tot, kept)
    rec = G[tag + "__rec"]                                                    # (i, j, h_start, w_start, nan_count) per kept patch
    assert np.array_equal(r["index"], rec[:, :4]) and np.array_equal(r["nan_counts"], rec[:, 4])
    sums = np.array([np.nansum(p.astype(np.float64)) for p in r["patches"]])
    assert np.allclose(sums, G[tag + "__sums"], rtol=1e-12)
    for (i, j, hs, ws), p in zip(r["index"][:3], r["patches"][:3]):
        assert np.array_equal(np.nan_to_num(p, nan=-1.0), np.nan_to_num(masked[:, hs:hs + int(ps), ws:ws + int(ps)], nan=-1.0))


def test_patch_edge_cases(gpu_device):
    small = np.zeros((5, 100, 300), np.float32)
    r = kmsr_amd.create_patches(small, 256, 0.5, 0.0)                         # image smaller than a patch: empty grid
    assert r["total"] == 0 and r["kept"] == 0 and r["patches"].shape == (0, 5, 256, 256)
    exact = np.ones((5, 256, 256), np.float32)
    r = kmsr_amd.create_patches(exact, 256, 0.5, 0.0)
    assert (r["total"], r["kept"]) == (1, 1) and np.array_equal(r["patches"][0], exact)
    exact[3, 17, 200] = np.nan                                               # one NaN: dropped at threshold 0, kept at 1e-5
    assert kmsr_amd.create_patches(exact, 256, 0.5, 0.0)["kept"] == 0
    assert kmsr_amd.create_patches(exact, 256, 0.5, 1e-5)["kept"] == 1
