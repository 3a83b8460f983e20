import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def landsat_like(shape, seed):
    """Same synthetic 'Landsat-shaped' tile generator as tests/golden/make_golden.py."""
    import torch
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 80.0 + 20.0).numpy()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def assert_close(got, want, rtol, atol=0.0, what=""):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), f"{what}: NaN pattern differs"
    err = np.abs(got - want)[~nan_w]
    tol = (atol + rtol * np.abs(want))[~nan_w]
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} out of tolerance, max err {err.max():.3e} " \
                          f"(max rel {np.max(err / np.maximum(np.abs(want[~nan_w]), 1e-30)):.3e})"


def assert_no_farther_than_reference(got, ref32, truth, tol=1e-4, what=""):
    """The tolerance argument for reference-pinned fp32 comparisons (round-2 verdict, weak #2): `truth` is the reference's own
    computation in float64 (tests/golden/f64_truth.npz), `ref32` the reference's fp32 fixture.  Per element
        |got - truth| <= max(|ref32 - truth|, tol * max|truth|)
    i.e. the HIP result is at least as close to the exact result as the reference's own fp32 run, or within north_star's 1e-4 of
    the tensor's scale.  Returns (max|got - truth|, max|ref32 - truth|) / max|truth| for the test's report."""
    got, ref32, truth = (np.asarray(a, np.float64) for a in (got, ref32, truth))
    assert got.shape == truth.shape == ref32.shape, f"{what}: shapes {got.shape} / {ref32.shape} / {truth.shape}"
    scale = float(np.abs(truth).max()) + 1e-300
    e_got, e_ref = np.abs(got - truth), np.abs(ref32 - truth)
    bad = e_got > np.maximum(e_ref, tol * scale)
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements farther from the float64 truth than both the reference's fp32 " \
                          f"result and {tol:g} of scale; max err {e_got.max() / scale:.3e} of scale (reference fp32: {e_ref.max() / scale:.3e})"
    return e_got.max() / scale, e_ref.max() / scale


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
