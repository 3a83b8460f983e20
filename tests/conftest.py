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


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
