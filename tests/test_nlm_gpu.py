"""GPU tests of the NLM denoise stage against oracle/nlm.py.  PARITY UNPINNED: scikit-image / PyWavelets (the reference's
arithmetic for denoise/denoise.py:34-67) are not installed and the reference ships no fixture for this path, so the oracle
is a restatement of the published algorithm and these tests prove HIP == restatement, not HIP == scikit-image."""
import numpy as np
import pytest
import torch

import kmsr_amd
from oracle import nlm as onlm

pytestmark = pytest.mark.gpu


def _band(H, W, seed, noise=0.8):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    clean = 60 + 10 * np.sin(xx / 9.0) + 5 * np.cos(yy / 5.0) + 3 * ((xx // 16 + yy // 16) % 2)
    return clean.astype(np.float32), (clean + rs.randn(H, W) * noise).astype(np.float32)


def test_estimate_sigma_matches_oracle(gpu_device):
    for H, W, seed in [(64, 64, 0), (70, 91, 1), (256, 256, 2)]:
        _, img = _band(H, W, seed)
        got = float(kmsr_amd.estimate_sigma(torch.from_numpy(img).cuda()))
        want = onlm.estimate_sigma(img)
        assert abs(got - want) < 1e-5 * want
        assert abs(got - 0.8) < 0.08                      # and it estimates the injected noise level


@pytest.mark.parametrize("H,W,d", [(48, 64, 11), (37, 53, 11), (40, 40, 5)])
def test_nl_means_matches_oracle(H, W, d, gpu_device):
    clean, img = _band(H, W, 3)
    sig = onlm.estimate_sigma(img)
    h = 1.8 * sig
    want = onlm.nl_means_fast(img, 7, d, h, sig)
    got = kmsr_amd.nl_means(torch.from_numpy(img).cuda()[None], torch.tensor([h]), torch.tensor([sig]), 7, d)[0].cpu().numpy()
    assert np.max(np.abs(got - want)) < 2e-4 * np.max(np.abs(want))
    assert np.sqrt(((got - clean) ** 2).mean()) < np.sqrt(((img - clean) ** 2).mean())            # and it does reduce the error


def test_denoise_band_float_nlm_drop_in(gpu_device):
    """denoise/denoise.py:34-67: NaN -> band mean, sigma-hat, NLM, NaN mask restored; all-NaN band returned untouched."""
    _, img = _band(64, 80, 4)
    img[10:14, 20:30] = np.nan
    got, sig = kmsr_amd.denoise_band_float_nlm(img, h_factor=1.8, verbose=False)
    want, wsig = onlm.denoise_band_float_nlm(img, h_factor=1.8)
    assert abs(sig - wsig) < 1e-5 * wsig
    assert np.array_equal(np.isnan(got), np.isnan(img))
    ok = ~np.isnan(img)
    assert np.max(np.abs(got[ok] - want[ok])) < 2e-4 * np.max(np.abs(want[ok]))
    nan_band = np.full((32, 32), np.nan, np.float32)
    out, s0 = kmsr_amd.denoise_band_float_nlm(nan_band, verbose=False)
    assert s0 == 0.0 and np.isnan(out).all()
    # several bands in one launch, each with its own sigma
    stack = np.stack([_band(48, 48, 5, 0.5)[1], _band(48, 48, 6, 1.2)[1]])
    outs, sigs = kmsr_amd.denoise_band_float_nlm(stack, h_factor=1.0, verbose=False)
    assert outs.shape == (2, 48, 48) and sigs[1] > 1.8 * sigs[0]
    with pytest.raises(NotImplementedError):
        kmsr_amd.nl_means(torch.zeros(1, 8, 8).cuda(), torch.ones(1), torch.ones(1), 7, 11)      # too small for the padding
