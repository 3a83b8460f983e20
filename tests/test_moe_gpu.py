"""GPU parity of SelectorNet / ContentAdaptiveDegradation / NoiseEstimator (muti_kernel/train_gemini.py:14-140,
muti_kernel/networks.py:210-224) against golden vectors produced by the reference with its shipped moe_model.pth."""
import numpy as np
import pytest
import torch

import kmsr_amd
import oracle
from conftest import assert_close, golden, landsat_like

pytestmark = pytest.mark.gpu
VB = golden("vb_cases.npz")
K = golden("kernels.npz")


def _model():
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in golden("moe_model_state.npz").items()}
    m = kmsr_amd.ContentAdaptiveDegradation(n_kernels=10, n_channels=5, kernel_size=13)
    missing, unexpected = m.load_state_dict(sd, strict=True), None      # the reference's keys match exactly
    return m.cuda().eval()


def test_state_dict_keys_match_reference_checkpoint():
    ref_keys = set(golden("moe_model_state.npz").files)
    ours = set(kmsr_amd.ContentAdaptiveDegradation().state_dict().keys())
    assert ours == ref_keys


def test_selector_logits(gpu_device):
    """conv3x3 stride 2 (MFMA) + folded eval-BN + ReLU x3 + GAP + Linear == reference logits."""
    m = _model()
    x = torch.from_numpy(landsat_like((4, 5, 256, 256), 20)).cuda()
    logits = m.selector(x).cpu().numpy()
    assert_close(logits, VB["fwd__logits"], rtol=1e-4, atol=2e-2, what="selector logits")
    assert np.array_equal(logits.argmax(1), VB["fwd__hard_weights"].argmax(1))
    # odd sizes: stride-2 output size floor((H-1)/2)+1 at every level
    x2 = torch.from_numpy(landsat_like((2, 5, 131, 77), 21)).cuda()
    sd = {k: np.asarray(v) for k, v in golden("moe_model_state.npz").items()}
    assert_close(m.selector(x2).cpu().numpy(), oracle.selector_logits(sd, x2.cpu().numpy()), rtol=1e-4, atol=2e-2,
                 what="selector logits odd size")


def test_effective_banks(gpu_device):
    m = _model()
    assert_close(m.get_effective_kernels().detach().cpu().numpy(), K["moe_kernels"], rtol=1e-5, what="softmax kernels")
    assert_close(m.get_effective_sigmas().detach().cpu().numpy(), K["moe_sigmas"], rtol=1e-6, what="softplus sigmas")


def test_forward_hard(gpu_device):
    """hard=True: the selection is the reference's (kernel 7 for all four samples), the clean part equals the
    reference's grouped conv + ::4, and the noise is the build's Philox stream scaled by the selected sigmas."""
    m = _model()
    m.noise_seed, m.noise_offset = 99, 5
    xn = landsat_like((4, 5, 256, 256), 20)
    x = torch.from_numpy(xn).cuda()
    torch.manual_seed(123)
    out, weights, vk = m(x, temp=1.0, hard=True)
    assert tuple(out.shape) == (4, 5, 64, 64) and tuple(weights.shape) == (4, 10) and tuple(vk.shape) == (10, 5, 13, 13)
    kid = weights.argmax(1).cpu().numpy()
    assert np.array_equal(kid, VB["fwd__hard_weights"].argmax(1))
    assert np.array_equal(weights.cpu().numpy(), VB["fwd__hard_weights"])
    clean = oracle.vb_blur_stride(xn, K["moe_kernels"][kid], 4)
    want = oracle.add_gaussian_noise(clean, K["moe_sigmas"][kid].reshape(-1), seed=99, offset=5)
    assert_close(out.cpu().numpy(), want, rtol=2e-5, atol=5e-6, what="hard forward")
    assert m.noise_offset == 6
    # same statistics as the reference's torch-noise output
    ref_resid = VB["fwd__hard_out"] - clean
    our_resid = out.cpu().numpy() - clean
    assert np.all(np.abs(our_resid.std(axis=(2, 3)) / ref_resid.std(axis=(2, 3)) - 1) < 0.06)


def test_forward_soft_given_weights(gpu_device):
    """hard=False with the reference's Gumbel-softmax weights injected: einsum-mixed kernels / sigmas (:112-115)."""
    m = _model()
    m.noise_seed, m.noise_offset = 7, 0
    xn = landsat_like((4, 5, 256, 256), 20)
    w = torch.from_numpy(VB["fwd__soft_weights"]).cuda()
    out, weights, _ = m(torch.from_numpy(xn).cuda(), weights=w)
    want = oracle.add_gaussian_noise(VB["fwd__soft_clean"], VB["fwd__soft_sigmas"].reshape(-1), seed=7, offset=0)
    assert_close(out.cpu().numpy(), want, rtol=3e-5, atol=5e-6, what="soft forward")


def test_forward_draws_gumbel(gpu_device):
    m = _model()
    x = torch.from_numpy(landsat_like((3, 5, 64, 64), 22)).cuda()
    out, w, _ = m(x, temp=5.0, hard=False)
    assert torch.allclose(w.sum(1), torch.ones(3, device="cuda"), atol=1e-5) and tuple(out.shape) == (3, 5, 16, 16)
    out, w, _ = m(x, temp=0.5, hard=True)
    assert bool(((w == 0) | (w == 1)).all())


def test_noise_estimator(gpu_device):
    """muti_kernel/networks.py:221-224"""
    ne = kmsr_amd.NoiseEstimator(channels=5, init_sigma=0.3, sigma_max=1.2).cuda()
    ne.noise_seed = 3
    x = torch.from_numpy(landsat_like((2, 5, 32, 32), 23)).cuda()
    noisy, sigma = ne(x)
    assert_close(sigma.cpu().numpy(), golden("kernelgan.npz")["muti__noise_sigma"], rtol=1e-6, what="sigma")
    want = oracle.add_gaussian_noise(x.cpu().numpy(), np.tile(sigma.cpu().numpy(), 2), seed=3, offset=0)
    assert_close(noisy.cpu().numpy(), want, rtol=1e-6, atol=2e-6, what="noise estimator")
    ne2 = kmsr_amd.NoiseEstimator(channels=5, init_sigma=5.0, sigma_max=0.2)
    assert float(torch.clamp(torch.exp(ne2.log_sigma.detach()), 1e-5, 0.2).max()) == pytest.approx(0.2)
