"""GPU parity of the KernelGAN kernel-composition path (BASELINE config 5) against golden vectors produced by the
reference's own extract_effective_kernels (muti_kernel/networks.py:135-202, single_kernel/networks.py:108-157)."""
import numpy as np
import pytest
import torch

import kmsr_amd
import oracle
from conftest import assert_close, assert_no_farther_than_reference, golden, landsat_like

pytestmark = pytest.mark.gpu
KG = golden("kernelgan.npz")
F64 = golden("f64_truth.npz")        # the reference's own classes in double on the same inputs (make_golden.py --f64-truth)


@pytest.fixture(autouse=True)
def _inference_mode():
    """These are the inference launches (the reference calls them under torch.no_grad(), muti train.py:118, networks.py:135);
    the differentiable path is tested in test_kernelgan_train_gpu.py."""
    with torch.no_grad():
        yield


def _sd(prefix):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(KG[k])) for k in KG.files if k.startswith(prefix)}


def _muti():
    g = kmsr_amd.DynamicMultiBandLinearGenerator(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1)
    sd = {k[len("generator."):]: v for k, v in _sd("muti_sd__").items() if k.startswith("generator.")}
    g.load_state_dict(sd, strict=True)          # the reference's parameter names load unchanged
    return g.cuda()


def test_condition_encoder_scales(gpu_device):
    g = _muti()
    x = torch.from_numpy(KG["muti__x"]).cuda()
    s = g.condition_encoder.scales_tensor(x)
    assert_close(s.cpu().numpy(), KG["muti__scales"], rtol=5e-6, what="condition scales")
    lst = g.condition_encoder(x)                 # reference return type: list[band][layer] of [B,out_c]
    assert len(lst) == 5 and len(lst[0]) == 6 and tuple(lst[2][5].shape) == (2, 1) and tuple(lst[4][0].shape) == (2, 32)


def test_muti_extract_effective_kernels(gpu_device):
    g = _muti()
    x = torch.from_numpy(KG["muti__x"]).cuda()
    kb = g.extract_effective_kernels(x, reduce_batch=False)
    assert tuple(kb.shape) == (2, 5, 13, 13)
    assert_close(kb.cpu().numpy(), KG["muti__kernels_batch"], rtol=3e-4, atol=3e-7, what="batch kernels")
    assert_close(g.extract_effective_kernels(x).cpu().numpy(), KG["muti__kernels_reduced"], rtol=3e-4, atol=3e-7, what="reduced")
    assert_close(g.extract_effective_kernels(None).cpu().numpy(), KG["muti__kernels_noinput"], rtol=3e-4, atol=3e-7, what="no input")
    assert_close(g.extract_merged_kernel(x).cpu().numpy(), KG["muti__kernels_reduced"].mean(0), rtol=3e-4, atol=3e-7, what="merged")
    s = kb.sum(dim=(2, 3)).cpu().numpy()
    assert np.allclose(s, 1.0, atol=1e-5) and float(kb.min()) >= 0.0      # invariant: non-negative, sums to 1
    # the 3e-4 above is ELEMENT-wise relative (taps down to 1e-4 of the peak); against the float64 truth the HIP kernels are no
    # farther than the reference's fp32 run or 1e-4 of the kernel's peak, element by element
    for got, key in ((kb, "muti__kernels_batch"), (g.extract_effective_kernels(x), "muti__kernels_reduced"),
                     (g.extract_effective_kernels(None), "muti__kernels_noinput")):
        e, er = assert_no_farther_than_reference(got.cpu().numpy(), KG[key], F64[key], 1e-4, key)
        assert e < 2e-5, f"{key}: {e:.2e} of scale from the float64 truth (reference fp32: {er:.2e})"


def test_single_extract_effective_kernels(gpu_device):
    g = kmsr_amd.MultiBandLinearGenerator(in_ch=5, mid_ch=32).cuda()
    # default Gaussian init (single_kernel/networks.py:55-81): 7x7 Gaussian sigma=2 centred in 13x13
    assert_close(g.extract_effective_kernels().cpu().numpy(), KG["single_default__kernels"], rtol=1e-5, atol=1e-9, what="default init")
    g.load_state_dict(_sd("single_perturbed_sd__"), strict=True)
    assert_close(g.extract_effective_kernels().cpu().numpy(), KG["single_perturbed__kernels"], rtol=3e-4, atol=3e-7, what="perturbed")
    e, er = assert_no_farther_than_reference(g.extract_effective_kernels().cpu().numpy(), KG["single_perturbed__kernels"],
                                             F64["single_perturbed__kernels"], 1e-4, "single perturbed kernels vs float64")
    assert e < 2e-5, f"{e:.2e} of scale from the float64 truth (reference fp32: {er:.2e})"


def test_compose_against_oracle_other_shapes(gpu_device):
    """Shapes the reference never uses (3 layers, 16 channels, 5/3/3 kernels) against the C oracle."""
    rs = np.random.RandomState(0)
    co, ci, ks = [16, 8, 2], [1, 16, 8], [5, 3, 3]
    Ws = [[rs.randn(o, i, k, k).astype(np.float32) * 0.3 for o, i, k in zip(co, ci, ks)] for _ in range(3)]
    sc = rs.rand(2, 3, sum(co)).astype(np.float32) + 0.5
    flat = torch.from_numpy(np.stack([np.concatenate([w.reshape(-1) for w in band]) for band in Ws])).cuda()
    got = kmsr_amd.compose_kernels(flat, torch.from_numpy(sc).cuda(), co, ci, ks).cpu().numpy()
    for b in range(2):
        for c in range(3):
            off = np.cumsum([0] + co)
            want = oracle.compose_kernel(Ws[c], [sc[b, c, off[l]:off[l + 1]] for l in range(3)])
            assert_close(got[b, c], want, rtol=3e-4, atol=3e-7, what=f"compose b{b} c{c}")
    with pytest.raises(AssertionError):        # channel chain mismatch -> KMSR_E_SHAPE
        kmsr_amd.compose_kernels(flat, None, co, [1, 16, 7], ks)


def test_bulk_extract_kernels_sharded(gpu_device):
    """muti_kernel/train.py:118-188 at array level; rank sharding returns disjoint, complete, world-size-invariant results."""
    g = _muti()
    patches = [landsat_like((5, 72, 80), 100 + i) / 100.0 for i in range(5)]
    full = dict(kmsr_amd.bulk_extract_kernels(patches, g, patch_size=64, batch_size=2))
    assert sorted(full) == [0, 1, 2, 3, 4] and tuple(full[0].shape) == (5, 13, 13)
    parts = {}
    for r in range(2):
        parts.update(dict(kmsr_amd.bulk_extract_kernels(patches, g, patch_size=64, batch_size=2, rank=r, world_size=2)))
    assert sorted(parts) == sorted(full)
    for i in full:
        assert torch.allclose(parts[i], full[i], rtol=1e-6, atol=1e-9)
    # centre crop (train.py:160-166) and agreement with the oracle on one patch
    sd = {k: np.asarray(v) for k, v in KG.items() if k.startswith("muti_sd__")}
    sd = {k[len("muti_sd__"):]: v for k, v in sd.items()}
    crop = patches[3][:, 4:68, 8:72]
    want = oracle.muti_extract_effective_kernels(sd, crop[None], reduce_batch=False)[0]
    assert_close(full[3].cpu().numpy(), want, rtol=3e-4, atol=3e-7, what="bulk vs oracle")
    with pytest.raises(ValueError):
        kmsr_amd.bulk_extract_kernels([np.zeros((5, 32, 32), np.float32)], g, patch_size=64)


def test_generator_forward_muti(gpu_device):
    """DynamicMultiBandLinearGenerator.forward (muti_kernel/networks.py:101-133): per-layer REFLECT padding, per-sample
    condition scales, 3 avg-pools -- against the reference's own output."""
    g = _muti()
    x = torch.from_numpy(KG["muti__x"]).cuda()
    out = g(x)
    assert tuple(out.shape) == (2, 5, 8, 8)
    assert_close(out.cpu().numpy(), KG["muti__forward"], rtol=2e-4, atol=5e-5, what="muti forward")
    e, er = assert_no_farther_than_reference(out.cpu().numpy(), KG["muti__forward"], F64["muti__forward"], 1e-4, "muti forward vs float64")
    assert e < 5e-5, f"{e:.2e} of scale from the float64 truth (reference fp32: {er:.2e})"


def test_generator_forward_single(gpu_device):
    """MultiBandLinearGenerator.forward (single_kernel/networks.py:83-106), default Gaussian init and perturbed weights."""
    x = torch.from_numpy(landsat_like((2, 5, 64, 64), 31)).cuda()
    g = kmsr_amd.MultiBandLinearGenerator(in_ch=5, mid_ch=32).cuda()
    assert_close(g(x).cpu().numpy(), KG["single_default__forward"], rtol=1e-4, atol=1e-3, what="single default forward")
    g.load_state_dict(_sd("single_perturbed_sd__"), strict=True)
    assert_close(g(x).cpu().numpy(), KG["single_perturbed__forward"], rtol=1e-4, atol=1e-3, what="single perturbed forward")
    e, er = assert_no_farther_than_reference(g(x).cpu().numpy(), KG["single_perturbed__forward"], F64["single_perturbed__forward"], 1e-4,
                                             "single perturbed forward vs float64")
    assert e < 2e-5, f"{e:.2e} of scale from the float64 truth (reference fp32: {er:.2e})"
    # odd tile coverage: a size that is not a multiple of the 16x16 workgroup tile, against the C oracle
    x2 = torch.from_numpy(landsat_like((1, 5, 40, 56), 32)).cuda()
    sd = {k: v.numpy() for k, v in _sd("single_perturbed_sd__").items()}
    assert_close(g(x2).cpu().numpy(), oracle.single_generator_forward(sd, x2.cpu().numpy()), rtol=1e-4, atol=1e-3, what="40x56")


def test_degradation_model(gpu_device):
    """DegradationModel (muti_kernel/networks.py:227-247): the reference checkpoint layout loads unchanged; clean output ==
    generator forward; noisy = clean + sigma * eps with the clamped per-band sigma."""
    m = kmsr_amd.DegradationModel(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1, noise_init=0.3, noise_max=1.2)
    m.load_state_dict(_sd("muti_sd__"), strict=True)
    m = m.cuda()
    m.noise_estimator.noise_seed = 11
    x = torch.from_numpy(KG["muti__x"]).cuda()
    clean, noisy, sigma = m(x)
    assert_close(clean.cpu().numpy(), KG["muti__forward"], rtol=2e-4, atol=5e-5, what="clean")
    assert_close(sigma.cpu().numpy(), KG["muti__noise_sigma"], rtol=1e-6, what="sigma")
    want = oracle.add_gaussian_noise(clean.cpu().numpy(), np.tile(sigma.cpu().numpy(), 2), seed=11, offset=0)
    assert_close(noisy.cpu().numpy(), want, rtol=1e-6, atol=2e-6, what="noisy")


def test_flat_weight_cache_follows_parameter_updates(gpu_device):
    """The composition kernel's flat weight operand is built once and reused (round 4: no torch.cat per call); an in-place parameter update
    (an optimizer step bumps the version counter) or load_state_dict must invalidate it."""
    torch.manual_seed(11)
    for gen, args in ((kmsr_amd.DynamicMultiBandLinearGenerator().cuda(), (None,)), (kmsr_amd.MultiBandLinearGenerator(in_ch=5, mid_ch=32).cuda(), ())):
        k0 = gen.extract_effective_kernels(*args).clone()
        flat0 = gen._flat_cache[1]
        assert gen.extract_effective_kernels(*args) is not None and gen._flat_cache[1] is flat0          # second call: the same buffer
        w = next(gen.parameters())
        with torch.no_grad():
            w.mul_(1.5)
        k1 = gen.extract_effective_kernels(*args)
        assert gen._flat_cache[1] is not flat0 and not torch.equal(k0, k1)
        ref = type(gen)().cuda() if args else type(gen)(in_ch=5, mid_ch=32).cuda()
        ref.load_state_dict(gen.state_dict())
        assert torch.equal(ref.extract_effective_kernels(*args), k1)
