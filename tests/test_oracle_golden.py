"""Pin the CPU oracle (oracle/) against golden vectors produced by the REFERENCE itself
(tests/golden/make_golden.py).  CPU-only; these tests are what make the oracle trustworthy
before any HIP kernel is compared with it.

Tolerances: the oracle follows the reference's order of operations but not ATen's internal
summation order (oneDNN blocking), so fp32 results agree to a few ulp, not bitwise:
rtol 2e-6 on O(60) radiances for the 169-tap blur.
"""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, assert_close, golden, landsat_like

VA = golden("va_cases.npz")
VA_NAMES = sorted({k.split("__")[0] for k in VA.files if "__" in k and not k.startswith("nan_case")})


@pytest.mark.parametrize("name", VA_NAMES)
def test_va_apply_kernel_degradation(name):
    """C_30apply_kernel_to_landsat.py:68-124"""
    C, H, W, f, seed = VA[name + "__meta"]
    img = landsat_like((C, H, W), int(seed))
    assert abs(img.astype(np.float64).sum() - VA[name + "__in_sum"][0]) < 1e-3, "input regeneration mismatch"
    out = oracle.apply_kernel_degradation(img, VA[name + "__kernel"], int(f))
    assert_close(out, VA[name + "__out"], rtol=2e-6, what=name)


def test_va_nan_propagation():
    out = oracle.apply_kernel_degradation(VA["nan_case__in"], VA["nan_case__kernel"], 4)
    want = VA["nan_case__out"]
    assert np.isnan(want).sum() == 16            # ceil(13/4)^2 outputs of band 1 only
    assert not np.isnan(want[[0, 2, 3, 4]]).any()
    assert_close(out, want, rtol=2e-6, what="nan_case")


def test_va_band_mismatch_raises():
    with pytest.raises(AssertionError):           # C_30:88
        oracle.apply_kernel_degradation(np.zeros((3, 16, 16), np.float32), np.ones((5, 13, 13), np.float32), 4)
    with pytest.raises(ValueError):               # C_31:69
        oracle.apply_kernel_degradation(np.zeros((3, 16, 16), np.float32), np.ones((1, 3, 13, 13), np.float32), 4)


VB = golden("vb_cases.npz")
K = golden("kernels.npz")


@pytest.mark.parametrize("name", ["b4_256", "b3_250x253", "b2_64"])
def test_vb_blur_stride(name):
    """train_gemini.py:124-134"""
    B, C, H, W, seed = VB[name + "__meta"]
    x = landsat_like((B, C, H, W), int(seed))
    assert abs(x.astype(np.float64).sum() - VB[name + "__in_sum"][0]) < 1e-2
    out = oracle.vb_blur_stride(x, K["moe_kernels"][VB[name + "__kid"]], 4)
    assert_close(out, VB[name + "__out"], rtol=2e-6, what=name)


def test_vb_peaked_kernels():
    x = landsat_like((2, 5, 96, 96), 23)
    out = oracle.vb_blur_stride(x, VB["kgan_b2_96__kernels"], 4)
    assert_close(out, VB["kgan_b2_96__out"], rtol=2e-6, what="kgan_b2_96")


def test_vb_bank_reparametrisation():
    """train_gemini.py:68-86 and the shipped kernel_i.npy / sigma_i.npy (train_gemini.py:241-249)."""
    sd = golden("moe_model_state.npz")
    vk = oracle.effective_kernels(sd["kernel_bank"])
    vs = oracle.effective_sigmas(sd["sigma_bank"])
    assert_close(vk, VB["fwd__valid_kernels"], rtol=1e-5, what="softmax kernels")
    assert_close(vs, VB["fwd__valid_sigmas"], rtol=1e-6, what="softplus sigmas")
    assert_close(vk, K["moe_kernels"], rtol=1e-5, what="kernel_i.npy")
    assert_close(vs, K["moe_sigmas"], rtol=1e-6, what="sigma_i.npy")


def test_selector_logits():
    """SelectorNet, train_gemini.py:14-39, eval mode, weights from moe_model.pth."""
    sd = golden("moe_model_state.npz")
    x = landsat_like((4, 5, 256, 256), 20)
    logits = oracle.selector_logits(sd, x)
    # logits are O(100..500) sums over 576-deep dot products of O(60) radiances
    assert_close(logits, VB["fwd__logits"], rtol=2e-5, atol=2e-3, what="selector logits")
    assert np.array_equal(logits.argmax(1), VB["fwd__hard_weights"].argmax(1))


def test_vb_soft_mixture():
    """train_gemini.py:112-134 with soft weights: einsum-mixed kernels then blur + ::4."""
    x = landsat_like((4, 5, 256, 256), 20)
    w = VB["fwd__soft_weights"]
    bk = np.einsum("bk,kchw->bchw", w, VB["fwd__valid_kernels"]).astype(np.float32)
    assert_close(bk, VB["fwd__soft_kernels"], rtol=1e-5, what="mixed kernels")
    out = oracle.vb_blur_stride(x, bk, 4)
    assert_close(out, VB["fwd__soft_clean"], rtol=3e-6, what="soft clean")


def test_vb_full_forward_statistics():
    """The full forward adds torch-CPU randn noise (train_gemini.py:137-138): statistical check only."""
    x = landsat_like((4, 5, 256, 256), 20)
    kid = VB["fwd__hard_weights"].argmax(1)
    clean = oracle.vb_blur_stride(x, K["moe_kernels"][kid], 4)
    resid = VB["fwd__hard_out"] - clean
    sig = K["moe_sigmas"][kid]                                        # [4,5]
    std = resid.std(axis=(2, 3))
    assert np.all(np.abs(std / sig - 1) < 0.03)
    assert np.all(np.abs(resid.mean(axis=(2, 3))) < 4 * sig / 64)


def test_gaussian_stream_moments_and_philox_kat():
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors): ctr=0,key=0 and ctr=key=ff..f
    assert list(oracle.philox_u32(0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(oracle.philox_u32(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    eps = oracle.gaussian_eps(4, 256, 256, seed=1234, offset=0)
    assert abs(eps.mean()) < 0.01 and abs(eps.std() - 1) < 0.01
    assert abs((eps ** 3).mean()) < 0.03 and abs((eps ** 4).mean() - 3) < 0.1
    # neighbouring rows/cols/planes are uncorrelated
    for a, b in ((eps[:, 1:], eps[:, :-1]), (eps[:, :, 1:], eps[:, :, :-1]), (eps[1:], eps[:-1])):
        assert abs(np.mean(a * b)) < 0.01
    # offset / seed change the stream
    assert np.abs(eps - oracle.gaussian_eps(4, 256, 256, seed=1234, offset=1)).mean() > 0.5


RNG = golden("rng_streams.npz")


def test_add_noise_index_stream():
    """E_make_train_data.py:65-74 with np.random.seed(42) (E:190)."""
    np.random.seed(42)
    idx = [np.random.randint(0, 1000) for _ in range(32)]
    assert idx == list(RNG["add_noise_idx_seed42_N1000"])
    pool = np.random.RandomState(0).randn(1000, 5, 4, 4).astype(np.float32)
    blurred = np.random.RandomState(1).randn(32, 5, 4, 4).astype(np.float32)
    out = oracle.add_pool_noise(blurred, pool, idx)
    assert np.array_equal(out, blurred + pool[idx])


def test_random_crop_offsets():
    """D_build_noise_pool.py:41-53 with random.seed(42) (D:65): inclusive randint bounds."""
    random.seed(42)
    offs = [(random.randint(0, 256 - 32), random.randint(0, 256 - 32)) for _ in range(16)]
    assert offs == [tuple(o) for o in RNG["random_crop_offs_seed42_256"]]
    random.seed(42)
    offs = [(random.randint(0, 40 - 32), random.randint(0, 33 - 32)) for _ in range(8)]
    assert offs == [tuple(o) for o in RNG["random_crop_offs_seed42_40x33"]]
    geo = np.random.RandomState(2).randn(5, 40, 33).astype(np.float32)
    den = np.random.RandomState(3).randn(5, 40, 33).astype(np.float32)
    c = oracle.noise_crop(geo, den, 8, 1, 32)
    assert np.array_equal(c, (geo - den)[:, 8:40, 1:33])


KG = golden("kernelgan.npz")


def _sd(prefix):
    return {k[len(prefix):]: KG[k] for k in KG.files if k.startswith(prefix)}


def test_condition_encoder_scales():
    """muti_kernel/networks.py:41-58"""
    s = oracle.condition_scales(_sd("muti_sd__"), KG["muti__x"])
    assert_close(s, KG["muti__scales"], rtol=2e-6, what="scales")


def test_muti_extract_effective_kernels():
    """muti_kernel/networks.py:135-202"""
    sd = _sd("muti_sd__")
    kb = oracle.muti_extract_effective_kernels(sd, KG["muti__x"], reduce_batch=False)
    assert_close(kb, KG["muti__kernels_batch"], rtol=2e-4, atol=2e-7, what="batch kernels")
    kr = oracle.muti_extract_effective_kernels(sd, KG["muti__x"], reduce_batch=True)
    assert_close(kr, KG["muti__kernels_reduced"], rtol=2e-4, atol=2e-7, what="reduced kernels")
    kn = oracle.muti_extract_effective_kernels(sd, None)
    assert_close(kn, KG["muti__kernels_noinput"], rtol=2e-4, atol=2e-7, what="no-input kernels")


def test_muti_generator_forward():
    """muti_kernel/networks.py:101-133 (reflect padding per layer, 3x avg-pool)"""
    out = oracle.muti_generator_forward(_sd("muti_sd__"), KG["muti__x"])
    assert_close(out, KG["muti__forward"], rtol=1e-4, atol=2e-5, what="muti forward")


def test_single_extract_and_forward():
    """single_kernel/networks.py:83-157"""
    sd = _sd("single_perturbed_sd__")
    k = oracle.single_extract_effective_kernels(sd)
    assert_close(k, KG["single_perturbed__kernels"], rtol=2e-4, atol=2e-7, what="single kernels")
    x = landsat_like((2, 5, 64, 64), 31)
    out = oracle.single_generator_forward(sd, x)
    assert_close(out, KG["single_perturbed__forward"], rtol=1e-4, atol=1e-3, what="single forward")


def test_noise_estimator_sigma():
    sd = _sd("muti_sd__")
    s = oracle.noise_estimator_sigma(sd["noise_estimator.log_sigma"], 1.2)
    assert_close(s, KG["muti__noise_sigma"], rtol=1e-6, what="noise sigma")


def test_load_kernel_semantics_fixture_is_consistent():
    """C_31:22-37: 4-D -> mean over batch (after float32 cast); 2-D -> [1,k,k] (C_31) / unchanged (C_30)."""
    lk = golden("load_kernel.npz")
    assert_close(lk["batch_in"].astype(np.float32).mean(axis=0), lk["batch_out"], rtol=1e-6)
    assert lk["k2d_out_c31"].shape == (1, 13, 13) and lk["k2d_out_c30"].shape == (13, 13)


def test_vb_backward_matches_reference_autograd():
    """oracle.vb_backward (per-sample dK) chained through einsum + softmax in float64 == the reference's own
    kernel_bank.grad for L = sum(out * G) (train_gemini.py:88-140 under backward; fixture moe_backward.npz)."""
    import torch
    MB = golden("moe_backward.npz")
    B, C, H, W, sx, sg, _ = (int(v) for v in MB["meta"])
    x = landsat_like((B, C, H, W), sx)
    G = torch.randn(B, C, -(-H // 4), -(-W // 4), generator=torch.Generator().manual_seed(sg)).numpy()
    dk = torch.from_numpy(oracle.vb_backward(x, G, 13, 4, "zero"))                    # [B,C,13,13]
    bank = torch.from_numpy(np.asarray(golden("moe_model_state.npz")["kernel_bank"])).double().requires_grad_(True)
    vk = torch.softmax(bank.view(10, 5, -1), dim=-1).view(10, 5, 13, 13)
    bk = torch.einsum("bk,kchw->bchw", torch.from_numpy(MB["weights"]).double(), vk)
    (bk * dk).sum().backward()
    assert_close(bank.grad.numpy(), MB["d_kernel_bank"], rtol=2e-4, atol=1e-6 * np.abs(MB["d_kernel_bank"]).max(),
                 what="d kernel_bank")



def test_moe_trainer_losses_match_reference():
    """kmsr_amd.train_moe losses == muti_kernel/loss.py:6-41 on the shipped kernel_0 and on seeded tensors."""
    import torch
    import kmsr_amd
    MB = golden("moe_backward.npz")
    ks = np.asarray(golden("kernels.npz")["moe_kernels"])[0]
    got = [float(kmsr_amd.kernel_regularization(torch.from_numpy(ks[c]), alpha=0.5)) for c in range(5)]
    assert_close(np.array(got), MB["kreg_kernel0"], rtol=1e-5, atol=1e-6, what="kernel_regularization")
    assert_close(np.array([float(kmsr_amd.kernel_regularization(torch.from_numpy(ks[0] - 0.004)))]), MB["kreg_neg"],
                 rtol=1e-5, atol=1e-6, what="kernel_regularization with negative taps")
    pr = torch.randn(2, 1, 8, 8, generator=torch.Generator().manual_seed(34))
    pf = torch.randn(2, 1, 8, 8, generator=torch.Generator().manual_seed(35))
    got = np.array([float(kmsr_amd.lsgan_d_loss(pr, pf)), float(kmsr_amd.lsgan_g_loss(pf))])
    assert_close(got, MB["lsgan"], rtol=1e-6, atol=1e-7, what="lsgan losses")


def test_patch_discriminator_state_dict_keys():
    """PatchDiscriminator (muti_kernel/networks.py:250-268): the reference's state_dict loads key for key.  Its convolutions
    are HIP-only now (kmsr_convk_*); outputs / gradients are compared on the GPU in tests/test_kernelgan_train_gpu.py."""
    import torch
    import kmsr_amd
    MB = golden("moe_backward.npz")
    sd = {k[3:]: torch.from_numpy(np.asarray(MB[k])) for k in MB.files if k.startswith("D__")}
    D = kmsr_amd.PatchDiscriminator(in_ch=5)
    assert set(D.state_dict().keys()) == set(sd.keys())
    D.load_state_dict(sd, strict=True)


# ------------------------------------------------------------------ stock-torch form of the degradation (bench.py's second CPU baseline)
@pytest.mark.parametrize("name", VA_NAMES)
def test_va_torch_ops_form(name):
    """oracle/torch_ops.va_torch (F.pad -> F.conv2d -> F.avg_pool2d, C_30:107-122) against the same reference fixtures."""
    from oracle.torch_ops import va_torch
    C, H, W, f, seed = VA[name + "__meta"]
    out = va_torch(landsat_like((C, H, W), int(seed)), VA[name + "__kernel"], int(f)).numpy()
    assert_close(out, VA[name + "__out"], rtol=2e-6, what=name)


@pytest.mark.parametrize("name", ["b4_256", "b3_250x253", "b2_64"])
def test_vb_torch_ops_form(name):
    """oracle/torch_ops.vb_torch (grouped F.conv2d, padding 6, then [::4, ::4]; train_gemini.py:124-134)."""
    from oracle.torch_ops import vb_torch
    B, C, H, W, seed = VB[name + "__meta"]
    out = vb_torch(landsat_like((B, C, H, W), int(seed)), K["moe_kernels"][VB[name + "__kid"]], 4).numpy()
    assert_close(out, VB[name + "__out"], rtol=2e-6, what=name)


def test_oracle_vs_float64_truth():
    """The CPU restatement against the reference's own classes run in float64 (tests/golden/f64_truth.npz): composed kernels and
    generator forward no farther from the exact result than the reference's fp32 run or 1e-4 of scale, element by element -- the
    fp32 fixture comparisons above use element-wise relative tolerances up to 3e-4 because composed kernels span 4 decades."""
    from conftest import assert_no_farther_than_reference
    KG, T = golden("kernelgan.npz"), golden("f64_truth.npz")
    sd = {k[len("muti_sd__"):]: np.asarray(v) for k, v in KG.items() if k.startswith("muti_sd__")}
    x = KG["muti__x"]
    for got, key in ((oracle.muti_extract_effective_kernels(sd, x, reduce_batch=False), "muti__kernels_batch"),
                     (oracle.muti_extract_effective_kernels(sd, None), "muti__kernels_noinput"),
                     (oracle.muti_generator_forward(sd, x), "muti__forward")):
        e, er = assert_no_farther_than_reference(got, KG[key], T[key], 1e-4, key)
        assert e < 5e-5, (key, e, er)
    ssd = {k[len("single_perturbed_sd__"):]: np.asarray(v) for k, v in KG.items() if k.startswith("single_perturbed_sd__")}
    e, er = assert_no_farther_than_reference(oracle.single_extract_effective_kernels(ssd), KG["single_perturbed__kernels"],
                                             T["single_perturbed__kernels"], 1e-4, "single kernels")
    assert e < 5e-5, ("single", e, er)


@pytest.mark.skipif(not os.path.isdir("/root/reference/kernel_from_lr_gan"), reason="reference tree only exists in the build container")
def test_make_golden_reproduces_fixtures(tmp_path):
    """The documented recipe `python tests/golden/make_golden.py` (no flags: every generator, in dependency order) runs to the
    end and reproduces every committed fixture array bit for bit (round-3 verdict: the no-flag form used to die with a NameError)."""
    script = os.path.join(GOLDEN, "make_golden.py")
    r = subprocess.run([sys.executable, script, "--out", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert committed and sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) == committed
    for f in committed:
        a, b = np.load(os.path.join(GOLDEN, f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), f"{f}:{k} differs"


def test_unshuffled_conv_channel_sums_closed_form_cpu():
    """sr.unshuffled_conv_channel_sums (round 4: the F -> 4F layer's bias gradient without a pass over the HR-sized gradient map): the per-(channel,
    row parity, column parity) sums of a zero-padded 3x3 convolution from parity / border sums of its INPUT, against a float64 convolution --
    even sizes, several images, the borders that drop a row / column per tap."""
    import torch
    from kmsr_amd.sr import unshuffled_conv_channel_sums
    g = torch.Generator().manual_seed(11)
    for (N, C, H, W, Co) in ((3, 5, 12, 16, 7), (1, 2, 2, 4, 3), (2, 5, 6, 8, 64)):
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(Co, C, 3, 3, generator=g)
        y = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
        ref = torch.stack([y[:, :, p::2, q::2].sum((0, 2, 3)) for p in range(2) for q in range(2)], 1).reshape(-1)      # [4 co + 2 p + q]
        got = unshuffled_conv_channel_sums(x, w)
        assert got.shape == (4 * Co,)
        assert float((got.double() - ref).abs().max()) <= 2e-6 * float(y.abs().sum((0, 2, 3)).max()), (N, C, H, W, Co)


def test_stride2_backward_index_maps_cpu():
    """Host logic of sr.Conv3x3S2Function (round 4), checked with stock torch on the CPU: the data gradient of a stride-2 pad-1 conv is
    PixelShuffle2 of a stride-1 conv of the output gradient with the re-indexed weights s2_dgrad_weights(w), and its weight gradient is a
    gather (s2_wgrad_gather) from the stride-1 weight gradient between PixelUnshuffle2(x) and the output gradient -- odd and even sizes."""
    import torch
    import torch.nn.functional as F
    from kmsr_amd.sr import s2_dgrad_weights, s2_wgrad_gather
    torch.manual_seed(0)
    for (N, Cin, H, W, Cout) in [(2, 3, 8, 12, 5), (1, 4, 9, 7, 6), (2, 2, 5, 5, 3), (1, 1, 2, 2, 1)]:
        x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
        w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w, stride=2, padding=1)
        g = torch.randn_like(y)
        y.backward(g)
        Ho, Wo = y.shape[2:]
        dx = F.pixel_shuffle(F.conv2d(g, s2_dgrad_weights(w.detach()), padding=1), 2)[:, :, :H, :W]
        xu = F.pixel_unshuffle(F.pad(x.detach(), (0, 2 * Wo - W, 0, 2 * Ho - H)), 2)
        w1 = torch.zeros(Cout, 4 * Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xu, w1, padding=1).backward(g)
        assert float((dx - x.grad).abs().max()) < 1e-12 and float((s2_wgrad_gather(w1.grad, Cin) - w.grad).abs().max()) < 1e-12
