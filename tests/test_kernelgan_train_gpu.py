"""KernelGAN trainers on the HIP kernels.

  * kmsr_convk_fwd / dgrad / wgrad_f32 against the stock-torch float64 form of the reference layer (oracle/torch_ops.py
    convk_torch = F.pad(reflect|zero) + F.conv2d, muti_kernel/networks.py:108-112, :257-264) through torch autograd;
  * one full iteration of each trainer against the gradients / updated weights the REFERENCE's classes produce for the same
    statements (tests/golden/kernelgan_train.npz, make_golden.py make_kernelgan_train_golden);
  * CPU: the samplers' draw order on the global torch RNG against the reference's functions.
"""
import os

import numpy as np
import pytest
import torch

import kmsr_amd
from conftest import assert_close, assert_no_farther_than_reference, golden, landsat_like
from oracle.torch_ops import convk_torch

KT = golden("kernelgan_train.npz")
F64 = golden("f64_truth.npz")
KG = golden("kernelgan.npz")


# ------------------------------------------------------------------------------------------------ CPU: samplers
def test_samplers_follow_reference_draw_order(tmp_path):
    for i in range(6):
        np.save(tmp_path / f"p{i}.npy", landsat_like((5, 40, 44), 60 + i) / 100.0)
    files, osz = kmsr_amd.load_patches_from_folder(str(tmp_path))
    assert osz == int(KT["sampler_muti_osz"][0]) and len(files) == 6
    torch.manual_seed(7)
    a = kmsr_amd.sample_patches_from_files(files, batch_size=4, target_size=32, original_size=osz)
    b = kmsr_amd.sample_patches_from_files(files, batch_size=3, target_size=40, original_size=osz)
    assert tuple(a.shape) == (4, 5, 32, 32) and tuple(b.shape) == (3, 5, 40, 44)
    assert np.array_equal(a.numpy()[:, :, ::8, ::8], KT["sampler_muti_a"])
    assert np.array_equal(b.numpy()[:, :, ::8, ::8], KT["sampler_muti_b"])
    torch.manual_seed(8)
    full = torch.stack([torch.from_numpy(landsat_like((5, 40, 44), 70 + i) / 100.0) for i in range(3)])
    assert np.array_equal(kmsr_amd.crop_patches(full, 32).numpy()[:, :, ::8, ::8], KT["sampler_single_crop"])
    with pytest.raises(ValueError):
        kmsr_amd.crop_patches(full, 41)
    bad = [np.full((5, 40, 44), np.nan, np.float32)]
    with pytest.raises(ValueError):
        kmsr_amd.sample_patches_from_files(bad, 1, 32, 40)
    with pytest.raises(ValueError):                                  # muti train.py:58-61: target must be strictly smaller
        kmsr_amd.sample_patches_from_files([np.zeros((5, 32, 44), np.float32)], 1, 32, 40)
    with pytest.raises(ValueError):
        kmsr_amd.load_patches_from_folder(str(tmp_path / "missing"))


def test_single_regulariser_and_noise_loss_formulas():
    """single_kernel/loss.py:35-96 and muti_kernel/loss.py:44-55 on hand-checkable inputs."""
    k = torch.zeros(13, 13)
    k[6, 6] = 0.6
    k[6, 7] = 0.4
    base = float(kmsr_amd.kernel_regularization(k))
    assert abs(float(kmsr_amd.kernel_regularization_single(k, epsilon=3.0)) - base) < 1e-7      # peak at the centre -> no extra term
    k2 = k.clone()
    k2[2, 3] = 0.9
    extra = float(kmsr_amd.kernel_regularization_single(k2, epsilon=3.0)) - float(kmsr_amd.kernel_regularization(k2))
    assert abs(extra - 3.0 * (0.9 - 0.6) ** 2) < 1e-6
    s = torch.tensor([0.5, 1.0])
    assert abs(float(kmsr_amd.noise_reg_loss(s, target=torch.tensor([0.25, 0.5]))) - (0.0625 + 0.25) / 2) < 1e-7
    assert abs(float(kmsr_amd.noise_reg_loss(s, target=0.5, mode="l1")) - 0.25) < 1e-7


# ------------------------------------------------------------------------------------------------ GPU: the conv family
CASES = [  # (k, Cin, Cout, pad, shared, bias, U, H, W)
    (7, 1, 32, "reflect", False, False, 3, 37, 45),
    (5, 32, 32, "reflect", False, False, 2, 37, 45),
    (3, 32, 1, "reflect", False, False, 3, 21, 50),
    (5, 32, 32, "reflect", False, False, 2, 6, 7),       # both mirrored borders reach the same pixels
    (3, 32, 1, "reflect", False, False, 2, 16, 16),
    (7, 5, 64, "zero", True, True, 3, 19, 33),
    (1, 64, 64, "zero", True, True, 2, 20, 17),
    (1, 64, 1, "zero", True, True, 2, 32, 32),
    (7, 1, 32, "zero", False, False, 2, 18, 18),
    # the 32 -> 32 paths of round 2: MFMA weight gradient (k*k = 9 and 1 column tiles), padded-domain + fold data gradient,
    # two pixels per thread in the forward / zero-padded data gradient, shared weights summed over units
    (3, 32, 32, "reflect", False, False, 2, 40, 64),
    (5, 32, 32, "zero", True, True, 2, 33, 70),
    (1, 32, 32, "reflect", False, False, 2, 17, 40),
]


@pytest.mark.gpu
@pytest.mark.parametrize("k,cin,cout,pad,shared,bias,U,H,W", CASES)
def test_convk_three_legs_against_torch(gpu_device, k, cin, cout, pad, shared, bias, U, H, W):
    g = torch.Generator().manual_seed(k * 1000 + cin * 10 + H)
    x = torch.randn(U, cin, H, W, generator=g)
    w = torch.randn(*(() if shared else (U,)), cin, k * k, cout, generator=g) / (k * cin ** 0.5)
    b = torch.randn(cout, generator=g) if bias else None
    gy = torch.randn(U, cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    ref = convk_torch(xr, wr, k, pad, br)
    ref.backward(gy.double())
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    out = kmsr_amd.ConvKFunction.apply(xg, wg, bg, k, pad)
    out.backward(gy.cuda())
    # fp32 sums of up to Cin*k*k = 3136 (forward / dgrad) or U*H*W = 3330 (wgrad) products against float64: 1e-5 of the
    # tensor's largest magnitude
    tol = 1e-5
    scale = float(ref.detach().abs().max())
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=tol * scale, what="forward")
    assert_close(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=tol * float(xr.grad.abs().max()), what="dgrad")
    assert_close(wg.grad.cpu().numpy(), wr.grad.numpy(), rtol=0, atol=tol * float(wr.grad.abs().max()), what="wgrad")
    if bias:
        assert_close(bg.grad.cpu().numpy(), br.grad.numpy(), rtol=0, atol=tol * float(br.grad.abs().max()), what="dbias")
    # fixed-order reductions: a second run is bit-identical
    assert torch.equal(kmsr_amd.convk_wgrad(x.cuda(), gy.cuda(), k, pad, shared=shared), wg.grad)


@pytest.mark.gpu
def test_convk_rejects_unsupported(gpu_device):
    x = torch.zeros(1, 3, 8, 8, device="cuda")
    with pytest.raises(NotImplementedError):
        kmsr_amd.convk_forward(x, torch.zeros(1, 3, 9, 7, device="cuda"), 3)          # Cout 7 is not instantiated
    with pytest.raises(ValueError):
        kmsr_amd.convk_forward(torch.zeros(1, 3, 2, 8, device="cuda"), torch.zeros(1, 3, 49, 32, device="cuda"), 7)   # H <= k/2
    with pytest.raises(AssertionError):
        kmsr_amd.convk_forward(x, torch.zeros(1, 4, 9, 32, device="cuda"), 3)         # Cin mismatch


# ------------------------------------------------------------------------------------------------ GPU: one trainer iteration
def _chk(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), t.pow(2).sum().item()])


def _compare_grads(named, prefix, rtol, what):
    n_full = 0
    for name, g in named:
        want = KT[f"{prefix}chk__{name}"]
        got = _chk(g)
        if want[1] < 1e-5:            # analytically zero (a conv bias in front of BatchNorm): rounding noise on both sides
            assert got[1] < 1e-4, f"{what} {name}: expected ~0, got sum|g| = {got[1]}"
            continue
        # sum of squares and sum of magnitudes pin the tensor's size; the signed sum is compared on the magnitude scale
        assert abs(got[2] - want[2]) <= 2 * rtol * want[2] + 1e-30, f"{what} {name}: sum sq {got[2]} vs {want[2]}"
        assert abs(got[1] - want[1]) <= rtol * want[1] + 1e-30, f"{what} {name}: sum abs {got[1]} vs {want[1]}"
        assert abs(got[0] - want[0]) <= rtol * want[1] + 1e-30, f"{what} {name}: sum {got[0]} vs {want[0]}"
        key = f"{prefix}full__{name}"
        if key in KT.files:
            ref = KT[key]
            assert_close(g.detach().cpu().numpy(), ref, rtol=0, atol=rtol * float(np.abs(ref).max()), what=f"{what} {name}")
            # ... and against the same iteration replayed with the reference's classes in float64 (f64_truth.npz): no farther than the
            # reference's own fp32 gradients or 1e-4 of the tensor's scale, element by element
            assert_no_farther_than_reference(g.detach().cpu().numpy(), ref, F64["train__" + key], 1e-4, f"{what} {name} vs float64")
            n_full += 1
    return n_full


def _critic():
    torch.manual_seed(0)
    D = kmsr_amd.PatchDiscriminator(in_ch=5, base_ch=64, num_blocks=4)
    MB = golden("moe_backward.npz")                       # the reference's seed-0 critic: same init, same state_dict keys
    D.load_state_dict({k[3:]: torch.from_numpy(np.asarray(MB[k])) for k in MB.files if k.startswith("D__")}, strict=True)
    return D.cuda()


@pytest.mark.gpu
def test_patch_discriminator_matches_reference(gpu_device):
    """PatchDiscriminator (muti_kernel/networks.py:250-268) on the HIP convs: eval and train-mode (spectral-norm power
    iteration + batch-statistics BN) outputs and a weight gradient against the reference's."""
    MB = golden("moe_backward.npz")
    D = _critic()
    xd = torch.randn(2, 5, 16, 16, generator=torch.Generator().manual_seed(36)).cuda()
    D.eval()
    with torch.no_grad():
        assert_close(D(xd).cpu().numpy(), MB["D_eval_out"], rtol=1e-4, atol=1e-5, what="D eval output")
    D.train()
    yd = D(xd)
    yd.pow(2).mean().backward()
    assert_close(yd.detach().cpu().numpy(), MB["D_train_out"], rtol=1e-4, atol=1e-5, what="D train output")
    assert_close(D.net[0].weight_u.cpu().numpy(), MB["D_train_u0"], rtol=1e-5, atol=1e-6, what="spectral-norm u after one step")
    assert_close(D.net[0].weight_orig.grad.cpu().numpy(), MB["D_train_d_w0"], rtol=1e-3,
                 atol=1e-4 * np.abs(MB["D_train_d_w0"]).max(), what="d weight_orig")
    with pytest.raises(kmsr_amd._lib.KmsrError):
        kmsr_amd.PatchDiscriminator(5)(torch.zeros(1, 5, 8, 8))            # no CPU fallback


@pytest.mark.gpu
def test_multi_kernel_trainer_iteration_matches_reference(gpu_device):
    """muti_kernel/train.py:262-286: D step, Adam, G step through the updated critic; tolerance 2e-4 of each tensor's scale
    (fp32 chains of 6 convs + critic + BatchNorm batch statistics on 2x8x8 maps, different summation orders)."""
    model = kmsr_amd.DegradationModel(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1, noise_init=0.3, noise_max=1.2)
    model.load_state_dict({n[len("muti_sd__"):]: torch.from_numpy(KG[n]) for n in KG.files if n.startswith("muti_sd__")}, strict=True)
    model = model.cuda()
    D = _critic()
    opt_D = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_G = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
    patches = torch.from_numpy(KG["muti__x"]).cuda()
    real_ds = torch.from_numpy(landsat_like((2, 5, 8, 8), 41) / 100.0).cuda()
    eps = torch.randn(2, 5, 8, 8, generator=torch.Generator().manual_seed(42)).cuda()
    tsig = torch.tensor([0.55, 0.72, 0.83, 0.63, 0.19], device="cuda")
    # the step function with the noise field injected (NoiseEstimator.forward(x, eps))
    fwd = model.noise_estimator.forward
    model.noise_estimator.forward = lambda x: fwd(x, eps)
    grads_D = {}
    step_D = opt_D.step
    opt_D.step = lambda: (grads_D.update({n: p.grad.clone() for n, p in D.named_parameters()}), step_D())[1]
    lD, lG, lR, lN, sigma = kmsr_amd.kernelgan_multi_step(model, D, opt_G, opt_D, patches, real_ds, tsig, 20.0)
    want = KT["muti_losses"]
    for got, w, nm in zip((lD, lG, lR, lN), want, ("loss_D", "loss_G_adv", "loss_reg", "loss_noise")):
        assert abs(float(got) - w) <= 3e-4 * abs(w), f"{nm}: {float(got)} vs {w}"
    assert_close(sigma.cpu().numpy(), KT["muti_sigma"], rtol=1e-6, what="sigma")
    assert int(KT["muti_reg_requires_grad"][0]) == 0          # the reference's regulariser is a constant (no_grad extract)
    n = _compare_grads(list(grads_D.items()), "muti_D_", 2e-4, "critic grad")
    assert n == 5
    n = _compare_grads([(k, p.grad) for k, p in model.named_parameters()], "muti_G_", 2e-4, "generator grad")
    assert n == 13
    # parameters after the two Adam steps
    assert_close(model.generator.weight_bands[0][0].detach().cpu().numpy(), KT["muti_after_G__weight_bands.0.0"], rtol=0, atol=2e-6, what="W after G step")
    assert_close(model.noise_estimator.log_sigma.detach().cpu().numpy(), KT["muti_after_G__log_sigma"], rtol=0, atol=2e-6, what="log_sigma after G step")
    assert_close(D.net[0].weight_orig.detach().cpu().numpy(), KT["muti_after_D__net.0.weight_orig"], rtol=0, atol=2e-6, what="critic after D step")


@pytest.mark.gpu
def test_single_kernel_trainer_iteration_matches_reference(gpu_device):
    """single_kernel/train.py:269-308 incl. clip_grad_norm_(20) on both networks."""
    G = kmsr_amd.MultiBandLinearGenerator(in_ch=5, mid_ch=32)
    G.load_state_dict({n[len("single_perturbed_sd__"):]: torch.from_numpy(KG[n]) for n in KG.files if n.startswith("single_perturbed_sd__")}, strict=True)
    G = G.cuda()
    D = _critic()
    opt_D = torch.optim.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999))
    opt_G = torch.optim.Adam(G.parameters(), lr=4e-4, betas=(0.5, 0.999))
    patches = torch.from_numpy(landsat_like((2, 5, 64, 64), 31) / 100.0).cuda()
    real_ds = torch.from_numpy(landsat_like((2, 5, 8, 8), 43) / 100.0).cuda()
    with torch.no_grad():
        assert_close(G(patches).cpu().numpy(), KT["single_fake"], rtol=0, atol=2e-5 * float(np.abs(KT["single_fake"]).max()), what="fake")
    pre = {}
    import torch.nn.utils as U
    real_clip = kmsr_amd.train_kernelgan.clip_grad_norm_

    def spy(params, max_norm):
        params = list(params)
        if len(params) == len(list(G.parameters())):
            pre.update({n: p.grad.clone() for n, p in G.named_parameters()})
        return real_clip(params, max_norm=max_norm)
    kmsr_amd.train_kernelgan.clip_grad_norm_ = spy
    try:
        lD, lG, lR, gnD, gnG = kmsr_amd.kernelgan_single_step(G, D, opt_G, opt_D, patches, real_ds, 0.002, 20.0)
    finally:
        kmsr_amd.train_kernelgan.clip_grad_norm_ = real_clip
    want = KT["single_losses"]
    for got, w, nm in zip((lD, lG, lR, gnD, gnG), want, ("loss_D", "loss_G_adv", "loss_reg", "gn_D", "gn_G")):
        assert abs(float(got) - w) <= 3e-4 * abs(w), f"{nm}: {float(got)} vs {w}"
    n = _compare_grads(list(pre.items()), "single_G_preclip_", 2e-4, "generator grad")
    assert n == 8
    assert_close(G.chains[0][0].weight.detach().cpu().numpy(), KT["single_after_G__chains.0.0.weight"], rtol=0, atol=2e-6, what="W0 after G step")
    assert_close(G.chains[1][2].weight.detach().cpu().numpy(), KT["single_after_G__chains.1.2.weight"], rtol=0, atol=2e-6, what="W2 after G step")


@pytest.mark.gpu
def test_trainers_run_and_write_reference_file_names(gpu_device, tmp_path):
    """A few iterations of both loops at the reference's patch geometry (256^2 HR, 32^2 LR): finite history, the files the
    reference writes (muti train.py:300-321, single train.py:338-353), kernels non-negative with unit sum."""
    pdir = tmp_path / "patches"
    os.makedirs(pdir)
    for i in range(4):
        np.save(pdir / f"p{i}.npy", landsat_like((5, 264, 264), 80 + i) / 100.0)
    files, osz = kmsr_amd.load_patches_from_folder(str(pdir))
    torch.manual_seed(3)
    model = kmsr_amd.DegradationModel(noise_init=0.3, noise_max=1.2).cuda()
    D = kmsr_amd.PatchDiscriminator(5).cuda()
    before = model.generator.weight_bands[2][1].detach().clone()
    h = kmsr_amd.train_kernelgan_multi(model, D, files, osz, iters=3, batch_size=2, outdir=str(tmp_path / "m"), kernel_log_every=3, log=None)
    assert h.shape == (3, 5) and np.isfinite(h).all()
    assert not torch.equal(before, model.generator.weight_bands[2][1].detach())
    for f in ("iter_kernels/batch_kernels_iter3.npy", "iter_kernels/kernel_merged_iter3.npy", "iter_kernels/kernel_per_band_iter3.npy",
              "final_results/kernel_per_band.npy", "final_results/kernel_merged.npy"):
        assert os.path.exists(tmp_path / "m" / f), f
    assert np.load(tmp_path / "m" / "iter_kernels/batch_kernels_iter3.npy").shape == (2, 5, 13, 13)
    G = kmsr_amd.MultiBandLinearGenerator(5, 32).cuda()
    D2 = kmsr_amd.PatchDiscriminator(5).cuda()
    arrays = [np.load(f)[:, :256, :256] for f in files]
    h2 = kmsr_amd.train_kernelgan_single(G, D2, arrays, iters=2, batch_size=2, outdir=str(tmp_path / "s"), kernel_log_every=2, log=None)
    assert h2.shape == (2, 5) and np.isfinite(h2).all()
    kpb = np.load(tmp_path / "s" / "kernel_per_band.npy")
    assert kpb.shape == (5, 13, 13) and kpb.min() >= 0 and np.allclose(kpb.sum(axis=(1, 2)), 1.0, atol=1e-5)
    lines = open(tmp_path / "s" / "training_log.txt").read().splitlines()
    assert lines[0] == "Iteration,Loss_D,Loss_G_adv,Loss_Reg,Loss_Reg_weighted" and len(lines) == 3
    assert os.path.exists(tmp_path / "s" / "kernel_iter2.npy") and os.path.exists(tmp_path / "s" / "kernel_merged.npy")
