#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE's own
functions (read-only, from /root/reference) and running them on seeded synthetic
inputs and on the data artefacts the reference ships (moe_kernels/*.npy, moe_model.pth,
output/single_kernel/kernelgan_out*/kernel_*.npy).

Run once in the build container (the GPU box never sees /root/reference):

    python tests/golden/make_golden.py [--out DIR]

What gets written is DATA only (inputs, expected outputs, re-serialised weights) --
never reference source text.  Each .npz notes the reference file:line that produced it.

Reference modules import `netCDF4` at top level; it is not installed here, so a stub
module is registered first.  Only pure-tensor functions are called.
"""
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _import_reference():
    stub = types.ModuleType("netCDF4")
    stub.Dataset = object
    sys.modules["netCDF4"] = stub
    import matplotlib
    matplotlib.use("Agg")
    k = os.path.join(REF, "kernel_from_lr_gan")
    mods = {}
    mods["c30"] = _load("ref_c30", os.path.join(k, "C_30apply_kernel_to_landsat.py"))
    mods["c31"] = _load("ref_c31", os.path.join(k, "C_31apply_muti_kernel_to_landsat.py"))
    mods["d"] = _load("ref_d", os.path.join(k, "D_build_noise_pool.py"))
    mods["e"] = _load("ref_e", os.path.join(k, "E_make_train_data.py"))
    # muti_kernel uses bare `from networks import ...`
    sys.path.insert(0, os.path.join(k, "muti_kernel"))
    mods["mnet"] = _load("networks", os.path.join(k, "muti_kernel", "networks.py"))
    mods["mloss"] = _load("loss", os.path.join(k, "muti_kernel", "loss.py"))
    mods["gemini"] = _load("ref_gemini", os.path.join(k, "muti_kernel", "train_gemini.py"))
    sys.path.pop(0)
    mods["snet"] = _load("ref_single_networks", os.path.join(k, "single_kernel", "networks.py"))
    return mods


def landsat_like(shape, seed):
    """Synthetic 'Landsat-shaped' radiance tile: rand*80+20, seeded."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * 80.0 + 20.0


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    m = _import_reference()
    quiet = open(os.devnull, "w")

    # ------------------------------------------------------------------ kernels / sigmas (data)
    moe_k = np.stack([np.load(f"{REF}/moe_kernels/kernel_{i}.npy") for i in range(10)]).astype(np.float32)
    moe_s = np.stack([np.load(f"{REF}/moe_kernels/sigma_{i}.npy") for i in range(10)]).astype(np.float32)
    kg_pb = np.load(f"{REF}/output/single_kernel/kernelgan_out/kernel_per_band.npy").astype(np.float32)
    kg_mg = np.load(f"{REF}/output/single_kernel/kernelgan_out/kernel_merged.npy").astype(np.float32)
    kg128_pb = np.load(f"{REF}/output/single_kernel/kernelgan_out_128/kernel_per_band.npy").astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "kernels.npz"),
        moe_kernels=moe_k, moe_sigmas=moe_s,
        kgan_per_band=kg_pb, kgan_merged=kg_mg, kgan128_per_band=kg128_pb,
        note="moe_kernels/kernel_i.npy,sigma_i.npy; output/single_kernel/kernelgan_out{,_128}/kernel_{per_band,merged}.npy",
    )

    # ------------------------------------------------------------------ V-A  apply_kernel_degradation
    # C_30apply_kernel_to_landsat.py:68-124 == C_31apply_muti_kernel_to_landsat.py:59-97
    va = {}
    zero_band = moe_k[0].copy()
    zero_band[2] = 0.0                      # a band with sum 0 is left un-normalised (C_30:93-97)
    unnorm = kg_pb * np.array([2.0, 0.5, 3.0, 1.0, 0.25], np.float32)[:, None, None]
    cases = [
        # name,            (C,H,W),        f,  kernel,              seed
        ("c5_512_f4_moe0", (5, 512, 512), 4, moe_k[0], 0),
        ("c5_512_f8_moe0", (5, 512, 512), 8, moe_k[0], 0),
        ("c5_256_f8_kgan", (5, 256, 256), 8, kg_pb, 1),
        ("c3_256_f4_merged2d", (3, 256, 256), 4, kg_mg, 2),
        ("c7_260x257_f4_merged2d", (7, 260, 257), 4, kg_mg, 3),
        ("c5_96_f6_is_f4", (5, 96, 96), 6, moe_k[3], 4),
        ("c5_64_f1", (5, 64, 64), 1, kg128_pb, 5),
        ("c5_64_f2", (5, 64, 64), 2, kg128_pb, 6),
        ("c5_128_f16", (5, 128, 128), 16, moe_k[7], 7),
        ("c5_70x90_f8_odd", (5, 70, 90), 8, kg_pb, 8),
        ("c5_64_f4_zero_band", (5, 64, 64), 4, zero_band, 9),
        ("c5_64_f4_unnormalised", (5, 64, 64), 4, unnorm, 10),
        ("c5_16_f4_tiny", (5, 16, 16), 4, moe_k[1], 11),
    ]
    stdout = sys.stdout
    for name, shape, f, kern, seed in cases:
        img = landsat_like(shape, seed)
        k = torch.from_numpy(kern.copy())
        o30 = m["c30"].apply_kernel_degradation(img, k, f)
        o31 = m["c31"].apply_kernel_degradation(img, k, f)
        assert torch.equal(o30, o31), name
        va[name + "__out"] = o30.numpy()
        va[name + "__kernel"] = kern
        va[name + "__meta"] = np.array([shape[0], shape[1], shape[2], f, seed], np.int64)
        # inputs are regenerated from the seed by the tests; keep a checksum to prove identity
        va[name + "__in_sum"] = np.array([img.double().sum().item(), img[0, 0, :8].double().sum().item()])
    # NaN propagation case (inputs stored: small)
    img = landsat_like((5, 64, 64), 12)
    img[1, 30, 41] = float("nan")
    o = m["c30"].apply_kernel_degradation(img, torch.from_numpy(moe_k[0].copy()), 4)
    va["nan_case__in"] = img.numpy()
    va["nan_case__out"] = o.numpy()
    va["nan_case__kernel"] = moe_k[0]
    va["nan_case__meta"] = np.array([5, 64, 64, 4, 12], np.int64)
    va["note"] = np.array("C_30apply_kernel_to_landsat.py:68-124 (bit-identical to C_31:59-97); inputs = rand(seed)*80+20")
    np.savez_compressed(os.path.join(OUT, "va_cases.npz"), **va)

    # load_kernel semantics (C_31:22-37): 4-D -> mean over batch, 2-D -> [1,k,k]
    sys.stdout = quiet
    tmp = os.path.join(OUT, "_tmp_kernel.npy")
    batch_k = np.stack([moe_k[0], moe_k[4], moe_k[9]]).astype(np.float64)   # float64 on disk -> cast to f32
    np.save(tmp, batch_k)
    lk4 = m["c31"].load_kernel(tmp).numpy()
    np.save(tmp, kg_mg)
    lk2_c31 = m["c31"].load_kernel(tmp).numpy()
    lk2_c30 = m["c30"].load_kernel(tmp).numpy()
    os.remove(tmp)
    sys.stdout = stdout
    np.savez_compressed(os.path.join(OUT, "load_kernel.npz"), batch_in=batch_k, batch_out=lk4,
                        k2d_in=kg_mg, k2d_out_c31=lk2_c31, k2d_out_c30=lk2_c30,
                        note="C_31:22-37 / C_30:18-33")

    # ------------------------------------------------------------------ V-B  deterministic part + full forward
    # train_gemini.py:124-134 (grouped conv, zero pad 6, ::4)
    vb = {}
    import torch.nn.functional as F
    for name, (B, C, H, W), seed in [("b4_256", (4, 5, 256, 256), 20), ("b3_250x253", (3, 5, 250, 253), 21),
                                     ("b2_64", (2, 5, 64, 64), 22)]:
        x = landsat_like((B, C, H, W), seed)
        kid = np.array([(3 * i + 1) % 10 for i in range(B)], np.int32)
        bk = torch.from_numpy(moe_k[kid])                          # [B,C,13,13]
        out = F.conv2d(x.view(1, B * C, H, W), bk.view(B * C, 1, 13, 13), padding=6, groups=B * C)
        out = out.view(B, C, H, W)[:, :, ::4, ::4].contiguous()
        vb[name + "__out"] = out.numpy()
        vb[name + "__kid"] = kid
        vb[name + "__meta"] = np.array([B, C, H, W, seed], np.int64)
        vb[name + "__in_sum"] = np.array([x.double().sum().item()])
    # peaked kernels (kgan) so that border attenuation / asymmetry is exercised
    x = landsat_like((2, 5, 96, 96), 23)
    bk = torch.from_numpy(np.stack([kg_pb, kg128_pb]))
    out = F.conv2d(x.view(1, 10, 96, 96), bk.view(10, 1, 13, 13), padding=6, groups=10).view(2, 5, 96, 96)[:, :, ::4, ::4]
    vb["kgan_b2_96__out"] = out.contiguous().numpy()
    vb["kgan_b2_96__kernels"] = bk.numpy()
    vb["kgan_b2_96__meta"] = np.array([2, 5, 96, 96, 23], np.int64)

    model = m["gemini"].ContentAdaptiveDegradation(n_kernels=10, n_channels=5, kernel_size=13)
    sd = torch.load(f"{REF}/moe_kernels/moe_model.pth", weights_only=True, map_location="cpu")
    model.load_state_dict(sd)
    model.eval()
    x = landsat_like((4, 5, 256, 256), 20)
    with torch.no_grad():
        logits = model.selector(x)
        vk = model.get_effective_kernels()
        vs = model.get_effective_sigmas()
        torch.manual_seed(123)
        out, weights, valid_k = model(x, temp=1.0, hard=True)
        # soft mixing path: weights given, deterministic part only
        torch.manual_seed(7)
        w_soft = F.gumbel_softmax(logits, tau=1.0, hard=False, dim=1)
        bk = torch.einsum("bk,kchw->bchw", w_soft, vk)
        bs = torch.einsum("bk,kc->bc", w_soft, vs)
        soft_clean = F.conv2d(x.view(1, 20, 256, 256), bk.reshape(20, 1, 13, 13), padding=6, groups=20)
        soft_clean = soft_clean.view(4, 5, 256, 256)[:, :, ::4, ::4].contiguous()
    vb["fwd__logits"] = logits.numpy()
    vb["fwd__valid_kernels"] = vk.numpy()
    vb["fwd__valid_sigmas"] = vs.numpy()
    vb["fwd__hard_weights"] = weights.numpy()
    vb["fwd__hard_out"] = out.numpy()           # includes torch-CPU randn noise: statistical use only
    vb["fwd__soft_weights"] = w_soft.numpy()
    vb["fwd__soft_kernels"] = bk.numpy()
    vb["fwd__soft_sigmas"] = bs.numpy()
    vb["fwd__soft_clean"] = soft_clean.numpy()
    vb["note"] = np.array("train_gemini.py:88-140; x=rand(seed 20)*80+20 [4,5,256,256]; moe_model.pth eval mode")
    np.savez_compressed(os.path.join(OUT, "vb_cases.npz"), **vb)

    # selector weights as data (state_dict tensors re-serialised) -> lets the GPU box rebuild the model
    np.savez_compressed(os.path.join(OUT, "moe_model_state.npz"),
                        **{k: v.numpy() for k, v in sd.items()})

    # ------------------------------------------------------------------ V-C  add_noise / random_crop index streams
    # E_make_train_data.py:65-74 with np.random.seed(42) (E:190); D_build_noise_pool.py:41-53 with random.seed(42) (D:65)
    pool = np.arange(1000 * 5 * 4 * 4, dtype=np.float32).reshape(1000, 5, 4, 4)
    blurred = np.zeros((5, 4, 4), np.float32)
    np.random.seed(42)
    idxs = []
    for _ in range(32):
        o = m["e"].add_noise(blurred, pool)
        idxs.append(int(o[0, 0, 0]) // (5 * 4 * 4))
    data = np.arange(5 * 256 * 256, dtype=np.float32).reshape(5, 256, 256)
    random.seed(42)
    crops = m["d"].random_crop(data, 32, 16)
    offs = [(int(c[0, 0, 0]) // 256, int(c[0, 0, 0]) % 256) for c in crops]
    random.seed(42)
    crops_b = m["d"].random_crop(data[:, :40, :33], 32, 8)   # tight bounds: inclusive upper limit matters
    offs_b = [(int(c[0, 0, 0]) // 256, int(c[0, 0, 0]) % 256) for c in crops_b]
    np.savez_compressed(os.path.join(OUT, "rng_streams.npz"),
                        add_noise_idx_seed42_N1000=np.array(idxs, np.int64),
                        random_crop_offs_seed42_256=np.array(offs, np.int64),
                        random_crop_offs_seed42_40x33=np.array(offs_b, np.int64),
                        note="E_make_train_data.py:65-74,190; D_build_noise_pool.py:41-53,65")

    # ------------------------------------------------------------------ KernelGAN: extract_effective_kernels + forward
    # muti_kernel/networks.py:101-202 ; single_kernel/networks.py:83-157
    kg = {}
    torch.manual_seed(0)
    dm = m["mnet"].DegradationModel(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1,
                                    noise_init=0.3, noise_max=1.2)
    # the default init (randn*0.01) composes to ~1e-12 kernels -> clamp/normalise degenerate; scale up so the
    # composed kernel has O(1) mass and mixed signs (exercises clamp>=0 and the sum normalisation)
    with torch.no_grad():
        for band in dm.generator.weight_bands:
            for li, w in enumerate(band):
                w.mul_(100.0 / (w.shape[1] ** 0.5) / w.shape[-1])
                if li == 0:
                    w.add_(0.05)
    dm.eval()
    x = landsat_like((2, 5, 64, 64), 30) / 100.0
    with torch.no_grad():
        k_b = dm.generator.extract_effective_kernels(x, reduce_batch=False)       # [2,5,13,13]
        k_r = dm.generator.extract_effective_kernels(x, reduce_batch=True)        # [5,13,13]
        k_none = dm.generator.extract_effective_kernels(None)                     # [5,13,13]
        scales = dm.generator.condition_encoder(x)
        fwd = dm.generator(x)                                                     # [2,5,8,8]
    for k_, v in dm.state_dict().items():
        kg["muti_sd__" + k_] = v.numpy()
    kg["muti__x"] = x.numpy()
    kg["muti__kernels_batch"] = k_b.numpy()
    kg["muti__kernels_reduced"] = k_r.numpy()
    kg["muti__kernels_noinput"] = k_none.numpy()
    kg["muti__forward"] = fwd.numpy()
    kg["muti__scales"] = np.stack([np.concatenate([s.numpy() for s in band], axis=1) for band in scales], axis=1)  # [B,5,161]
    # NoiseEstimator sigma clamp (networks.py:221-224)
    with torch.no_grad():
        _, sig = dm.noise_estimator(torch.zeros(1, 5, 4, 4))
    kg["muti__noise_sigma"] = sig.numpy()

    sg = m["snet"].MultiBandLinearGenerator(in_ch=5, mid_ch=32)
    with torch.no_grad():
        kg["single_default__kernels"] = sg.extract_effective_kernels().numpy()
        xs = landsat_like((2, 5, 64, 64), 31)
        kg["single_default__forward"] = sg(xs).numpy()
        torch.manual_seed(1)
        for p in sg.parameters():
            p.add_(torch.randn_like(p) * 0.02)
        kg["single_perturbed__kernels"] = sg.extract_effective_kernels().numpy()
        kg["single_perturbed__forward"] = sg(xs).numpy()
    for k_, v in sg.state_dict().items():
        kg["single_perturbed_sd__" + k_] = v.numpy()
    kg["single__x_meta"] = np.array([2, 5, 64, 64, 31], np.int64)
    kg["note"] = np.array("muti_kernel/networks.py:101-202 (DegradationModel seed 0, weights rescaled, see make_golden.py); "
                          "single_kernel/networks.py:83-157")
    np.savez_compressed(os.path.join(OUT, "kernelgan.npz"), **kg)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"{f:28s} {os.path.getsize(os.path.join(OUT, f)) / 1024:9.1f} KiB")


def patch_cutter_scene(seed=0, H=700, W=900):
    """Synthetic mostly-water scene for the patch cutter: 5 bands, NIR (band 4) in [0.5, 6.5) over water, land blobs with
    NIR 50, blocks of the invalid marker -9999.  Regenerated from the seed by the tests (NumPy RandomState is portable)."""
    rs = np.random.RandomState(seed)
    data = (rs.rand(5, H, W) * 6.0 + 0.5).astype(np.float32)
    data[4, 300:380, 100:400] = 50.0          # land: NIR above THRESHOLD_MAX
    data[4, 20:25, 600:640] = 0.0             # below THRESHOLD_MIN
    data[:, 100:140, 200:260] = -9999.0       # invalid marker in all bands
    data[2, 600:601, 850:851] = -9999.0       # a single invalid value in one band only
    return data


def make_patch_cutter_golden():
    """A_00_patch_cutter_universal.py:89-197: apply_water_mask + create_patches_nc (its NetCDF writer replaced by a recorder)."""
    import contextlib
    import io
    import tempfile
    stub = types.ModuleType("netCDF4")
    stub.Dataset = object
    sys.modules["netCDF4"] = stub
    pc = _load("ref_patch_cutter", os.path.join(REF, "kernel_from_lr_gan", "A_00_patch_cutter_universal.py"))
    data = patch_cutter_scene()
    rec = []
    pc.save_patch_as_nc = lambda patch, path, metadata, i, j, hs, ws: rec.append(
        (i, j, hs, ws, int(np.isnan(patch).sum()), float(np.nansum(patch.astype(np.float64)))))
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        masked = pc.apply_water_mask(data.copy(), pc.THRESHOLD_MIN, pc.THRESHOLD_MAX)
        for tag, thr, ps, sr in (("thr0", 0.0, 256, 0.5), ("thr002", 0.02, 256, 0.5), ("ps128", 0.01, 128, 0.75)):
            rec.clear()
            tot, kept = pc.create_patches_nc(masked, ps, sr, thr, tempfile.mkdtemp(), "p", {})
            out[tag + "__total_kept"] = np.array([tot, kept], np.int64)
            out[tag + "__rec"] = np.array([r[:5] for r in rec], np.int64).reshape(-1, 5)
            out[tag + "__sums"] = np.array([r[5] for r in rec], np.float64)
            out[tag + "__args"] = np.array([thr, ps, sr], np.float64)
    out["masked_nan_count"] = np.array([int(np.isnan(masked).sum())], np.int64)
    out["masked_nan_rowsum"] = np.isnan(masked).sum(axis=(0, 2)).astype(np.int64)
    out["masked_sum"] = np.array([float(np.nansum(masked.astype(np.float64)))])
    out["consts"] = np.array([pc.THRESHOLD_MIN, pc.THRESHOLD_MAX, pc.INVALID_VALUE, pc.NIR_BAND_INDEX], np.float64)
    out["note"] = np.array("A_00_patch_cutter_universal.py:89-197; scene = make_golden.patch_cutter_scene(seed 0)")
    np.savez_compressed(os.path.join(OUT, "patch_cutter.npz"), **out)
    print({k: (v.tolist() if v.size <= 3 else v.shape) for k, v in out.items() if k.endswith("total_kept") or k == "masked_nan_count"})


def make_moe_backward_golden():
    """Gradients the reference's MoE trainer computes (train_gemini.py:169-232: loss.backward() through
    ContentAdaptiveDegradation.forward :88-140 and SelectorNet :14-39 in TRAIN mode -- the trainer never calls .eval()).
    Inputs are regenerated from seeds by the test; only the reference's outputs are stored.
      d kernel_bank of  L = sum(out * G)   (independent of the forward's randn noise; weights = the forward's own Gumbel draw)
      train-mode selector logits and d conv0.weight / d classifier.weight of  L = sum(logits * Gl)"""
    m = _import_reference()
    model = m["gemini"].ContentAdaptiveDegradation(n_kernels=10, n_channels=5, kernel_size=13)
    sd = torch.load(f"{REF}/moe_kernels/moe_model.pth", weights_only=True, map_location="cpu")
    model.load_state_dict(sd)
    model.train()
    B, C, H, W = 3, 5, 40, 44
    x = landsat_like((B, C, H, W), 31)
    G = torch.randn(B, C, 10, 11, generator=torch.Generator().manual_seed(32))
    Gl = torch.randn(B, 10, generator=torch.Generator().manual_seed(33))
    out = {}
    torch.manual_seed(5)
    o, weights, vk = model(x, temp=2.0, hard=False)
    model.zero_grad()
    (o * G).sum().backward()
    out["weights"] = weights.detach().numpy()
    out["d_kernel_bank"] = model.kernel_bank.grad.clone().numpy()
    # selector alone, fresh copy of the checkpoint (the forward above updated the BN running stats)
    sel = m["gemini"].SelectorNet(in_ch=5, num_classes=10)
    sel.load_state_dict({k[len("selector."):]: v for k, v in sd.items() if k.startswith("selector.")})
    sel.train()
    logits = sel(x)
    (logits * Gl).sum().backward()
    out["sel_train_logits"] = logits.detach().numpy()
    out["sel_d_conv0_w"] = sel.features[0].weight.grad.numpy()
    out["sel_d_conv6_w_sum"] = np.array([sel.features[6].weight.grad.double().sum().item(),
                                         sel.features[6].weight.grad.double().abs().sum().item()])
    out["sel_d_cls_w"] = sel.classifier.weight.grad.numpy()
    out["sel_running_mean1"] = sel.features[1].running_mean.numpy()
    # losses (muti_kernel/loss.py:6-41) and the critic (muti_kernel/networks.py:250-268) the same trainer uses
    L = m["mloss"]
    ks = np.load(f"{REF}/moe_kernels/kernel_0.npy")
    kneg = ks[0] - 0.004
    out["kreg_kernel0"] = np.array([L.kernel_regularization(torch.from_numpy(ks[c]), alpha=0.5).item() for c in range(5)])
    out["kreg_neg"] = np.array([L.kernel_regularization(torch.from_numpy(kneg)).item()])
    pr = torch.randn(2, 1, 8, 8, generator=torch.Generator().manual_seed(34))
    pf = torch.randn(2, 1, 8, 8, generator=torch.Generator().manual_seed(35))
    out["lsgan"] = np.array([L.lsgan_d_loss(pr, pf).item(), L.lsgan_g_loss(pf).item()])
    nets = m["mnet"]
    torch.manual_seed(0)
    Dref = nets.PatchDiscriminator(in_ch=5)
    for k_, v_ in Dref.state_dict().items():
        out["D__" + k_] = v_.detach().clone().numpy()
    xd = torch.randn(2, 5, 16, 16, generator=torch.Generator().manual_seed(36))
    Dref.eval()
    with torch.no_grad():
        out["D_eval_out"] = Dref(xd).numpy()
    Dref.train()
    yd = Dref(xd)                      # one power iteration of every spectral norm + batch-statistics BatchNorm
    yd.pow(2).mean().backward()
    out["D_train_out"] = yd.detach().numpy()
    out["D_train_u0"] = Dref.net[0].weight_u.detach().numpy()
    out["D_train_d_w0"] = Dref.net[0].weight_orig.grad.numpy()
    out["meta"] = np.array([B, C, H, W, 31, 32, 33], np.int64)
    out["note"] = np.array("train_gemini.py:88-140,169-232; x=landsat_like((3,5,40,44),31), G=randn(seed 32) [3,5,10,11], "
                           "Gl=randn(seed 33) [3,10], moe_model.pth, train mode, temp=2.0")
    np.savez_compressed(os.path.join(OUT, "moe_backward.npz"), **out)
    print({k: v.shape for k, v in out.items()})


def _grad_record(out, prefix, named_grads, full):
    """Store a gradient set: tensors whose name is in `full` whole, every tensor as (sum, sum|.|, sum of squares) in float64."""
    for name, g in named_grads:
        g = g.detach()
        out[f"{prefix}chk__{name}"] = np.array([g.double().sum().item(), g.double().abs().sum().item(),
                                               g.double().pow(2).sum().item()])
        if name in full:
            out[f"{prefix}full__{name}"] = g.numpy().copy()


def make_kernelgan_train_golden(dt=torch.float32):
    """(dt = torch.float64: the same replay with every module and input cast to double -- returns the gradient records instead of
    writing the fixture; make_f64_truth_golden stores them as the float64 truth the tolerance tests measure both sides against.)
    One iteration of each KernelGAN trainer, run with the reference's own classes and loss functions in the statement
    order of muti_kernel/train.py:262-286 and single_kernel/train.py:269-308 (the loops live inside main() and read files from
    H:\\..., so the statements are replayed here on seeded inputs).  Weights: the state_dicts already stored in kernelgan.npz;
    critic: PatchDiscriminator(in_ch=5) under torch.manual_seed(0) (its state_dict is stored in moe_backward.npz as D__*).
    torch.randn_like inside NoiseEstimator.forward is replaced by a seeded eps field so the noisy fake is reproducible."""
    m = _import_reference()
    k = os.path.join(REF, "kernel_from_lr_gan")
    sloss = _load("ref_single_loss", os.path.join(k, "single_kernel", "loss.py"))
    KG = np.load(os.path.join(OUT, "kernelgan.npz"))
    out = {}
    # ---------------------------------------------------------------- multi-kernel trainer
    nets, L = m["mnet"], m["mloss"]
    model = nets.DegradationModel(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1, noise_init=0.3, noise_max=1.2)
    model.load_state_dict({n[len("muti_sd__"):]: torch.from_numpy(KG[n]) for n in KG.files if n.startswith("muti_sd__")})
    torch.manual_seed(0)
    D = nets.PatchDiscriminator(in_ch=5, base_ch=64, num_blocks=4)
    model, D = model.to(dt), D.to(dt)
    opt_D = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_G = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
    patches = torch.from_numpy(KG["muti__x"]).to(dt)                                     # [2,5,64,64]
    real_ds = (landsat_like((2, 5, 8, 8), 41) / 100.0).to(dt)
    eps = torch.randn(2, 5, 8, 8, generator=torch.Generator().manual_seed(42)).to(dt)
    target_sigma = torch.tensor([0.55, 0.72, 0.83, 0.63, 0.19]).to(dt)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.clone()
    try:
        clean_fake, fake_ds, sigma = model(patches)
    finally:
        torch.randn_like = real_randn_like
    D.train(); model.train()
    loss_D = L.lsgan_d_loss(D(real_ds), D(fake_ds.detach()))
    opt_D.zero_grad(); loss_D.backward()
    _grad_record(out, "muti_D_", [(n, p.grad) for n, p in D.named_parameters()], {"net.0.weight_orig", "net.0.bias", "net.2.weight_orig",
                                                                                "net.3.weight", "net.14.weight_orig"})
    opt_D.step()
    pred_fake = D(fake_ds)
    loss_G_adv = L.lsgan_g_loss(pred_fake)
    ks_band = model.generator.extract_effective_kernels(patches, reduce_batch=True)
    loss_reg = torch.mean(torch.stack([L.kernel_regularization(k=ks_band[i], alpha=0.5, beta=0.5, gamma=5.0, delta=1.0)
                                       for i in range(ks_band.shape[0])]))
    loss_noise = L.noise_reg_loss(sigma, target=target_sigma, mode='l2')
    loss_G = loss_G_adv + loss_reg + 20.0 * loss_noise
    out["muti_reg_requires_grad"] = np.array([int(loss_reg.requires_grad)])
    opt_G.zero_grad(); loss_G.backward()
    full = {"generator.weight_bands.0.%d" % l for l in range(6)} | {"generator.weight_bands.3.1", "generator.weight_bands.4.5",
            "generator.condition_encoder.encoder.0.weight", "generator.condition_encoder.encoder.0.bias",
            "generator.condition_encoder.encoder.4.bias", "generator.condition_encoder.fc.bias", "noise_estimator.log_sigma"}
    _grad_record(out, "muti_G_", [(n, p.grad) for n, p in model.named_parameters()], full)
    opt_G.step()
    out["muti_losses"] = np.array([loss_D.item(), loss_G_adv.item(), loss_reg.item(), loss_noise.item()])
    out["muti_clean"] = clean_fake.detach().numpy()
    out["muti_sigma"] = sigma.detach().numpy()
    out["muti_pred_fake_G"] = pred_fake.detach().numpy()
    out["muti_after_G__weight_bands.0.0"] = model.generator.weight_bands[0][0].detach().numpy().copy()
    out["muti_after_G__log_sigma"] = model.noise_estimator.log_sigma.detach().numpy().copy()
    out["muti_after_D__net.0.weight_orig"] = D.net[0].weight_orig.detach().numpy().copy()
    # ---------------------------------------------------------------- single-kernel trainer
    snets = m["snet"]
    G = snets.MultiBandLinearGenerator(in_ch=5, mid_ch=32)
    G.load_state_dict({n[len("single_perturbed_sd__"):]: torch.from_numpy(KG[n]) for n in KG.files
                       if n.startswith("single_perturbed_sd__")})
    torch.manual_seed(0)
    D = snets.PatchDiscriminator(in_ch=5, base_ch=64, num_blocks=4)
    G, D = G.to(dt), D.to(dt)
    opt_D = torch.optim.Adam(D.parameters(), lr=4e-4, betas=(0.5, 0.999))
    opt_G = torch.optim.Adam(G.parameters(), lr=4e-4, betas=(0.5, 0.999))
    from torch.nn.utils import clip_grad_norm_
    patches = (landsat_like((2, 5, 64, 64), 31) / 100.0).to(dt)
    real_ds = (landsat_like((2, 5, 8, 8), 43) / 100.0).to(dt)
    fake_ds = G(patches)
    D.train(); G.train()
    loss_D = sloss.lsgan_d_loss(D(real_ds), D(fake_ds.detach()))
    opt_D.zero_grad(); loss_D.backward()
    gn_D = clip_grad_norm_(parameters=D.parameters(), max_norm=20.0)
    opt_D.step()
    pred_fake = D(fake_ds)
    loss_G_adv = sloss.lsgan_g_loss(pred_fake)
    ks_band = G.extract_effective_kernels()
    loss_reg = torch.mean(torch.stack([sloss.kernel_regularization(k=ks_band[i], alpha=0.5, beta=0.5, gamma=5.0, delta=1.0,
                                                                   epsilon=3.0) for i in range(ks_band.shape[0])]))
    loss_G = loss_G_adv + 0.002 * loss_reg
    opt_G.zero_grad(); loss_G.backward()
    _grad_record(out, "single_G_preclip_", [(n, p.grad.clone()) for n, p in G.named_parameters()],
                 {"chains.0.%d.weight" % l for l in range(6)} | {"chains.2.1.weight", "chains.4.5.weight"})
    gn_G = clip_grad_norm_(parameters=G.parameters(), max_norm=20.0)
    opt_G.step()
    out["single_losses"] = np.array([loss_D.item(), loss_G_adv.item(), loss_reg.item(), float(gn_D), float(gn_G)])
    out["single_fake"] = fake_ds.detach().numpy()
    out["single_after_G__chains.0.0.weight"] = G.chains[0][0].weight.detach().numpy().copy()
    out["single_after_G__chains.1.2.weight"] = G.chains[1][2].weight.detach().numpy().copy()
    if dt != torch.float32:
        return out
    # ---------------------------------------------------------------- the trainers' samplers (global torch RNG draw order)
    import tempfile
    sys.path.insert(0, os.path.join(k, "muti_kernel"))
    mtrain = _load("ref_muti_train", os.path.join(k, "muti_kernel", "train.py"))
    sys.path.pop(0)
    sys.modules.pop("networks", None); sys.modules.pop("loss", None)
    sys.path.insert(0, os.path.join(k, "single_kernel"))
    strain = _load("ref_single_train", os.path.join(k, "single_kernel", "train.py"))
    sys.path.pop(0)
    with tempfile.TemporaryDirectory() as td:
        for i in range(6):
            np.save(os.path.join(td, f"p{i}.npy"), (landsat_like((5, 40, 44), 60 + i) / 100.0).numpy())
        files, osz = mtrain.load_patches_from_folder(td)
        torch.manual_seed(7)
        a = mtrain.sample_patches_from_files(files, batch_size=4, target_size=32, original_size=osz)
        b = mtrain.sample_patches_from_files(files, batch_size=3, target_size=40, original_size=osz)
    out["sampler_muti_a"] = a.numpy()[:, :, ::8, ::8].copy()
    out["sampler_muti_b"] = b.numpy()[:, :, ::8, ::8].copy()
    out["sampler_muti_osz"] = np.array([osz])
    torch.manual_seed(8)
    full = torch.stack([landsat_like((5, 40, 44), 70 + i) / 100.0 for i in range(3)])
    out["sampler_single_crop"] = strain.crop_patches(full, crop_size=32).numpy()[:, :, ::8, ::8].copy()
    out["note"] = np.array("muti_kernel/train.py:262-286 + single_kernel/train.py:269-308, one iteration each; inputs: kernelgan.npz "
                           "weights, muti__x / landsat_like((2,5,64,64),31)/100, real_ds landsat_like((2,5,8,8),41|43)/100, "
                           "eps randn seed 42, D = PatchDiscriminator under manual_seed(0)")
    np.savez_compressed(os.path.join(OUT, "kernelgan_train.npz"), **out)
    print("kernelgan_train.npz", os.path.getsize(os.path.join(OUT, "kernelgan_train.npz")) // 1024, "KiB;",
          "muti losses", out["muti_losses"], "single", out["single_losses"], "reg requires grad:", out["muti_reg_requires_grad"])


def make_f64_truth_golden():
    """float64 TRUTH for the reference-pinned comparisons whose fp32 tolerance is looser than north_star's 1e-4 (round-2 verdict,
    weak #2): the reference's OWN classes, cast to double, on the same state_dicts / inputs as the fp32 fixtures.  The tests assert
    |HIP - truth| <= max(|reference fp32 fixture - truth|, 1e-4 * scale) per element: the HIP path is at least as close to the
    exact result as the reference's own fp32 run is, or within 1e-4.  Only arrays are stored (f64_truth.npz)."""
    m = _import_reference()
    KG = np.load(os.path.join(OUT, "kernelgan.npz"))
    out = {}
    with torch.no_grad():
        dm = m["mnet"].DegradationModel(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1, noise_init=0.3, noise_max=1.2)
        dm.load_state_dict({n[len("muti_sd__"):]: torch.from_numpy(KG[n]) for n in KG.files if n.startswith("muti_sd__")})
        dm = dm.double().eval()
        x = torch.from_numpy(KG["muti__x"]).double()
        out["muti__kernels_batch"] = dm.generator.extract_effective_kernels(x, reduce_batch=False).numpy()
        out["muti__kernels_reduced"] = dm.generator.extract_effective_kernels(x, reduce_batch=True).numpy()
        out["muti__kernels_noinput"] = dm.generator.extract_effective_kernels(None).numpy()
        out["muti__forward"] = dm.generator(x).numpy()
        sg = m["snet"].MultiBandLinearGenerator(in_ch=5, mid_ch=32)
        sg.load_state_dict({n[len("single_perturbed_sd__"):]: torch.from_numpy(KG[n]) for n in KG.files if n.startswith("single_perturbed_sd__")})
        sg = sg.double()
        out["single_perturbed__kernels"] = sg.extract_effective_kernels().numpy()
        out["single_perturbed__forward"] = sg(landsat_like((2, 5, 64, 64), 31).double()).numpy()
    # SelectorNet in train mode (train_gemini.py:14-39, batch-statistics BatchNorm) and d kernel_bank of the V-B forward (:107-134)
    sd = torch.load(f"{REF}/moe_kernels/moe_model.pth", weights_only=True, map_location="cpu")
    B, C, H, W = 3, 5, 40, 44
    x = landsat_like((B, C, H, W), 31).double()
    Gl = torch.randn(B, 10, generator=torch.Generator().manual_seed(33)).double()
    sel = m["gemini"].SelectorNet(in_ch=5, num_classes=10)
    sel.load_state_dict({k[len("selector."):]: v for k, v in sd.items() if k.startswith("selector.")})
    sel = sel.double().train()
    logits = sel(x)
    (logits * Gl).sum().backward()
    out["sel_train_logits"] = logits.detach().numpy()
    out["sel_d_conv0_w"] = sel.features[0].weight.grad.numpy()
    out["sel_d_cls_w"] = sel.classifier.weight.grad.numpy()
    out["sel_d_conv6_w_abs_sum"] = np.array([sel.features[6].weight.grad.abs().sum().item()])
    out["sel_running_mean1"] = sel.features[1].running_mean.numpy()
    # d kernel_bank for the weights the fp32 fixture recorded (the Gumbel draw itself is not reproducible in double): the statements of
    # ContentAdaptiveDegradation.forward :107-134 in float64
    MB = np.load(os.path.join(OUT, "moe_backward.npz"))
    G = torch.randn(B, C, 10, 11, generator=torch.Generator().manual_seed(32)).double()
    kb = sd["kernel_bank"].double().requires_grad_(True)
    w = torch.from_numpy(MB["weights"]).double()
    vk = torch.nn.functional.softmax(kb.view(10, C, -1), dim=-1).view(10, C, 13, 13)
    bk = torch.einsum("bk,kchw->bchw", w, vk)
    o = torch.nn.functional.conv2d(x.view(1, B * C, H, W), bk.reshape(B * C, 1, 13, 13), padding=6, groups=B * C).view(B, C, H, W)[:, :, ::4, ::4]
    (o * G).sum().backward()
    out["d_kernel_bank"] = kb.grad.numpy()
    # one iteration of each KernelGAN trainer in double: the stored gradient tensors
    tr = make_kernelgan_train_golden(torch.float64)
    for k_, v in tr.items():
        if "full__" in k_ or k_ in ("muti_clean", "single_fake", "muti_losses", "single_losses"):
            out["train__" + k_] = np.asarray(v, np.float64)
    out["note"] = np.array("the reference's classes cast to double on the inputs of kernelgan.npz / moe_backward.npz / kernelgan_train.npz "
                           "(muti_kernel/networks.py:101-202, single_kernel/networks.py:83-157, train_gemini.py:14-39,107-134, "
                           "muti_kernel/train.py:262-286, single_kernel/train.py:269-308)")
    np.savez_compressed(os.path.join(OUT, "f64_truth.npz"), **out)
    print("f64_truth.npz", os.path.getsize(os.path.join(OUT, "f64_truth.npz")) // 1024, "KiB", len(out), "arrays")
    # how far the reference's own fp32 run is from this truth (max |fp32 fixture - truth| / max |truth|)
    TR = np.load(os.path.join(OUT, "kernelgan_train.npz"))
    for k_ in ("muti__kernels_batch", "muti__forward", "single_perturbed__kernels", "single_perturbed__forward"):
        print(f"  reference fp32 vs float64  {k_:34s} {np.abs(KG[k_] - out[k_]).max() / np.abs(out[k_]).max():.2e}")
    for k_ in ("sel_train_logits", "sel_d_conv0_w", "sel_d_cls_w", "d_kernel_bank"):
        print(f"  reference fp32 vs float64  {k_:34s} {np.abs(MB[k_] - out[k_]).max() / np.abs(out[k_]).max():.2e}")
    for k_ in sorted(out):
        if k_.startswith("train__") and "full__" in k_:
            print(f"  reference fp32 vs float64  {k_:60s} {np.abs(TR[k_[7:]] - out[k_]).max() / np.abs(out[k_]).max():.2e}")


if __name__ == "__main__":
    # `--out DIR` writes somewhere else (tests/test_oracle_golden.py::test_make_golden_reproduces_fixtures regenerates every
    # fixture into a tmp dir and compares with the committed ones); one flag regenerates one family, no flag everything, in
    # dependency order (kernelgan.npz -> kernelgan_train.npz, moe_backward.npz -> f64_truth.npz).
    if "--out" in sys.argv:
        OUT = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
        os.makedirs(OUT, exist_ok=True)
    if "--f64-truth" in sys.argv:
        make_f64_truth_golden()
    elif "--kernelgan-train" in sys.argv:
        make_kernelgan_train_golden()
    elif "--moe-backward" in sys.argv:
        make_moe_backward_golden()
    elif "--patch-cutter" in sys.argv:
        make_patch_cutter_golden()
    elif "--base" in sys.argv:
        main()
    else:
        # every family in a fresh interpreter: ATen's fp32 / fp64 reductions depend in their last bits on the thread-pool
        # state an earlier generator leaves behind (measured: 1e-7 / 1e-15 relative), and the committed fixtures are the
        # stand-alone runs of each flag
        import subprocess
        for flag in ("--base", "--patch-cutter", "--moe-backward", "--kernelgan-train", "--f64-truth"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), flag, "--out", OUT])
