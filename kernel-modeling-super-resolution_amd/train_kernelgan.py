"""KernelGAN trainers: the callers of the generator chains, with every convolution of the step -- generator
forward / data gradient / weight gradient (kmsr_convk_*), condition encoder (kmsr_conv3x3_*), critic (kmsr_convk_*, zero
padding) and the kernel composition used for the regulariser / logging (kmsr_compose_kernels_f32) -- on the HIP kernels.

    muti_kernel/train.py:188-323    -> train_kernelgan_multi(model, D, patch_files | arrays, ...)
    single_kernel/train.py:124-347  -> train_kernelgan_single(G, D, arrays, ...)

Both loops keep the reference's order of operations: ONE generator forward per iteration, critic step on the detached
fake, then the generator step through the just-updated critic.  In both reference trainers the kernel regulariser is
computed from extract_effective_kernels(), which is decorated @torch.no_grad() (muti_kernel/networks.py:135,
single_kernel/networks.py:108): the term is logged and added to the loss but contributes NO gradient, so the generator is
driven by the adversarial term (and, in the multi-kernel trainer, sigma by the noise regulariser).  That behaviour is
reproduced here as is; it is also why no backward of the composition kernel exists.

Data: the multi-kernel trainer reads `.npy` patches exactly like the reference (train.py:21-69).  The single-kernel
trainer's reader is NetCDF (train.py:40-91; SURVEY 8f N4, not installable here), so it takes the patches as arrays.
Data parallel: every rank samples its own patches; gradients of both networks are averaged with kmsr_amd.dist.GradBucket.
"""
import glob
import os

import numpy as np
import torch
from torch.nn.utils import clip_grad_norm_

from . import dist as kd
from .train_moe import kernel_regularization, lsgan_d_loss, lsgan_g_loss


def kernel_regularization_single(k, alpha=0.5, beta=0.5, gamma=5.0, delta=1.0, epsilon=2.0):
    """single_kernel/loss.py:35-96: the four terms of muti_kernel/loss.py:16-41 plus epsilon * (max(k) - k[centre])^2."""
    kh, kw = k.shape
    centre_gap = (k.max() - k[int((kh - 1) / 2.0), int((kw - 1) / 2.0)]) ** 2
    return kernel_regularization(k, alpha, beta, gamma, delta) + epsilon * centre_gap


def noise_reg_loss(sigma, target=0.01, mode="l2"):
    """muti_kernel/loss.py:44-55: mean |sigma - target| (l1) or mean (sigma - target)^2; target scalar or per band."""
    target = torch.as_tensor(target, device=sigma.device, dtype=sigma.dtype)
    return (sigma - target).abs().mean() if mode == "l1" else ((sigma - target) ** 2).mean()


def load_patches_from_folder(patch_dir):
    """muti_kernel/train.py:21-30 -> (sorted .npy paths, edge length of the first patch)."""
    files = sorted(glob.glob(os.path.join(patch_dir, "*.npy")))
    if not files:
        raise ValueError(f"no .npy files in {patch_dir}")
    return files, int(np.load(files[0]).shape[1])


def sample_patches_from_files(patch_files, batch_size, target_size=128, original_size=128, device=None):
    """muti_kernel/train.py:33-69.  `patch_files`: .npy paths or in-memory [C,H,W] arrays.  Draw order on the global torch
    RNG is the reference's: the batch of file indices first, then per patch (y0, x0) when a crop is needed.  A patch holding
    NaN or smaller than the target raises ValueError."""
    picks = torch.randint(low=0, high=len(patch_files), size=(batch_size,))
    out = []
    for idx in picks.tolist():
        src = patch_files[idx]
        a = np.load(src) if isinstance(src, (str, os.PathLike)) else np.asarray(src)
        t = torch.from_numpy(a.astype(np.float32))
        if torch.isnan(t).any():
            raise ValueError(f"patch {src if isinstance(src, (str, os.PathLike)) else idx} holds {int(torch.isnan(t).sum())} NaN values")
        if original_size != target_size:
            H, W = t.shape[-2:]
            if H - target_size <= 0 or W - target_size <= 0:
                raise ValueError(f"patch {H}x{W} cannot be cropped to {target_size}x{target_size}")
            y0 = int(torch.randint(0, H - target_size + 1, (1,)))
            x0 = int(torch.randint(0, W - target_size + 1, (1,)))
            t = t[:, y0:y0 + target_size, x0:x0 + target_size]
        out.append(t)
    res = torch.stack(out)
    return res.to(device) if device is not None else res


def load_full_patches(patches, batch_size, device=None):
    """single_kernel/train.py:40-91 at array level: batch_size uniformly drawn whole patches (one torch.randint call)."""
    picks = torch.randint(low=0, high=len(patches), size=(batch_size,))
    out = []
    for idx in picks.tolist():
        t = torch.as_tensor(np.asarray(patches[idx], dtype=np.float32))
        if torch.isnan(t).any():
            raise ValueError(f"patch {idx} holds {int(torch.isnan(t).sum())} NaN values")
        out.append(t)
    res = torch.stack(out)
    return res.to(device) if device is not None else res


def crop_patches(full_patches, crop_size):
    """single_kernel/train.py:94-122: an independent random crop_size window per sample (y0 then x0; no draw when the size fits)."""
    B, C, H, W = full_patches.shape
    if H < crop_size or W < crop_size:
        raise ValueError(f"patch {H}x{W} is smaller than the crop {crop_size}x{crop_size}")
    rows = []
    for i in range(B):
        y0 = int(torch.randint(0, H - crop_size + 1, (1,))) if H > crop_size else 0
        x0 = int(torch.randint(0, W - crop_size + 1, (1,))) if W > crop_size else 0
        rows.append(full_patches[i, :, y0:y0 + crop_size, x0:x0 + crop_size])
    return torch.stack(rows)


def kernel_metrics(k):
    """muti_kernel/train.py:88-107: shape, sum, max, min, std, share of taps above 5 % of the peak, centre-of-mass offset."""
    kh, kw = k.shape
    mass = k + 1e-12
    rows = torch.arange(kh, device=k.device, dtype=torch.float32).view(kh, 1)
    cols = torch.arange(kw, device=k.device, dtype=torch.float32).view(1, kw)
    cy, cx = (rows * mass).sum() / mass.sum(), (cols * mass).sum() / mass.sum()
    return {"k_shape": f"{kh}x{kw}", "k_sum": float(k.sum()), "k_max": float(k.max()), "k_min": float(k.min()),
            "k_std": float(k.std()), "sparsity": float((k > k.max() * 0.05).float().mean()),
            "center_offset": float(((cy - (kh - 1) / 2.0) ** 2 + (cx - (kw - 1) / 2.0) ** 2).sqrt())}


def _adam(params, lr):
    return torch.optim.Adam(params, lr=lr, betas=(0.5, 0.999))            # muti train.py:227-228, single train.py:165-166


def kernelgan_multi_step(model, D, opt_G, opt_D, patches, real_ds, target_sigma, noise_reg_weight=20.0, buckets=None):
    """One iteration of muti_kernel/train.py:262-286.  Returns the detached (loss_D, loss_G_adv, loss_reg, loss_noise, sigma)."""
    bG, bD = buckets if buckets is not None else (None, None)
    clean_fake, fake_ds, sigma = model(patches)                                           # :262
    D.train()
    model.train()
    loss_D = lsgan_d_loss(D(real_ds), D(fake_ds.detach()))                                # :267-269
    opt_D.zero_grad(set_to_none=False) if bD is None else bD.zero()
    loss_D.backward()
    if bD is not None:
        bD.all_reduce_mean()
    opt_D.step()                                                                          # :270
    loss_G_adv = lsgan_g_loss(D(fake_ds))                                                 # :274-275
    ks_band = model.generator.extract_effective_kernels(patches, reduce_batch=True)       # :276  (no_grad: a constant)
    loss_reg = torch.stack([kernel_regularization(ks_band[i], alpha=0.5, beta=0.5, gamma=5.0, delta=1.0)
                            for i in range(ks_band.shape[0])]).mean()                     # :277-278
    loss_noise = noise_reg_loss(sigma, target=target_sigma, mode="l2")                    # :279
    loss_G = loss_G_adv + loss_reg + noise_reg_weight * loss_noise                        # :280
    opt_G.zero_grad(set_to_none=False) if bG is None else bG.zero()
    loss_G.backward()
    if bG is not None:
        bG.all_reduce_mean()
    opt_G.step()                                                                          # :281
    return loss_D.detach(), loss_G_adv.detach(), loss_reg.detach(), loss_noise.detach(), sigma.detach()


def train_kernelgan_multi(model, D, patch_files, original_patch_size=None, iters=3000, patch_size=256, batch_size=8,
                          lr_rate=1e-4, target_sigma=(0.55, 0.72, 0.83, 0.63, 0.19), noise_reg_weight=20.0, outdir=None,
                          kernel_log_every=100, mini_log_every=10, save_intermediate=True, log=print):
    """muti_kernel/train.py:199-323 on the HIP kernels.  model: kmsr_amd.DegradationModel, D: kmsr_amd.PatchDiscriminator (both
    on the GPU); patch_files: .npy paths (or arrays).  With outdir the same files are written as the reference's
    (iter_kernels/batch_kernels_iter<N>.npy, kernel_merged_iter<N>.npy, kernel_per_band_iter<N>.npy, final_results/...).
    Returns the history [iters, 5] = (loss_D, loss_G_adv, loss_reg, loss_noise, mean sigma), read back once at the end."""
    dev = next(model.parameters()).device
    if original_patch_size is None:
        first = patch_files[0]
        original_patch_size = int((np.load(first) if isinstance(first, (str, os.PathLike)) else np.asarray(first)).shape[1])
    tsig = torch.tensor(list(target_sigma), device=dev, dtype=torch.float32)              # :211
    opt_D, opt_G = _adam(D.parameters(), lr_rate), _adam(model.parameters(), lr_rate)
    kd.broadcast_parameters(model)
    kd.broadcast_parameters(D)
    buckets = (kd.GradBucket(model.parameters()), kd.GradBucket(D.parameters()))
    if outdir is not None:
        os.makedirs(os.path.join(outdir, "iter_kernels"), exist_ok=True)
    hist = torch.zeros(iters, 5, device=dev)
    prev_k = None
    patches = None
    for t in range(iters):
        it = t + 1
        patches = sample_patches_from_files(patch_files, batch_size, patch_size, original_patch_size, dev)       # :243-249
        real_ds = sample_patches_from_files(patch_files, batch_size, 32, original_patch_size, dev)               # :251-257
        lD, lG, lR, lN, sigma = kernelgan_multi_step(model, D, opt_G, opt_D, patches, real_ds, tsig, noise_reg_weight, buckets)
        hist[t, 0], hist[t, 1], hist[t, 2], hist[t, 3], hist[t, 4] = lD, lG, lR, lN, sigma.mean()
        if log is not None and mini_log_every and it % mini_log_every == 0:
            log(f"Iter {it}/{iters} | D: {float(lD):.4f} | G_adv: {float(lG):.4f} | Noise: {float(lN):.4f} | sigma_mean: {float(sigma.mean()):.4f}")
        if kernel_log_every and it % kernel_log_every == 0:                               # :290-321
            ks_all = model.generator.extract_effective_kernels(patches, reduce_batch=False)
            k_merged = ks_all.mean(dim=(0, 1))
            delta = 0.0 if prev_k is None else float(torch.norm(k_merged - prev_k))
            prev_k = k_merged.detach().clone()
            if log is not None:
                km = kernel_metrics(k_merged)
                log(f"  [Kernel] shape={km['k_shape']} sum={km['k_sum']:.4f} max={km['k_max']:.4f} min={km['k_min']:.4f} "
                    f"std={km['k_std']:.4f} sparsity(>5%max)={km['sparsity']:.3f} center_offset={km['center_offset']:.3f} delta_L2={delta:.5f}")
            if save_intermediate and outdir is not None and kd.rank() == 0:
                d = os.path.join(outdir, "iter_kernels")
                np.save(os.path.join(d, f"batch_kernels_iter{it}.npy"), ks_all.cpu().numpy())
                np.save(os.path.join(d, f"kernel_merged_iter{it}.npy"), k_merged.cpu().numpy())
                np.save(os.path.join(d, f"kernel_per_band_iter{it}.npy"), ks_all.mean(dim=0).cpu().numpy())
    if outdir is not None and patches is not None and kd.rank() == 0:                     # :323-331
        ks_final = model.generator.extract_effective_kernels(patches, reduce_batch=False).mean(dim=0).cpu().numpy()
        fin = os.path.join(outdir, "final_results")
        os.makedirs(fin, exist_ok=True)
        np.save(os.path.join(fin, "kernel_per_band.npy"), ks_final)
        np.save(os.path.join(fin, "kernel_merged.npy"), ks_final.mean(axis=0))
    return hist.cpu().numpy()


def kernelgan_single_step(G, D, opt_G, opt_D, patches, real_ds, reg_weight=0.002, grad_clip_norm=20.0, buckets=None):
    """One iteration of single_kernel/train.py:269-308.  Returns detached (loss_D, loss_G_adv, loss_reg, |grad D|, |grad G|)."""
    bG, bD = buckets if buckets is not None else (None, None)
    fake_ds = G(patches)                                                                  # :269
    D.train()
    G.train()
    loss_D = lsgan_d_loss(D(real_ds), D(fake_ds.detach()))                                # :273-275
    opt_D.zero_grad(set_to_none=False) if bD is None else bD.zero()
    loss_D.backward()
    if bD is not None:
        bD.all_reduce_mean()
    gn_D = clip_grad_norm_(D.parameters(), max_norm=grad_clip_norm)                       # :277
    opt_D.step()
    loss_G_adv = lsgan_g_loss(D(fake_ds))                                                 # :281-282
    ks_band = G.extract_effective_kernels()                                               # :283  (no_grad: a constant)
    loss_reg = torch.stack([kernel_regularization_single(ks_band[i], alpha=0.5, beta=0.5, gamma=5.0, delta=1.0, epsilon=3.0)
                            for i in range(ks_band.shape[0])]).mean()                     # :285-296
    loss_G = loss_G_adv + reg_weight * loss_reg                                           # :297-299
    opt_G.zero_grad(set_to_none=False) if bG is None else bG.zero()
    loss_G.backward()
    if bG is not None:
        bG.all_reduce_mean()
    gn_G = clip_grad_norm_(G.parameters(), max_norm=grad_clip_norm)                       # :301
    opt_G.step()
    return loss_D.detach(), loss_G_adv.detach(), loss_reg.detach(), gn_D.detach(), gn_G.detach()


def train_kernelgan_single(G, D, patches, iters=10000, lr_crop_size=32, batch_size=16, lr_rate=4e-4, reg_weight=0.002,
                           grad_clip_norm=20.0, outdir=None, kernel_log_every=100, save_intermediate=True, log=print):
    """single_kernel/train.py:134-347 on the HIP kernels.  G: kmsr_amd.MultiBandLinearGenerator, D: PatchDiscriminator; patches:
    a sequence of [5,H,W] arrays (the HR inputs; the real LR crops are cut from an independently drawn batch, :259-266).
    With outdir: training_log.txt (same CSV columns), kernel_iter<N>.npy / kernel_per_band_iter<N>.npy, and the final
    kernel_per_band.npy / kernel_merged.npy.  Returns the history [iters, 5] = (loss_D, loss_G_adv, loss_reg, |gD|, |gG|)."""
    dev = next(G.parameters()).device
    opt_D, opt_G = _adam(D.parameters(), lr_rate), _adam(G.parameters(), lr_rate)
    kd.broadcast_parameters(G)
    kd.broadcast_parameters(D)
    buckets = (kd.GradBucket(G.parameters()), kd.GradBucket(D.parameters()))
    hist = torch.zeros(iters, 5, device=dev)
    prev_k = None
    for t in range(iters):
        hr = load_full_patches(patches, batch_size, dev)                                  # :255-259
        real_ds = crop_patches(load_full_patches(patches, batch_size, dev), lr_crop_size) # :261-266
        vals = kernelgan_single_step(G, D, opt_G, opt_D, hr, real_ds, reg_weight, grad_clip_norm, buckets)
        for j, v in enumerate(vals):
            hist[t, j] = v
        if kernel_log_every and (t + 1) % kernel_log_every == 0:                          # :325-345
            ks_all = G.extract_effective_kernels()
            k_merged = ks_all.mean(dim=0)
            delta = 0.0 if prev_k is None else float(torch.norm(k_merged - prev_k))
            prev_k = k_merged.detach().clone()
            if log is not None:
                km = kernel_metrics(k_merged)
                log(f"[LOG] Iter {t + 1}/{iters} | D: {float(vals[0]):.4f} | G_adv: {float(vals[1]):.4f} | Reg: {float(vals[2]):.4f} "
                    f"| [Kernel] sum={km['k_sum']:.4f} max={km['k_max']:.4f} center_offset={km['center_offset']:.3f} delta_L2={delta:.5f}")
            if save_intermediate and outdir is not None and kd.rank() == 0:
                os.makedirs(outdir, exist_ok=True)
                np.save(os.path.join(outdir, f"kernel_iter{t + 1}.npy"), k_merged.cpu().numpy())
                np.save(os.path.join(outdir, f"kernel_per_band_iter{t + 1}.npy"), ks_all.cpu().numpy())
    h = hist.cpu().numpy()
    if outdir is not None and kd.rank() == 0:
        os.makedirs(outdir, exist_ok=True)
        with open(os.path.join(outdir, "training_log.txt"), "w", encoding="utf-8") as f:  # :156-158, :304-306
            f.write("Iteration,Loss_D,Loss_G_adv,Loss_Reg,Loss_Reg_weighted\n")
            for t in range(iters):
                f.write(f"{t + 1},{h[t, 0]:.6f},{h[t, 1]:.6f},{h[t, 2]:.6f},{reg_weight * h[t, 2]:.6f}\n")
        ks_final = G.extract_effective_kernels().cpu().numpy()                            # :347-353
        np.save(os.path.join(outdir, "kernel_per_band.npy"), ks_final)
        np.save(os.path.join(outdir, "kernel_merged.npy"), ks_final.mean(axis=0))
    return h
