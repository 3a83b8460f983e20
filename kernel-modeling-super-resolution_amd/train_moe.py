"""Mixture-of-kernels GAN trainer -- the caller of the degradation path in
muti_kernel/train_gemini.py:145-254, with the generator side on the HIP kernels.

    model = ContentAdaptiveDegradation(10, 5).cuda();  D = PatchDiscriminator(5).cuda()
    hist = train_moe(model, D, patches, real_ds, total_iters=5000)      # train_gemini.py:169-232
    save_moe(model, "./moe_kernels")                                     # train_gemini.py:237-251 (same file names)

What runs where: the fused degradation forward, its backward w.r.t. kernels / sigmas (kmsr_degrade_bwd_f32) and the
selector's 3x3 convs (forward, dgrad, wgrad) are HIP kernels behind the C ABI.  The PatchDiscriminator's convs (zero-padded
7x7 5->64 and 1x1 64->64 / 64->1 on the LR patches) run on the small-channel HIP family (kmsr_convk_fwd / dgrad / wgrad_f32,
shared weights + bias); its spectral normalisation, BatchNorm and LeakyReLU are the stock torch modules (weight-side /
elementwise on [B,64,h,w] LR maps).  Losses are the reference's formulas
(muti_kernel/loss.py:6-41).  Data parallel: every rank trains on its own patches; gradients of both networks are
averaged through kmsr_amd.dist.GradBucket (one flat all-reduce each per step).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import dist as kd


def lsgan_d_loss(pred_real, pred_fake):
    """muti_kernel/loss.py:6-9: least-squares GAN, real -> 1, fake -> 0, each term halved."""
    return 0.5 * (pred_real - 1.0).pow(2).mean() + 0.5 * pred_fake.pow(2).mean()


def lsgan_g_loss(pred_fake):
    """muti_kernel/loss.py:12-13"""
    return 0.5 * (pred_fake - 1.0).pow(2).mean()


def kernel_regularization(k, alpha=0.5, beta=0.5, gamma=5.0, delta=1.0):
    """muti_kernel/loss.py:16-41 for one 2-D kernel: alpha*(sum-1)^2 + beta*(energy on the 4 border lines, corners
    counted twice) + gamma*sum(sqrt(relu(k))) + delta*(squared distance of the centre of mass of relu(k)+1e-12 from
    the geometric centre)."""
    kh, kw = k.shape
    unit_sum = (k.sum() - 1.0) ** 2
    border = k[0].pow(2).sum() + k[-1].pow(2).sum() + k[:, 0].pow(2).sum() + k[:, -1].pow(2).sum()
    pos = k.clamp(min=0)
    sparse = pos.sqrt().sum()
    mass = pos + 1e-12
    rows = torch.arange(kh, device=k.device, dtype=torch.float32).view(kh, 1)
    cols = torch.arange(kw, device=k.device, dtype=torch.float32).view(1, kw)
    total = mass.sum()
    cy, cx = (rows * mass).sum() / total, (cols * mass).sum() / total
    centre = (cy - (kh - 1) / 2.0) ** 2 + (cx - (kw - 1) / 2.0) ** 2
    return alpha * unit_sum + beta * border + gamma * sparse + delta * centre


class _HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters (so spectral_norm and state_dict keys are the stock ones); the convolution itself (zero padding
    k//2, stride 1) and both of its gradients are the HIP kernels behind kmsr_convk_* (one weight set shared by the batch)."""

    def forward(self, x):
        from .kernelgan import ConvKFunction
        k = self.kernel_size[0]
        assert self.stride == (1, 1) and self.padding == (k // 2, k // 2) and self.kernel_size == (k, k)
        w = self.weight.permute(1, 2, 3, 0).reshape(self.in_channels, k * k, self.out_channels)      # [ci][tap][co]
        return ConvKFunction.apply(x, w, self.bias, k, "zero")


class PatchDiscriminator(nn.Module):
    """muti_kernel/networks.py:250-268: SN-conv7x7(in->64) LReLU(0.2), 4 x [SN-conv1x1(64->64) BN LReLU], SN-conv1x1(64->1).
    Same module indices -> same state_dict keys (net.0.weight_orig, net.0.weight_u, net.3.running_mean, ...)."""

    def __init__(self, in_ch=5, base_ch=64, num_blocks=4):
        super().__init__()
        layers = [spectral_norm(_HipConv2d(in_ch, base_ch, 7, 1, 3)), nn.LeakyReLU(0.2, inplace=True)]
        for _ in range(num_blocks):
            layers += [spectral_norm(_HipConv2d(base_ch, base_ch, 1, 1, 0)), nn.BatchNorm2d(base_ch),
                       nn.LeakyReLU(0.2, inplace=True)]
        layers.append(spectral_norm(_HipConv2d(base_ch, 1, 1, 1, 0)))
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


def moe_regularizer(current_kernels, alpha=0.5):
    """train_gemini.py:206-217: kernel_regularization of the bank's MEAN kernel per band, averaged over bands."""
    avg = current_kernels.mean(dim=0)
    return torch.stack([kernel_regularization(avg[c], alpha=alpha) for c in range(avg.shape[0])]).mean()


def train_moe(model, D, patches, real_ds, total_iters=5000, lr=1e-4, temp_range=(5.0, 0.5), log_every=100, log=print,
              batch_fn=None):
    """train_gemini.py:156-232.  patches [B,C,256,256] / real_ds [B,C,64,64] CUDA tensors (the reference trains on a fixed
    dummy batch, :164-165), or batch_fn(i) -> (patches, real_ds) to feed real data.  Returns the per-iteration history
    [(loss_D, loss_G_adv, reg_loss)] as a float32 array (read back once at the end: no host sync inside the loop)."""
    model.train()
    D.train()
    opt_G = torch.optim.Adam(model.parameters(), lr=lr)                      # :156
    opt_D = torch.optim.Adam(D.parameters(), lr=lr)                          # :157
    temps = np.linspace(temp_range[0], temp_range[1], total_iters)           # :161
    bucket_G = kd.GradBucket(model.parameters())
    bucket_D = kd.GradBucket(D.parameters())
    kd.broadcast_parameters(model)
    kd.broadcast_parameters(D)
    hist = torch.zeros(total_iters, 3, device=patches.device if patches is not None else "cuda")
    for i in range(total_iters):
        if batch_fn is not None:
            patches, real_ds = batch_fn(i)
        if i == 0 and getattr(model, "sample_addr", None) is None:
            kd.assert_equal_batch(patches.shape[0], device=patches.device)     # noise is addressed as global sample rank * B + n
        temp = float(temps[i])
        # ---- discriminator step (:178-189)
        with torch.no_grad():
            fake_ds, weights, _ = model(patches, temp=temp, hard=False)
        loss_D = lsgan_d_loss(D(real_ds), D(fake_ds.detach()))
        bucket_D.zero()
        loss_D.backward()
        bucket_D.all_reduce_mean()
        opt_D.step()
        # ---- generator step: selector + kernel / sigma banks (:194-222)
        fake_ds, weights, current_kernels = model(patches, temp=temp, hard=False)
        loss_G_adv = lsgan_g_loss(D(fake_ds))
        reg_loss = moe_regularizer(current_kernels)
        bucket_G.zero()
        (loss_G_adv + reg_loss).backward()     # D's gradients from this pass are never applied (zeroed at the next D step)
        bucket_G.all_reduce_mean()
        opt_G.step()
        hist[i, 0], hist[i, 1], hist[i, 2] = loss_D.detach(), loss_G_adv.detach(), reg_loss.detach()
        if log is not None and log_every and i % log_every == 0:
            counts = torch.bincount(weights.argmax(dim=1), minlength=model.n_kernels)     # :226-229
            log(f"Iter {i} | Temp: {temp:.2f} | D: {float(loss_D.detach()):.3f} | Selection Dist: {counts.cpu().numpy()}")
    return hist.cpu().numpy()


def save_moe(model, save_dir="./moe_kernels"):
    """train_gemini.py:237-251: kernel_<i>.npy [C,13,13], sigma_<i>.npy [C], moe_model.pth (state_dict)."""
    os.makedirs(save_dir, exist_ok=True)
    with torch.no_grad():
        kernels = model.get_effective_kernels().cpu().numpy()
        sigmas = model.get_effective_sigmas().cpu().numpy()
    for i in range(kernels.shape[0]):
        np.save(os.path.join(save_dir, f"kernel_{i}.npy"), kernels[i])
        np.save(os.path.join(save_dir, f"sigma_{i}.npy"), sigmas[i])
    torch.save(model.state_dict(), os.path.join(save_dir, "moe_model.pth"))
