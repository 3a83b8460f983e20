"""KernelGAN generators: per-patch kernel estimation on the GPU (BASELINE config 5).

Mirrors, with identical class names, constructor arguments and parameter names (so state_dicts interchange):
    ConditionEncoder                       muti_kernel/networks.py:10-58
    DynamicMultiBandLinearGenerator        muti_kernel/networks.py:61-207   (extract_effective_kernels, extract_merged_kernel)
    MultiBandLinearGenerator               single_kernel/networks.py:8-168  (Gaussian init, extract_*)
    bulk_extract_kernels                   muti_kernel/train.py:118-188     (array-level: patches in, kernels out)

extract_effective_kernels is ONE launch of kmsr_compose_kernels_f32 (one workgroup per (sample, band)) instead of
~20 000 tiny F.conv2d calls per sample.  The generators' forward (reflect-padded 6-layer chain, used only by the
GAN trainer) is SURVEY 8f N1 "next" and is not implemented here; calling it raises.
"""
import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .sr import conv3x3, pack_conv3x3_weight

KS = (7, 5, 3, 1, 1, 1)


def compose_kernels(weights, scales, co, ci, ks):
    """weights [C, sum_l co*ci*k*k] (CUDA), scales [B, C, sum_l co] or None -> [B, C, K, K] (B = 1 if scales is None)."""
    lib = L.load()
    L.require_gpu()
    weights = weights.detach().to(torch.float32).contiguous()
    assert weights.is_cuda and weights.ndim == 2
    Cc = weights.shape[0]
    B = 1 if scales is None else scales.shape[0]
    n = len(ks)
    K = 1 + sum(k - 1 for k in ks)
    assert weights.shape[1] == sum(o * i * k * k for o, i, k in zip(co, ci, ks)), "weight block size does not match the layer list"
    if scales is not None:
        scales = scales.detach().to(torch.float32).contiguous()
        assert scales.shape == (B, Cc, sum(co)), f"scales must be [B,C,{sum(co)}], got {tuple(scales.shape)}"
    out = torch.empty((B, Cc, K, K), device=weights.device, dtype=torch.float32)
    arr = lambda v: (C.c_int32 * n)(*[int(t) for t in v])
    with torch.cuda.device(weights.device):
        L.check(lib.kmsr_compose_kernels_f32(L.ptr(weights), L.ptr(scales), B, Cc, n, arr(co), arr(ci), arr(ks),
                                             L.ptr(out), L.current_stream()))
    return out


class ConditionEncoder(nn.Module):
    """muti_kernel/networks.py:10-58: conv3x3(5->32) ReLU, conv3x3 s2 (32->64) ReLU, conv3x3 s2 (64->64) ReLU,
    spatial mean, Linear(64 -> in_ch*161); scales = 1 + scale_gain * tanh(.)"""

    def __init__(self, in_ch: int = 5, mid_ch: int = 32, ks: Sequence[int] = KS, scale_gain: float = 0.1) -> None:
        super().__init__()
        self.in_ch, self.mid_ch, self.ks = in_ch, mid_ch, list(ks)
        self.layer_out_channels = [mid_ch, mid_ch, mid_ch, mid_ch, mid_ch, 1]
        self.total_scales = in_ch * sum(self.layer_out_channels)
        self.scale_gain = scale_gain
        self.encoder = nn.Sequential(
            nn.Conv2d(in_ch, 32, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(32, 64, kernel_size=3, stride=2, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(64, 64, kernel_size=3, stride=2, padding=1), nn.ReLU(inplace=True),
        )
        self.fc = nn.Linear(64, self.total_scales)
        self._packed = None

    def _packs(self):
        convs = [self.encoder[0], self.encoder[2], self.encoder[4]]
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, [pack_conv3x3_weight(c.weight) for c in convs])
        return convs, self._packed[1]

    @torch.no_grad()
    def scales_tensor(self, x):
        """[B, in_ch, 161]: for every band the 6 layers' scales concatenated (the reference returns the same numbers
        as a list[band][layer] of [B,out_c] tensors, :46-58)."""
        convs, packs = self._packs()
        h = x
        for conv, pw in zip(convs, packs):
            h = conv3x3(h, pw, conv.out_channels, bias=conv.bias, relu=True, stride=conv.stride[0])
        h = h.mean(dim=(2, 3))
        raw = torch.nn.functional.linear(h, self.fc.weight, self.fc.bias)
        return (1.0 + self.scale_gain * torch.tanh(raw)).view(x.shape[0], self.in_ch, sum(self.layer_out_channels))

    def forward(self, x):
        s = self.scales_tensor(x)
        out, st = [], 0
        per_layer = []
        for oc in self.layer_out_channels:
            per_layer.append((st, st + oc))
            st += oc
        return [[s[:, band, a:b] for a, b in per_layer] for band in range(self.in_ch)]


class DynamicMultiBandLinearGenerator(nn.Module):
    """muti_kernel/networks.py:61-207"""

    def __init__(self, in_ch: int = 5, mid_ch: int = 32, ks: Sequence[int] = KS, scale_gain: float = 0.1) -> None:
        super().__init__()
        self.in_ch, self.mid_ch, self.ks = in_ch, mid_ch, list(ks)
        self.layer_out_channels = [mid_ch, mid_ch, mid_ch, mid_ch, mid_ch, 1]
        bands = []
        for _ in range(in_ch):
            ws, in_c = nn.ParameterList(), 1
            for idx, ksize in enumerate(self.ks):
                out_c = self.layer_out_channels[idx]
                ws.append(nn.Parameter(torch.randn(out_c, in_c, ksize, ksize) * 0.01))
                in_c = out_c
            bands.append(ws)
        self.weight_bands = nn.ModuleList(bands)
        self.condition_encoder = ConditionEncoder(in_ch=in_ch, mid_ch=mid_ch, ks=ks, scale_gain=scale_gain)

    def _layer_dims(self):
        co = self.layer_out_channels
        ci = [1] + co[:-1]
        return co, ci, self.ks

    def _flat_weights(self):
        return torch.stack([torch.cat([w.detach().reshape(-1) for w in band]) for band in self.weight_bands])

    def forward(self, x):
        raise NotImplementedError("generator forward (GAN training path) is SURVEY 8f N1 'next'; "
                                  "use extract_effective_kernels for kernel estimation")

    @torch.no_grad()
    def extract_effective_kernels(self, x: Optional[torch.Tensor] = None, reduce_batch: bool = True) -> torch.Tensor:
        """:135-202.  x given -> dynamic kernels from its condition scales ([B,C,K,K], or the batch mean [C,K,K]);
        x None -> unit scales ([C,K,K])."""
        co, ci, ks = self._layer_dims()
        scales = None if x is None else self.condition_encoder.scales_tensor(x)
        k = compose_kernels(self._flat_weights(), scales, co, ci, ks)
        return k.mean(dim=0) if reduce_batch else k

    @torch.no_grad()
    def extract_merged_kernel(self, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.extract_effective_kernels(x=x, reduce_batch=True).mean(dim=0)      # :204-207


class MultiBandLinearGenerator(nn.Module):
    """single_kernel/networks.py:8-168 (static per-band chains, Gaussian initialisation :55-81)."""

    def __init__(self, in_ch: int = 5, mid_ch: int = 32, gaussian_sigma: float = 2.0):
        super().__init__()
        self.in_ch, self.mid_ch, self.gaussian_sigma = in_ch, mid_ch, gaussian_sigma
        ks = list(KS)
        chains = []
        for _ in range(in_ch):
            convs = nn.ModuleList()
            convs.append(nn.Conv2d(1, mid_ch, ks[0], 1, ks[0] // 2, bias=False, padding_mode='reflect'))
            for ksize in ks[1:-1]:
                convs.append(nn.Conv2d(mid_ch, mid_ch, ksize, 1, ksize // 2, bias=False, padding_mode='reflect'))
            convs.append(nn.Conv2d(mid_ch, 1, ks[-1], 1, ks[-1] // 2, bias=False, padding_mode='reflect'))
            chains.append(convs)
        self.chains = nn.ModuleList(chains)
        self._init_gaussian_kernels()

    @torch.no_grad()
    def _init_gaussian_kernels(self) -> None:
        for convs in self.chains:
            first = convs[0]
            size = first.kernel_size[0]
            coords = torch.arange(size, dtype=first.weight.dtype) - (size - 1) * 0.5
            yy, xx = torch.meshgrid(coords, coords, indexing='ij')
            g = torch.exp(-(xx ** 2 + yy ** 2) / (2 * self.gaussian_sigma ** 2))
            g = g / g.sum()
            first.weight.zero_()
            first.weight.copy_(g.expand(self.mid_ch, 1, size, size))
            for conv in convs[1:-1]:
                conv.weight.zero_()
                _, _, kH, kW = conv.weight.shape
                for i in range(min(conv.out_channels, conv.in_channels)):
                    conv.weight[i, i, kH // 2, kW // 2] = 1.0
            convs[-1].weight.zero_()
            convs[-1].weight.fill_(1.0 / self.mid_ch)

    def forward(self, x):
        raise NotImplementedError("generator forward (GAN training path) is SURVEY 8f N1 'next'")

    @torch.no_grad()
    def extract_effective_kernels(self) -> torch.Tensor:
        """:108-157 -> [C,K,K]"""
        co = [self.mid_ch] * 5 + [1]
        ci = [1] + co[:-1]
        flat = torch.stack([torch.cat([c.weight.detach().reshape(-1) for c in convs]) for convs in self.chains])
        return compose_kernels(flat, None, co, ci, list(KS))[0]

    @torch.no_grad()
    def extract_merged_kernel(self) -> torch.Tensor:
        return self.extract_effective_kernels().mean(dim=0)


@torch.no_grad()
def bulk_extract_kernels(patches, generator: DynamicMultiBandLinearGenerator, patch_size: int = 256, batch_size: int = 8,
                         rank: int = 0, world_size: int = 1):
    """Array-level muti_kernel/train.py:118-188: for each [C,H,W] patch (numpy / tensor) centre-crop to patch_size
    (:160-166), run extract_effective_kernels(batch, reduce_batch=False) and return one [C,K,K] kernel per patch
    (the reference np.saves them as kernel_<name>.npy, :180-185).  Patches are independent: with world_size > 1 each rank
    takes patches[rank::world_size] (no collective); the returned list holds (global_index, kernel) pairs."""
    dev = L.require_gpu()
    idxs = list(range(rank, len(patches), world_size))
    out = []
    for s in range(0, len(idxs), batch_size):
        chunk = idxs[s:s + batch_size]
        batch = []
        for i in chunk:
            a = patches[i]
            a = torch.from_numpy(a) if isinstance(a, np.ndarray) else a
            _, H, W = a.shape
            if H < patch_size or W < patch_size:
                raise ValueError(f"patch {i} is {H}x{W}, smaller than {patch_size}")
            t, l = (H - patch_size) // 2, (W - patch_size) // 2
            batch.append(a[:, t:t + patch_size, l:l + patch_size].to(dev, torch.float32))
        ks = generator.extract_effective_kernels(torch.stack(batch), reduce_batch=False)
        out.extend((i, ks[j]) for j, i in enumerate(chunk))
    return out
