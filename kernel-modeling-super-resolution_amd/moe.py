"""Content-adaptive "mixture of kernels" degradation model on the HIP kernels.

Mirrors muti_kernel/train_gemini.py:14-140 -- same class names, constructor arguments, parameter / buffer names
(state_dict keys `kernel_bank`, `sigma_bank`, `selector.features.{0,1,3,4,6,7}.*`, `selector.classifier.*`, so the
reference's `moe_model.pth` loads unchanged) and the same forward contract:

    out, weights, valid_kernels = model(x, temp=1.0, hard=False)       # train_gemini.py:88-140

Runs on the GPU only.  eval() mode: inference -- BatchNorm uses the running statistics of the checkpoint, folded
into the stride-2 convs (SURVEY 8a': the deployment mode).  train() mode (the nn.Module default): the differentiable
path the reference's trainer runs (train_gemini.py:169-232 never calls .eval(), so BatchNorm uses batch
statistics there): selector convs through Conv3x3Function (HIP forward / dgrad / wgrad), degradation through
DegradeFunction (HIP forward + kmsr_degrade_bwd_f32 for d/d kernel and d/d sigma); softmax / softplus / Gumbel /
einsum over the tiny banks stay torch ops so autograd finishes the chain rule into kernel_bank, sigma_bank and the
selector.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import dist as kd
from .degradation import DegradeFunction, degrade_batch
from .sr import Conv3x3Function, Conv3x3S2Function, conv3x3, pack_conv3x3_weight


class SelectorNet(nn.Module):
    """train_gemini.py:14-39: 3 x [conv3x3 s2 p1 + BN + ReLU] -> AdaptiveAvgPool2d(1) -> Linear(128 -> num_classes)."""

    def __init__(self, in_ch=5, num_classes=10):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(in_ch, 32, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 64, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
            nn.AdaptiveAvgPool2d(1),
        )
        self.classifier = nn.Linear(128, num_classes)
        self._folded = None

    def _fold(self):
        """conv + eval-mode BN -> (packed weights, bias) per layer; rebuilt when a parameter changes."""
        ps = list(self.features.parameters()) + list(self.features.buffers())
        key = tuple((t.data_ptr(), t._version) for t in ps)
        if self._folded is None or self._folded[0] != key:
            layers = []
            for ci, bi in ((0, 1), (3, 4), (6, 7)):
                conv, bn = self.features[ci], self.features[bi]
                inv = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (conv.weight * inv.view(-1, 1, 1, 1)).contiguous()
                b = ((conv.bias - bn.running_mean) * inv + bn.bias).contiguous()
                layers.append((pack_conv3x3_weight(w), b, conv.out_channels))
            self._folded = (key, layers)
        return self._folded[1]

    def forward(self, x):
        L.require_gpu()
        if not x.is_cuda:
            raise L.KmsrError("SelectorNet.forward runs on the GPU only (no CPU fallback)")
        if self.training:          # batch-statistics BatchNorm also under no_grad (the trainer's D step, train_gemini.py:181-183)
            return self._forward_train(x)
        with torch.no_grad():
            return self._forward_infer(x)

    def _forward_train(self, x):
        """Differentiable path, BatchNorm as the module's mode says (batch statistics in train()).  The stride-2 convs run
        natively in both directions (Conv3x3S2Function, round 4: the stride-2 forward kernel; data / weight gradients as stride-1
        problems in pixel-(un)shuffled coordinates on the F(4x4,3x3) kernels) -- until round 3 the stride-1 conv was computed and
        three quarters of it discarded, forward and backward."""
        h = x
        for ci, bi in ((0, 1), (3, 4), (6, 7)):
            conv, bn = self.features[ci], self.features[bi]
            h = Conv3x3S2Function.apply(h, conv.weight, conv.bias)
            h = F.relu(bn(h))
        feat = h.mean(dim=(2, 3))
        return self.classifier(feat)

    def _forward_infer(self, x):
        h = x
        for pw, b, cout in self._fold():
            h = conv3x3(h, pw, cout, bias=b, relu=True, stride=2)
        feat = h.mean(dim=(2, 3))                                        # AdaptiveAvgPool2d(1) + view  (:36-37)
        return F.linear(feat, self.classifier.weight, self.classifier.bias)


class ContentAdaptiveDegradation(nn.Module):
    """train_gemini.py:44-140."""

    def __init__(self, n_kernels=10, n_channels=5, kernel_size=13):
        super().__init__()
        self.n_kernels = n_kernels
        self.k_size = kernel_size
        self.pad = kernel_size // 2
        self.selector = SelectorNet(in_ch=n_channels, num_classes=n_kernels)
        initial_kernels = torch.zeros(n_kernels, n_channels, kernel_size, kernel_size)
        center = kernel_size // 2
        initial_kernels[:, :, center, center] = 1.0
        initial_kernels += torch.randn_like(initial_kernels) * 0.01
        self.kernel_bank = nn.Parameter(initial_kernels)                  # :62
        self.sigma_bank = nn.Parameter(torch.ones(n_kernels, n_channels) * 0.5)   # :66
        self.noise_seed = 0       # (seed, offset) of the counter-based noise stream; offset advances per forward.  Samples are
        #                           addressed globally (dist.sample_base: rank r's batch is samples r*B .. r*B+B-1 of the step), so
        #                           data-parallel ranks draw different eps fields and the fields do not depend on the world size
        self.noise_offset = 0
        self.sample_addr = None   # (first global sample, step) of this rank's batch, set by a trainer whose ranks hold unequal or
        #                           round-robin shards (dist.batch_offsets); None = equal contiguous blocks, rank * B
        self.downscale = 4        # train_gemini.py:134  out[:, :, ::4, ::4]

    def _sample_addr(self, B):
        return self.sample_addr if self.sample_addr is not None else (kd.sample_base(B), 1)

    def get_effective_kernels(self):
        """:68-80 softmax over the 169 taps of every (kernel, band)."""
        B_k, C_k, H_k, W_k = self.kernel_bank.shape
        return F.softmax(self.kernel_bank.view(B_k, C_k, -1), dim=-1).view(B_k, C_k, H_k, W_k)

    def get_effective_sigmas(self):
        """:82-86"""
        return F.softplus(self.sigma_bank)

    def forward(self, x, temp=1.0, hard=False, weights=None):
        """x [B,C,H,W] CUDA float32.  Returns (out [B,C,ceil(H/4),ceil(W/4)], weights [B,K], valid_kernels [K,C,k,k]).
        `weights` may be passed to bypass the Gumbel draw (tests / a uniformly random kernel per sample)."""
        if self.training:
            return self._forward_train(x, temp, hard, weights)
        with torch.no_grad():
            return self._forward_infer(x, temp, hard, weights)

    def _forward_train(self, x, temp, hard, weights):
        """train_gemini.py:88-140 with autograd: every line is the reference's; only the grouped conv + ::4 + noise
        (:117-138) is the fused HIP kernel pair."""
        if weights is None:
            logits = self.selector(x)                                               # :98
            weights = F.gumbel_softmax(logits, tau=temp, hard=hard, dim=1)          # :103
        valid_kernels = self.get_effective_kernels()                                # :107
        valid_sigmas = self.get_effective_sigmas()                                  # :108
        batch_kernels = torch.einsum('bk, kchw -> bchw', weights, valid_kernels)    # :112
        batch_sigmas = torch.einsum('bk, kc -> bc', weights, valid_sigmas)          # :115
        out = DegradeFunction.apply(x.detach(), batch_kernels, batch_sigmas, self.downscale, "zero",
                                    self.noise_seed, self.noise_offset, *self._sample_addr(x.shape[0]))
        self.noise_offset += 1
        return out, weights, valid_kernels

    def _forward_infer(self, x, temp, hard, weights):
        B, C, H, W = x.shape
        if weights is None:
            logits = self.selector(x)                                               # :98
            weights = F.gumbel_softmax(logits, tau=temp, hard=hard, dim=1)          # :103  (torch's device generator)
        valid_kernels = self.get_effective_kernels()                                # :107
        valid_sigmas = self.get_effective_sigmas()                                  # :108
        one_hot = bool(((weights == 0) | (weights == 1)).all()) and bool((weights.sum(dim=1) == 1).all())
        base, step = self._sample_addr(B)
        if one_hot:
            # hard selection: einsum('bk,kchw->bchw') with one-hot weights IS a gather (:112-115)
            out = degrade_batch(x, valid_kernels, self.downscale, down="stride", pad="zero", normalize=False,
                                kernel_idx=weights.argmax(dim=1).to(torch.int32), sigma=valid_sigmas, noise="gaussian",
                                seed=self.noise_seed, offset=self.noise_offset, sample_base=base, sample_step=step,
                                validate_idx=False)        # an argmax over the K logits is in [0, K) by construction
        else:
            batch_kernels = torch.einsum('bk, kchw -> bchw', weights, valid_kernels)   # :112
            batch_sigmas = torch.einsum('bk, kc -> bc', weights, valid_sigmas)        # :115
            out = degrade_batch(x, batch_kernels, self.downscale, down="stride", pad="zero", normalize=False,
                                sigma=batch_sigmas, noise="gaussian", seed=self.noise_seed,
                                offset=self.noise_offset, sample_base=base, sample_step=step)
        self.noise_offset += 1
        return out, weights, valid_kernels


class NoiseEstimator(nn.Module):
    """muti_kernel/networks.py:210-224: learnable per-band sigma, clamp(exp(log_sigma), 1e-5, sigma_max), x + sigma*eps."""

    def __init__(self, channels=5, init_sigma=0.01, sigma_max=0.2):
        super().__init__()
        self.channels = channels
        self.sigma_max = sigma_max
        self.log_sigma = nn.Parameter(torch.log(torch.full(size=(channels,), fill_value=init_sigma)))
        self.noise_seed = 0
        self.noise_offset = 0
        self.sample_addr = None   # see ContentAdaptiveDegradation.sample_addr

    def forward(self, x, eps=None):
        """-> (x + sigma * eps, sigma).  eps: the unit-variance field of the HIP counter-based generator (or a caller-supplied
        one); with autograd enabled the result carries gradients to x and log_sigma like the reference's (:219-224)."""
        sigma = torch.clamp(torch.exp(self.log_sigma), min=1e-5, max=self.sigma_max)      # :222
        one = torch.ones((1, x.shape[1], 1, 1), device=x.device)
        base, step = self.sample_addr if self.sample_addr is not None else (kd.sample_base(x.shape[0]), 1)
        if torch.is_grad_enabled() or eps is not None:
            if eps is None:
                with torch.no_grad():     # identity blur of zeros + unit-sigma Gaussian epilogue = the eps field itself
                    eps = degrade_batch(torch.zeros_like(x), one, 1, down="stride", pad="zero", normalize=False,
                                        sigma=torch.ones((1, x.shape[1]), device=x.device), noise="gaussian",
                                        seed=self.noise_seed, offset=self.noise_offset,
                                        sample_base=base, sample_step=step)
                self.noise_offset += 1
            return x + eps * sigma.view(1, -1, 1, 1), sigma
        # identity blur (1x1 kernel, stride 1) + Gaussian epilogue of the fused kernel
        out = degrade_batch(x, one, 1, down="stride", pad="zero", normalize=False, sigma=sigma.view(1, -1),
                            noise="gaussian", seed=self.noise_seed, offset=self.noise_offset,
                            sample_base=base, sample_step=step)
        self.noise_offset += 1
        return out, sigma
