"""GPU tests of the SR training step: HIP dgrad / wgrad against torch autograd on the CPU (float64), and one full
optimisation step of SRNet against the stock-torch oracle module with identical weights and data."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import kmsr_amd
from kmsr_amd.sr import Conv3x3Function, SRNet, conv3x3, conv3x3_wgrad, pack_conv3x3_weight
from oracle.sr_torch import SRNetTorch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("N,Cin,H,W,Cout", [(2, 64, 20, 40, 64), (1, 3, 17, 33, 64), (2, 64, 16, 32, 3), (1, 64, 9, 70, 256),
                                            (3, 20, 7, 5, 12), (1, 130, 6, 34, 70),
                                            (2, 64, 33, 45, 5), (1, 100, 5, 70, 7), (1, 8, 1, 1, 1), (2, 64, 6, 64, 20),   # small-Cout kernels (<= 7, <= 32)
                                            # even H, W % 4 == 0 and Cout > 32: the Winograd-domain kernel (partial column blocks, one tile
                                            # row, several channel groups on either side, more stages than workgroups)
                                            (1, 5, 2, 4, 64), (2, 64, 4, 68, 256), (1, 70, 10, 28, 40), (3, 64, 64, 96, 64), (9, 64, 128, 128, 64),
                                            # W % 4 == 0, H >= 4, Cout > 32: the F(4x4,3x3)-domain kernel (round 4) -- odd H and H % 4 != 0 (partial
                                            # last tile row), partial 32-column blocks, channel tails on both sides (ci groups of 32, co groups of 64,
                                            # odd Cout: the last dy plane pair is half empty), fewer stages than workgroups, several images per range
                                            (1, 64, 4, 4, 64), (2, 33, 7, 36, 65), (1, 96, 13, 100, 33), (5, 64, 32, 32, 64), (2, 7, 64, 256, 40),
                                            (40, 64, 16, 32, 64)])
def test_wgrad_and_dgrad_vs_torch(N, Cin, H, W, Cout, gpu_device):
    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) / (3 * Cin ** 0.5)).requires_grad_()
    dy = torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, None, padding=1)
    y.backward(dy)
    dw = conv3x3_wgrad(x.detach().float().cuda(), dy.float().cuda())
    assert tuple(dw.shape) == (Cout, Cin, 3, 3)
    assert _rel(dw, w.grad) < 2e-5, f"wgrad rel err {_rel(dw, w.grad):.3e}"
    dwd = conv3x3_wgrad(x.detach().float().cuda(), dy.float().cuda(), direct=True)      # the 36-product kernel on the same shape
    assert _rel(dwd, w.grad) < 2e-5, f"direct wgrad rel err {_rel(dwd, w.grad):.3e}"
    dw2 = conv3x3_wgrad(x.detach().float().cuda(), dy.float().cuda(), wino2=True)       # F(2x2,3x3) domain where it applies (else direct)
    assert _rel(dw2, w.grad) < 2e-5, f"F(2x2) wgrad rel err {_rel(dw2, w.grad):.3e}"
    dx = conv3x3(dy.float().cuda(), pack_conv3x3_weight(w.detach().float().cuda(), transpose_flip=True), Cin)
    assert _rel(dx, x.grad) < 2e-5, f"dgrad rel err {_rel(dx, x.grad):.3e}"


def test_wgrad_is_deterministic(gpu_device):
    x = torch.randn(4, 64, 64, 64, device="cuda")
    dy = torch.randn(4, 64, 64, 64, device="cuda")
    a = conv3x3_wgrad(x, dy)
    for _ in range(3):
        assert torch.equal(conv3x3_wgrad(x, dy), a)          # fixed-order slab reduction, no float atomics


def test_autograd_function_matches_torch(gpu_device):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 12, 36, generator=g)
    w = torch.randn(24, 16, 3, 3, generator=g) * 0.1
    b = torch.randn(24, generator=g)
    for relu in (False, True):
        xr, wr, br = (t.clone().double().requires_grad_() for t in (x, w, b))
        yr = F.conv2d(xr, wr, br, padding=1)
        yr = F.relu(yr) if relu else yr
        (yr ** 2).sum().backward()
        xc, wc, bc = (t.clone().cuda().requires_grad_() for t in (x, w, b))
        yc = Conv3x3Function.apply(xc, wc, bc, relu)
        (yc ** 2).sum().backward()
        assert _rel(yc, yr) < 1e-5 and _rel(xc.grad, xr.grad) < 2e-5 and _rel(wc.grad, wr.grad) < 2e-5 and _rel(bc.grad, br.grad) < 2e-5


def test_autograd_function_rejects_unsupported_epilogues(gpu_device):
    """relu + residual (mask not recoverable from the saved output) and residual + pixel_shuffle are refused, not silently wrong."""
    x = torch.randn(1, 16, 8, 8, device="cuda", requires_grad=True)
    w = torch.randn(16, 16, 3, 3, device="cuda", requires_grad=True)
    r = torch.randn(1, 16, 8, 8, device="cuda", requires_grad=True)
    with pytest.raises(ValueError):
        Conv3x3Function.apply(x, w, None, True, r)
    with pytest.raises(ValueError):
        Conv3x3Function.apply(x, w, None, False, torch.randn(1, 4, 16, 16, device="cuda"), True)
    # residual alone: d/d residual is the incoming gradient
    y = Conv3x3Function.apply(x, w, None, False, r)
    y.sum().backward()
    assert torch.equal(r.grad, torch.ones_like(r))


def test_srnet_training_step_matches_oracle(gpu_device):
    """One Adam step on identical weights / data: loss, every gradient and the updated weights agree with the
    stock-torch oracle on the CPU."""
    torch.manual_seed(0)
    ref = SRNetTorch(3, 32, 2, 4)
    net = SRNet(3, 32, 2, 4)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    g = torch.Generator().manual_seed(1)
    lr_img = torch.rand(2, 3, 24, 20, generator=g)
    hr_img = torch.rand(2, 3, 96, 80, generator=g)
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.5, 0.999))     # the reference's GAN Adam betas (muti_kernel/train.py:229)
    opt_n = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    loss_r = F.l1_loss(ref(lr_img), hr_img)
    loss_r.backward()
    loss_n = F.l1_loss(net(lr_img.cuda()), hr_img.cuda())
    loss_n.backward()
    assert abs(float(loss_n) - float(loss_r)) < 1e-5 * abs(float(loss_r))
    for (k, pr), (_, pn) in zip(ref.named_parameters(), net.named_parameters()):
        assert _rel(pn.grad, pr.grad) < 5e-4, f"grad {k}: {_rel(pn.grad, pr.grad):.3e}"
    opt_r.step()
    opt_n.step()
    for (k, pr), (_, pn) in zip(ref.named_parameters(), net.named_parameters()):
        assert torch.allclose(pn.detach().cpu(), pr.detach(), rtol=1e-3, atol=2e-5), k
    # the inference path picks up the updated weights (pack cache keyed on the parameter version)
    with torch.no_grad():
        assert _rel(net(lr_img.cuda()), ref(lr_img)) < 1e-4


@pytest.mark.parametrize("N,C,H,W", [(2, 3, 6, 8), (1, 64, 40, 36), (3, 5, 33, 4), (1, 2, 5, 6)])
def test_pixel_unshuffle2_sum_vs_torch(N, C, H, W, gpu_device):
    """kmsr_pixel_unshuffle2_sum_f32 (backward of the fused PixelShuffle store): the permutation is exact, the channel sums are
    fp32 sums in a fixed order (run-to-run identical); W % 4 != 0 takes the torch path."""
    from kmsr_amd.sr import pixel_unshuffle2_sum
    g = torch.randn(N, C, 2 * H, 2 * W, generator=torch.Generator().manual_seed(N + C + H)).cuda()
    gu, s = pixel_unshuffle2_sum(g)
    ref = F.pixel_unshuffle(g, 2)
    assert torch.equal(gu, ref)
    want = ref.double().sum(dim=(0, 2, 3))
    assert float((s.double() - want).abs().max()) <= 2e-6 * float(ref.abs().double().sum(dim=(0, 2, 3)).max())
    gu2, s2 = pixel_unshuffle2_sum(g)
    assert torch.equal(s, s2)


@pytest.mark.parametrize("N,Cin,H,W,Cout", [(2, 5, 16, 64, 64), (1, 3, 34, 132, 64), (3, 7, 8, 8, 48), (1, 5, 64, 256, 64)])
def test_smallci_unshuffle_equals_conv_then_unshuffle(N, Cin, H, W, Cout, gpu_device):
    """kmsr_conv3x3_smallci_unshuffle_fwd_f32: the small-Cin conv stored pixel-unshuffled is BIT-IDENTICAL to the same kernel's normal
    output passed through pixel_unshuffle(2), and its channel sums match a float64 reduction (ragged tiles, several images, Cout < 64)."""
    from kmsr_amd.sr import conv3x3_smallci_unshuffle
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = torch.nn.functional.pixel_unshuffle(conv3x3(x, w, Cout, bias=b, algo="smallci"), 2)
    got, sums = conv3x3_smallci_unshuffle(x, w, bias=b)
    assert got.shape == want.shape and torch.equal(got, want)
    ref = want.double().sum(dim=(0, 2, 3))
    assert float((sums.double() - ref).abs().max()) <= 2e-6 * float(want.double().abs().sum(dim=(0, 2, 3)).max() + 1)
    assert torch.equal(conv3x3_smallci_unshuffle(x, w, bias=b)[1], sums)          # fixed-order reduction: run-to-run identical
    # the store-only instance (chan_sums = NULL) writes the same tensor, and the closed form of the channel sums (parity / border sums of x
    # times the weights: sr.unshuffled_conv_channel_sums) agrees with a float64 reduction of the bias-free convolution
    got0, s0 = conv3x3_smallci_unshuffle(x, w, sums="closed_form")
    want0 = torch.nn.functional.pixel_unshuffle(conv3x3(x, w, Cout, algo="smallci"), 2)
    assert torch.equal(got0, want0)
    ref0 = want0.double().sum(dim=(0, 2, 3))
    assert float((s0.double() - ref0).abs().max()) <= 2e-6 * float(want0.double().abs().sum(dim=(0, 2, 3)).max() + 1)
    assert torch.equal(conv3x3_smallci_unshuffle(x, w, sums="closed_form")[1], s0)
    assert conv3x3_smallci_unshuffle(x, w, sums=None)[1] is None


def test_up_tail_node_matches_separate_nodes(gpu_device):
    """SRNet's training graph with the last upsampler conv + tail as ONE node (tail data gradient written un-shuffled, no
    pixel_unshuffle2_sum pass) against the graph of separate Conv3x3Function nodes: same loss, gradients equal within fp32 rounding
    of the bias-gradient sums (data / weight gradients bit-identical: the same kernels on a permuted store)."""
    torch.manual_seed(5)
    net = SRNet(5, 64, 2, 4).cuda().train()
    x = torch.rand(2, 5, 16, 32, device="cuda") * 80 + 20
    tgt = torch.rand(2, 5, 64, 128, device="cuda") * 80 + 20
    grads = []
    for fuse in (True, False):
        net.fuse_up_tail_backward = fuse
        net.zero_grad()
        loss = F.l1_loss(net(x), tgt)
        loss.backward()
        grads.append((float(loss), {k: p.grad.clone() for k, p in net.named_parameters()}))
    assert grads[0][0] == grads[1][0]
    for k in grads[0][1]:
        a, b = grads[0][1][k], grads[1][1][k]
        if k.endswith("weight") or not k.startswith("up.1"):
            assert torch.equal(a, b), k
        else:
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max() + 1e-30), k


@pytest.mark.parametrize("N,Cin,H,W,Cout", [(2, 5, 64, 64, 32), (2, 32, 32, 32, 64), (1, 64, 16, 24, 128), (1, 32, 17, 30, 64), (3, 5, 9, 9, 32)])
def test_conv3x3_stride2_function_vs_torch(N, Cin, H, W, Cout, gpu_device):
    """sr.Conv3x3S2Function (stride-2 forward kernel; data gradient = one F(4x4) / direct launch with the fused PixelShuffle store on
    re-indexed weights; weight gradient = the stride-1 kernels on the pixel-unshuffled input) against torch autograd in float64: the
    SelectorNet / ConditionEncoder shapes, odd sizes (cropped data gradient, padded un-shuffle), widths that are not multiples of 8."""
    from kmsr_amd.sr import Conv3x3S2Function
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().double().requires_grad_() for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xc, wc, bc = (t.clone().cuda().requires_grad_() for t in (x, w, b))
    yc = Conv3x3S2Function.apply(xc, wc, bc)
    yc.backward(gy.cuda())
    assert yc.shape == yr.shape and _rel(yc, yr) < 2e-5
    assert _rel(xc.grad, xr.grad) < 2e-5 and _rel(wc.grad, wr.grad) < 2e-5 and _rel(bc.grad, br.grad) < 2e-5


@pytest.mark.parametrize("shape", [(2, 5, 64, 64), (1, 3, 7, 11), (3, 1, 1, 1), (1, 5, 256, 1024)])
def test_l1_loss_value_and_gradient_vs_torch(shape, gpu_device):
    """kmsr_l1_loss_f32 (value and sign(a - b) / n in one pass; the SR training step's loss) against torch.nn.functional.l1_loss and its
    autograd: value to fp32 summation rounding, gradient exactly; an upstream factor goes through kmsr_scale_by_device_scalar_f32; exact zeros
    of a - b get a zero gradient; deterministic; the target gets no gradient."""
    from kmsr_amd.sr import l1_loss
    g = torch.Generator().manual_seed(sum(shape))
    a = torch.randn(*shape, generator=g).cuda().requires_grad_()
    b = torch.randn(*shape, generator=g).cuda()
    with torch.no_grad():
        b.view(-1)[::5] = a.detach().view(-1)[::5]                 # exact zeros of the difference
    ref_a = a.detach().clone().requires_grad_()
    ref = torch.nn.functional.l1_loss(ref_a, b)
    got = l1_loss(a, b)
    assert got.shape == () and abs(float(got) - float(ref)) <= 2e-6 * float(ref) + 1e-12
    ref.backward()
    got.backward()
    assert torch.equal(a.grad, ref_a.grad)                          # sign(a - b) / n, bit for bit
    assert float(a.grad.view(-1)[::5].abs().max()) == 0.0
    a.grad, ref_a.grad = None, None
    got2 = l1_loss(a, b)
    (2.5 * got2).backward()                                          # the upstream factor is applied on the device
    (2.5 * torch.nn.functional.l1_loss(ref_a, b)).backward()
    assert torch.equal(got2, got)
    assert float((a.grad - ref_a.grad).abs().max()) <= 1e-6 * float(ref_a.grad.abs().max())
    assert l1_loss(a.detach(), b).requires_grad is False


@pytest.mark.parametrize("Cout,Cin", [(64, 64), (256, 64), (64, 256), (40, 24), (5, 64)])
def test_wino4_pack_transpose_flip_is_the_pack_of_the_flipped_tensor(Cout, Cin, gpu_device):
    """kmsr_conv3x3_wino4_pack_weights_ex_f32(transpose_flip=1) (round 4: the data-gradient weights of a layer packed straight from its forward
    tensor) is bit-identical to flipping / transposing the tensor with torch and packing the result."""
    from kmsr_amd.sr import pack_conv3x3_weight_wino4
    w = torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(Cout + Cin)).cuda()
    want = pack_conv3x3_weight_wino4(w.flip(2, 3).transpose(0, 1).contiguous())
    got = pack_conv3x3_weight_wino4(w, transpose_flip=True)
    assert got.shape == want.shape and torch.equal(got, want)
    assert torch.equal(pack_conv3x3_weight_wino4(w, transpose_flip=False), pack_conv3x3_weight_wino4(w))
