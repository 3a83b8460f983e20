"""BASELINE.json configs 2-5 at their REAL sizes on the GPU (round-1 verdict: the only parity hole).

The CPU oracles finish the full sizes in minutes, not seconds, so every full-size run is pinned by
  * an oracle comparison on a window / crop / sub-batch of the SAME full-size tensors (float64 torch on the CPU for single
    layers, the stock-torch oracle module oracle/sr_torch.py for the network, oracle.muti_extract_effective_kernels for
    the KernelGAN composition), and
  * size-independent properties: two independent kernels agree, tiles are independent of their neighbours beyond the
    receptive field (bit-exact), the batch gradient is the mean of the sub-batch gradients, shards == full, determinism.
The SR network has no counterpart in the reference: its parity is HIP vs the stock-torch module on identical
weights.  Tolerances: 2e-5 of the output scale per layer (fp32 re-association only), 1e-4 for the network (north_star),
one bf16 ulp for the bf16 layer.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import kmsr_amd
import oracle
from conftest import assert_close, golden
from kmsr_amd import _lib as L
from kmsr_amd.sr import (Conv3x3Function, SRNet, conv3x3, conv3x3_wgrad, pack_conv3x3_weight, pack_conv3x3_weight_wino_vreg)
from oracle.sr_torch import SRNetTorch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _elementwise(out, ref, tol, what):
    """|out - ref| <= tol * (|ref| + mean|ref|) for every element (the single-layer criterion of test_conv_gpu.py)."""
    ref = ref.double().cpu()
    err = (out.double().cpu() - ref).abs()
    scale = float(ref.abs().mean()) + 1e-30
    bad = err > tol * (ref.abs() + scale)
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} / {bad.numel()} out of tolerance, max err {float(err.max()):.3e}"


# ------------------------------------------------------------------------------------------------ config 2
def test_config2_dma_winograd_fullsize_windows_vs_float64(gpu_device):
    """The kernel the headline runs (persistent LDS-DMA Winograd, body layer 64->64 at [16,64,256,256], ReLU and residual epilogues)
    against a float64 convolution on windows of the full-size result: the four image corners, an interior window that
    straddles workgroup-tile seams (tiles are 4 rows x 64 columns), and the last image of the batch."""
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(16, 64, 256, 256, device="cuda", generator=g)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) / 24.0
    b = torch.randn(64, device="cuda", generator=g)
    r = torch.randn(16, 64, 256, 256, device="cuda", generator=g)
    pw = pack_conv3x3_weight_wino_vreg(w)
    y_relu = conv3x3(x, pw, 64, bias=b, relu=True, algo="winograd_vreg")
    y_res = conv3x3(x, pw, 64, bias=b, residual=r, algo="winograd_vreg")
    wd, bd = w.double().cpu(), b.double().cpu()
    windows = [(0, 0, 0, 40, 0, 72), (0, 0, 216, 256, 184, 256), (5, 5, 98, 142, 30, 162), (15, 15, 0, 36, 190, 256),
               (15, 15, 220, 256, 0, 70)]
    for n0, n1, y0, y1, x0, x1 in windows:
        ya, yb, xa, xb = max(y0 - 1, 0), min(y1 + 1, 256), max(x0 - 1, 0), min(x1 + 1, 256)
        xin = x[n0:n1 + 1, :, ya:yb, xa:xb].double().cpu()
        # zero padding only where the window touches the image border; elsewhere the halo is real data
        pad = (1 if xa == 0 and x0 == 0 else 0, 1 if xb == 256 and x1 == 256 else 0,
               1 if ya == 0 and y0 == 0 else 0, 1 if yb == 256 and y1 == 256 else 0)
        ref = F.conv2d(F.pad(xin, pad), wd, bd)
        assert ref.shape[-2:] == (y1 - y0, x1 - x0)
        _elementwise(y_relu[n0:n1 + 1, :, y0:y1, x0:x1], F.relu(ref), 2e-5, f"relu window {(n0, y0, x0)}")
        _elementwise(y_res[n0:n1 + 1, :, y0:y1, x0:x1], ref + r[n0:n1 + 1, :, y0:y1, x0:x1].double().cpu(), 2e-5,
                     f"residual window {(n0, y0, x0)}")
    # whole tensor: the independent direct-MFMA kernel agrees everywhere
    d = conv3x3(x, pack_conv3x3_weight(w), 64, bias=b, relu=True)
    assert float((d - y_relu).abs().max()) < 2e-5 * (float(d.abs().max()) + float(d.abs().mean()))


def test_config2_winograd4_fullsize_windows_vs_float64(gpu_device):
    """The kernel the headline's body runs since round 3 (Winograd F(4x4,3x3), kmsr_conv3x3_wino4_fwd_f32): body layer 64 -> 64 at
    [16,64,256,256], ReLU and residual epilogues, against a float64 convolution on windows of the full-size result (the four image
    corners, an interior window that straddles workgroup-tile seams -- tiles are 8 rows x 64 columns -- and the last image).  Gate of
    the round-2 verdict: max error <= 5e-5 of the output scale; held here: 2e-5 norm-wise, 1e-4 element-wise against |ref| + mean|ref|
    (the F(2x2) kernel above: 2e-5 element-wise).  Whole tensor: agrees with the F(2x2,3x3) kernel and is bit-reproducible."""
    from kmsr_amd.sr import pack_conv3x3_weight_wino4
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(16, 64, 256, 256, device="cuda", generator=g)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) / 24.0
    b = torch.randn(64, device="cuda", generator=g)
    r = torch.randn(16, 64, 256, 256, device="cuda", generator=g)
    pw = pack_conv3x3_weight_wino4(w)
    y_relu = conv3x3(x, pw, 64, bias=b, relu=True, algo="winograd4")
    y_res = conv3x3(x, pw, 64, bias=b, residual=r, algo="winograd4")
    wd, bd = w.double().cpu(), b.double().cpu()
    windows = [(0, 0, 0, 40, 0, 72), (0, 0, 216, 256, 184, 256), (5, 5, 98, 142, 30, 162), (15, 15, 0, 36, 190, 256),
               (15, 15, 220, 256, 0, 70)]
    worst = 0.0
    for n0, n1, y0, y1, x0, x1 in windows:
        ya, yb, xa, xb = max(y0 - 1, 0), min(y1 + 1, 256), max(x0 - 1, 0), min(x1 + 1, 256)
        xin = x[n0:n1 + 1, :, ya:yb, xa:xb].double().cpu()
        pad = (1 if xa == 0 and x0 == 0 else 0, 1 if xb == 256 and x1 == 256 else 0,
               1 if ya == 0 and y0 == 0 else 0, 1 if yb == 256 and y1 == 256 else 0)
        ref = F.conv2d(F.pad(xin, pad), wd, bd)
        assert ref.shape[-2:] == (y1 - y0, x1 - x0)
        rr = ref + r[n0:n1 + 1, :, y0:y1, x0:x1].double().cpu()
        _elementwise(y_relu[n0:n1 + 1, :, y0:y1, x0:x1], F.relu(ref), 1e-4, f"relu window {(n0, y0, x0)}")
        _elementwise(y_res[n0:n1 + 1, :, y0:y1, x0:x1], rr, 1e-4, f"residual window {(n0, y0, x0)}")
        worst = max(worst, _rel(y_relu[n0:n1 + 1, :, y0:y1, x0:x1], F.relu(ref)), _rel(y_res[n0:n1 + 1, :, y0:y1, x0:x1], rr))
    assert worst < 2e-5, f"max error {worst:.2e} of the output scale"
    y2 = conv3x3(x, pack_conv3x3_weight_wino_vreg(w), 64, bias=b, relu=True, algo="winograd_vreg")
    assert float((y2 - y_relu).abs().max()) < 2e-5 * float(y2.abs().max())
    assert torch.equal(y_res, conv3x3(x, pw, 64, bias=b, residual=r, algo="winograd4"))


def test_config2_fused_upsampler_fullsize(gpu_device):
    """Collapsed upsampler (conv5x5 + PixelShuffle(4) + exact frame strips) vs the layer-by-layer path at config 2's
    full size [16,3,256,256] -> [16,3,1024,1024]: <= 2e-5 of the output scale everywhere, border ring reported separately;
    plus the float64 reference of the three layers on corner / edge / interior windows of the full-size feature map."""
    torch.manual_seed(0)
    net = SRNet(3, 64, 16, 4).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(3)
    lr = torch.rand(16, 3, 256, 256, device="cuda", generator=g) * 80 + 20
    net.fuse_upsampler = False
    layerwise = net(lr)
    net.fuse_upsampler = True
    fused = net(lr)
    assert tuple(fused.shape) == (16, 3, 1024, 1024)
    scale = float(layerwise.abs().max())
    diff = (fused - layerwise).abs()
    ring = torch.ones(1024, 1024, dtype=torch.bool, device="cuda")
    ring[8:-8, 8:-8] = False
    assert float(diff[..., ring].max()) < 2e-5 * scale, f"border ring: {float(diff[..., ring].max()) / scale:.3e} of scale"
    assert float(diff[..., ~ring].max()) < 2e-5 * scale, f"interior: {float(diff[..., ~ring].max()) / scale:.3e} of scale"
    # float64 reference of up1 -> PS -> up2 -> PS -> tail on windows of the trunk output h (re-derived with the layer-wise kernels)
    f = net._conv("head", net.head, lr)
    h = f
    for i, blk in enumerate(net.body):
        t = net._conv(f"b{i}.1", blk.conv1, h, relu=True)
        h = net._conv(f"b{i}.2", blk.conv2, t, residual=h)
    h = net._conv("body_tail", net.body_tail, h, residual=f)
    cv = lambda m: (m.weight.detach().double().cpu(), m.bias.detach().double().cpu())
    for n, y0, y1, x0, x1 in [(0, 0, 12, 0, 12), (7, 244, 256, 100, 116), (15, 120, 134, 244, 256), (3, 60, 72, 60, 72)]:
        m = 2                                   # LR halo: up1 1 px + up2 1/2 px + tail 1/4 px
        ya, yb, xa, xb = max(y0 - m, 0), min(y1 + m, 256), max(x0 - m, 0), min(x1 + m, 256)
        z = h[n:n + 1, :, ya:yb, xa:xb].double().cpu()
        # emulate "the rest of the image exists": pad with real data where available (already cropped in), zero pad at true borders
        w1, b1 = cv(net.up[0]); w2, b2 = cv(net.up[1]); wt, bt = cv(net.tail)
        z = F.pixel_shuffle(F.conv2d(z, w1, b1, padding=1), 2)
        z = F.pixel_shuffle(F.conv2d(z, w2, b2, padding=1), 2)
        z = F.conv2d(z, wt, bt, padding=1)
        # valid region of the crop: rows/cols whose receptive field did not see the crop's artificial zero padding
        oy0, ox0 = 4 * (y0 - ya), 4 * (x0 - xa)
        ref = z[:, :, oy0:oy0 + 4 * (y1 - y0), ox0:ox0 + 4 * (x1 - x0)]
        got = fused[n:n + 1, :, 4 * y0:4 * y1, 4 * x0:4 * x1]
        assert float((got.double().cpu() - ref).abs().max()) < 2e-5 * scale, f"fused vs float64 window {(n, y0, x0)}"


# ------------------------------------------------------------------------------------------------ config 3
def _config3_batch(device, batch=32, seed=3000):
    """bench.py's TrainWorkload data: HR [B,5,1024,1024] U(20,100) -> fused V-B degradation with a random kernel of the
    shipped 10-kernel bank per sample + Gaussian sigma noise -> LR [B,5,256,256]."""
    g = torch.Generator(device=device).manual_seed(seed)
    hr = torch.rand(batch, 5, 1024, 1024, device=device, generator=g) * 80.0 + 20.0
    k = golden("kernels.npz")
    bank, sigma = torch.from_numpy(k["moe_kernels"]).to(device), torch.from_numpy(k["moe_sigmas"]).to(device)
    kid = torch.randint(0, 10, (batch,), generator=torch.Generator().manual_seed(0), dtype=torch.int32).to(device)
    lr = kmsr_amd.degrade_batch(hr, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=kid, sigma=sigma,
                                noise="gaussian", seed=42, offset=0)
    return hr, lr


def _grads(net, lr, hr):
    for p in net.parameters():
        p.grad = None
    loss = F.l1_loss(net(lr), hr)
    loss.backward()
    return float(loss.detach()), [p.grad.detach().clone() for p in net.parameters()]


def test_config3_training_step_fullsize(gpu_device):
    """One SRNet(5,64,16,4).train() step at BASELINE config 3's per-GPU batch: LR [32,5,256,256] produced on the fly from
    HR [32,5,1024,1024].
      (1) finite loss and gradients; two backward passes from the same state are bit-identical (fixed-order reductions);
      (2) a 2-sample sub-batch of the same data at full spatial size agrees with the stock-torch oracle (CPU autograd):
          loss 1e-5, every parameter gradient 5e-4 of its scale;
      (3) the batch-32 gradient equals the mean of the sixteen 2-sample gradients (loss is a mean; checks the batch
          dimension of dgrad / wgrad / bias reductions at full size);
      (4) captured (input, output-gradient) pairs of three layers at full size: wgrad vs float64 on a sub-batch,
          Winograd data gradient vs the direct-kernel data gradient on the whole batch."""
    torch.manual_seed(0)
    ref = SRNetTorch(5, 64, 16, 4)
    net = SRNet(5, 64, 16, 4)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    hr, lr = _config3_batch(gpu_device)
    assert tuple(lr.shape) == (32, 5, 256, 256)

    # capture what three layers see in the backward pass
    captured = {}
    orig_wgrad = kmsr_amd.sr.conv3x3_wgrad

    def spy(x, dy):
        key = (x.shape[1], dy.shape[1], x.shape[2])
        if key in ((64, 64, 256), (64, 256, 256), (64, 5, 1024)) and key not in captured:
            captured[key] = (x.detach(), dy.detach().clone())
        return orig_wgrad(x, dy)
    kmsr_amd.sr.conv3x3_wgrad = spy
    try:
        loss, g_full = _grads(net, lr, hr)
    finally:
        kmsr_amd.sr.conv3x3_wgrad = orig_wgrad
    assert np.isfinite(loss) and all(bool(torch.isfinite(t).all()) for t in g_full)
    loss2, g_again = _grads(net, lr, hr)
    assert loss2 == loss and all(torch.equal(a, b) for a, b in zip(g_full, g_again)), "training step is not deterministic"

    # (2) sub-batch vs the CPU oracle
    sub = slice(4, 6)
    loss_s, g_sub = _grads(net, lr[sub], hr[sub])
    out_r = ref(lr[sub].cpu())
    loss_r = F.l1_loss(out_r, hr[sub].cpu())
    loss_r.backward()
    assert abs(loss_s - float(loss_r.detach())) < 1e-5 * abs(float(loss_r.detach()))
    for (name, pr), gs in zip(ref.named_parameters(), g_sub):
        assert _rel(gs, pr.grad) < 5e-4, f"grad {name}: {_rel(gs, pr.grad):.3e}"

    # (3) batch gradient == mean of sub-batch gradients
    acc = [torch.zeros_like(t, dtype=torch.float64) for t in g_full]
    for s in range(0, 32, 2):
        _, gs = _grads(net, lr[s:s + 2], hr[s:s + 2])
        for a, t in zip(acc, gs):
            a += t.double() / 16.0
    for (name, _), a, t in zip(net.named_parameters(), acc, g_full):
        assert _rel(t, a) < 2e-4, f"batch-32 grad {name} != mean of sub-batch grads: {_rel(t, a):.3e}"

    # (4) layer-level checks on the captured full-size operands
    assert len(captured) == 3
    for (cin, cout, hh), (x, dy) in captured.items():
        nsub = 2 if hh == 256 else 1
        xs, dys = x[:nsub], dy[:nsub]
        dw = conv3x3_wgrad(xs, dys)
        want = torch.nn.grad.conv2d_weight(xs.double().cpu(), (cout, cin, 3, 3), dys.double().cpu(), padding=1)
        assert _rel(dw, want) < 2e-5, f"wgrad {cin}->{cout}@{hh}: {_rel(dw, want):.3e}"
        # additivity over the batch at full size (the persistent wgrad kernel walks all 32 images)
        parts = sum(conv3x3_wgrad(x[i:i + 8], dy[i:i + 8]).double() for i in range(0, 32, 8))
        assert _rel(conv3x3_wgrad(x, dy), parts) < 2e-5
        if cout >= 32:
            wt = torch.randn(cin, cout, 3, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(cin + cout)) / 24.0
            Conv3x3Function.use_winograd = True
            dx_w = Conv3x3Function._run(dy, wt, None, False)
            Conv3x3Function.use_winograd = False
            try:
                dx_d = Conv3x3Function._run(dy, wt, None, False)
            finally:
                Conv3x3Function.use_winograd = True
            assert float((dx_w - dx_d).abs().max()) < 2e-5 * (float(dx_d.abs().max()) + float(dx_d.abs().mean()))


# ------------------------------------------------------------------------------------------------ config 4
def test_config4_bf16_fused_upsampler_fullsize(gpu_device):
    """Config 4's tile through SRNetBF16 with the collapsed upsampler (one bf16 conv5x5 + PixelShuffle(4) launch + border strips) and
    layer by layer: (1) the outer 4-pixel HR frame is bit-identical (it IS the layer-by-layer computation, on four LR strips);
    (2) the collapsed launch alone, on the bf16 trunk features of the full tile: windows (four corners' interiors, an edge, the centre,
    tile seams) within one bf16 ulp of the fp32 conv5x5 of the bf16-rounded operands + PixelShuffle(4), > 97 % bit-identical;
    (3) fused vs layer-wise over the whole 4096^2 output: <= 3e-2 of scale (the layer-wise path rounds two intermediate maps to bf16)."""
    from kmsr_amd.sr import compose_upsampler
    from kmsr_amd.sr_bf16 import SRNetBF16, conv5x5_ps4_bf16
    torch.manual_seed(0)
    net = SRNet(7, 64, 16, 4).cuda().eval()
    ex = SRNetBF16(net)
    g = torch.Generator(device="cuda").manual_seed(4000)
    lr = torch.rand(1, 7, 1024, 1024, device="cuda", generator=g) * 80 + 20
    ex.fuse_upsampler = False
    lay = ex(lr)
    ex.fuse_upsampler = True
    fus = ex(lr)
    assert tuple(fus.shape) == (1, 7, 4096, 4096) and fus.is_contiguous()
    ring = torch.ones(4096, 4096, dtype=torch.bool, device="cuda")
    ring[4:-4, 4:-4] = False
    assert torch.equal(fus[..., ring], lay[..., ring]), "frame differs from the layer-by-layer result"
    scale = float(lay.float().abs().max())
    assert float((fus.float() - lay.float()).abs().max()) < 3e-2 * scale
    # (2) the collapsed launch on known features
    bf = lambda t: t.to(torch.bfloat16).float()
    h = torch.randn(1, 1024, 1024, 64, device="cuda", generator=g).to(torch.bfloat16)
    w5, b5 = compose_upsampler(net.up[0], net.up[1], net.tail)
    out = conv5x5_ps4_bf16(h, ex.up5[0], ex.up5[2], bias=ex.up5[1])
    hc = h.float().permute(0, 3, 1, 2)
    w5r, b5c = bf(w5).cpu(), b5.cpu()
    same = []
    for (wy, wx) in [(0, 0), (992, 992), (0, 500), (500, 37), (250, 992), (120, 120)]:
        ya, yb, xa, xb = max(wy - 2, 0), min(wy + 34, 1024), max(wx - 2, 0), min(wx + 34, 1024)
        pad = (2 - (wx - xa), 2 - (xb - wx - 32), 2 - (wy - ya), 2 - (yb - wy - 32))     # zero padding only at the image border
        ref = bf(F.pixel_shuffle(F.conv2d(F.pad(hc[:, :, ya:yb, xa:xb].cpu(), pad), w5r, b5c), 4))
        got = out[:, :, 4 * wy:4 * wy + 128, 4 * wx:4 * wx + 128].float().cpu()
        err = (got - ref).abs()
        assert bool((err <= ref.abs().clamp_min(1e-3) * 2.0 ** -7).all()), f"window {(wy, wx)}: max err {float(err.max()):.4g}"
        same.append(float((err == 0).float().mean()))
    assert min(same) > 0.97, same


def test_config4_bf16_fullsize_tile(gpu_device):
    """SRNetBF16 on BASELINE config 4's tile [1,7,1024,1024] -> [1,7,4096,4096]:
      (1) tile independence, bit-exact: a crop of the LR tile (plus the network's receptive-field halo) gives exactly the
          full run's pixels away from the crop border -- no cross-tile state in the persistent workgroups;
      (2) the same crop through the fp32 stock-torch oracle (CPU) agrees at the bf16-storage level (3e-2 of scale, PSNR > 40 dB);
      (3) a body layer (64->64 + residual) at [1,1024,1024,64]: windows vs the fp32 conv of the bf16-rounded operands: within one
          bf16 ulp per rounding (conv result, then conv + skip), > 97 % bit-identical."""
    from kmsr_amd.sr_bf16 import SRNetBF16, conv3x3_bf16, from_nhwc, pack_conv3x3_weight_bf16, to_nhwc_bf16
    torch.manual_seed(0)
    ref = SRNetTorch(7, 64, 16, 4).eval()
    net = SRNet(7, 64, 16, 4)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    ex = SRNetBF16(net)
    g = torch.Generator(device="cuda").manual_seed(4000)
    lr = torch.rand(1, 7, 1024, 1024, device="cuda", generator=g) * 80 + 20
    full = ex(lr)
    assert tuple(full.shape) == (1, 7, 4096, 4096) and bool(torch.isfinite(full.float()).all())
    HALO = 40            # receptive field: head 1 + 32 body convs + body_tail 1 + up1 1 + up2 1/2 + tail 1/4  < 37 LR px
    for (y0, x0, sz) in [(448, 448, 128), (0, 0, 160), (864, 700, 160)]:
        crop = ex(lr[:, :, y0:y0 + sz, x0:x0 + sz].contiguous())
        a0y, a0x = (0 if y0 == 0 else HALO), (0 if x0 == 0 else HALO)
        a1y = sz if y0 + sz == 1024 else sz - HALO
        a1x = sz if x0 + sz == 1024 else sz - HALO
        got = crop[:, :, 4 * a0y:4 * a1y, 4 * a0x:4 * a1x]
        want = full[:, :, 4 * (y0 + a0y):4 * (y0 + a1y), 4 * (x0 + a0x):4 * (x0 + a1x)]
        assert torch.equal(got, want), f"tile at {(y0, x0)} depends on its neighbourhood beyond the receptive field"
    y0 = x0 = 448
    with torch.no_grad():
        want = ref(lr[:, :, y0:y0 + 128, x0:x0 + 128].cpu())[:, :, 4 * HALO:-4 * HALO, 4 * HALO:-4 * HALO]
    got = full[:, :, 4 * (y0 + HALO):4 * (y0 + 128 - HALO), 4 * (x0 + HALO):4 * (x0 + 128 - HALO)].float().cpu()
    rel = float((got - want).abs().max() / want.abs().max())
    assert rel < 3e-2 and kmsr_amd.psnr(got, want, float(want.abs().max())) > 40.0, f"bf16 net vs fp32 oracle: {rel:.3e}"

    # (3) body layer at the tile size
    bf = lambda t: t.to(torch.bfloat16).float()
    xs = torch.randn(1, 64, 1024, 1024, device="cuda", generator=g)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) / 24.0
    b = torch.zeros(64, device="cuda")
    b[:] = torch.randn(64, device="cuda", generator=g)
    xh = to_nhwc_bf16(xs, 64)
    out = conv3x3_bf16(xh, pack_conv3x3_weight_bf16(w, 64, 64), 64, bias=b, residual=xh)
    outc = from_nhwc(out, 64).float()
    for (wy, wx) in [(0, 0), (960, 960), (500, 37), (0, 960)]:
        ya, yb, xa, xb = max(wy - 1, 0), min(wy + 65, 1024), max(wx - 1, 0), min(wx + 65, 1024)
        xin = bf(xs[:, :, ya:yb, xa:xb]).cpu()
        pad = (1 if wx == 0 else 0, 1 if wx + 64 == 1024 else 0, 1 if wy == 0 else 0, 1 if wy + 64 == 1024 else 0)
        conv = F.conv2d(F.pad(xin, pad), bf(w).cpu(), b.cpu())
        refw = bf(bf(conv) + bf(xs[:, :, wy:wy + 64, wx:wx + 64]).cpu())
        err = (outc[:, :, wy:wy + 64, wx:wx + 64].cpu() - refw).abs()
        # two roundings: the conv result is rounded to bf16 before the skip is added (a last-bit difference there survives the
        # add unchanged, even when conv + skip cancels to a much smaller number) and the sum is rounded again (which can straddle
        # a boundary once more): bound = one bf16 ulp of the conv result + one of the final value
        ulp = (refw.abs() + bf(conv).abs()).clamp_min(1e-3) * 2.0 ** -7
        assert bool((err <= ulp).all()), f"bf16 body layer window {(wy, wx)}: max err {float(err.max()):.4g}"
        assert float((err == 0).float().mean()) > 0.97


# ------------------------------------------------------------------------------------------------ config 5
def test_config5_bulk_kernel_estimation_512_patches(gpu_device):
    """BASELINE config 5: 512 seeded [5,256,256] GOCI-shaped patches through bulk_extract_kernels in the reference's batches of
    8 (muti_kernel/train.py:118-188) with the reference-trained-layout generator weights of the golden fixture:
    8 of them against oracle.muti_extract_effective_kernels (restatement of networks.py:135-202, pinned by the reference's
    own outputs in kernelgan.npz); world-size-2 shards == the full run (bit-exact); kernel invariants for all 512."""
    KG = golden("kernelgan.npz")
    gen = kmsr_amd.DynamicMultiBandLinearGenerator(in_ch=5, mid_ch=32, ks=(7, 5, 3, 1, 1, 1), scale_gain=0.1)
    sd_t = {k[len("muti_sd__generator."):]: torch.from_numpy(np.asarray(KG[k])) for k in KG.files if k.startswith("muti_sd__generator.")}
    gen.load_state_dict(sd_t, strict=True)
    gen = gen.cuda()
    rs = np.random.RandomState(512)
    # GOCI radiance-like patches: a smooth field + texture, different per patch so the condition scales (and kernels) differ
    patches = []
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32) / 256.0
    for i in range(512):
        amp, ph = rs.rand(5, 1, 1).astype(np.float32), rs.rand(5, 1, 1).astype(np.float32) * 6.28
        patches.append((0.5 + 0.4 * amp * np.sin(6.28 * (1 + i % 7) * yy[None] + ph) * np.cos(6.28 * (1 + i % 5) * xx[None])
                        + 0.05 * rs.randn(5, 256, 256).astype(np.float32)).astype(np.float32))
    full = dict(kmsr_amd.bulk_extract_kernels(patches, gen, patch_size=256, batch_size=8))
    assert sorted(full) == list(range(512))
    allk = torch.stack([full[i] for i in range(512)])
    assert tuple(allk.shape) == (512, 5, 13, 13)
    assert float(allk.min()) >= 0.0 and np.allclose(allk.sum(dim=(2, 3)).cpu().numpy(), 1.0, atol=1e-5)
    assert float((allk[0] - allk[1]).abs().max()) > 0.0                       # dynamic kernels: patches differ
    parts = {}
    for r in range(2):
        mine = dict(kmsr_amd.bulk_extract_kernels(patches, gen, patch_size=256, batch_size=8, rank=r, world_size=2))
        assert sorted(mine) == list(range(r, 512, 2))
        parts.update(mine)
    for i in range(512):
        assert torch.equal(parts[i], full[i]), f"patch {i}: shard result differs from the full run"
    sd = {k[len("muti_sd__"):]: np.asarray(KG[k]) for k in KG.files if k.startswith("muti_sd__")}
    for i in (0, 1, 77, 200, 255, 256, 400, 511):
        want = oracle.muti_extract_effective_kernels(sd, patches[i][None], reduce_batch=False)[0]
        assert_close(full[i].cpu().numpy(), want, rtol=3e-4, atol=3e-7, what=f"patch {i} vs oracle")
