"""GPU tests of the bf16 channels-last conv path (BASELINE config 4).  Reference: torch fp32 conv on the bf16-ROUNDED
inputs and weights (products of bf16 values are exact in fp32, so only the fp32 accumulation order and the final bf16
rounding differ): every output must be within one bf16 ulp of the reference, and almost all must be bit-identical."""
import pytest
import torch
import torch.nn.functional as F

import kmsr_amd
from kmsr_amd.sr_bf16 import SRNetBF16, conv3x3_bf16, from_nhwc, pack_conv3x3_weight_bf16, to_nhwc_bf16
from oracle.sr_torch import SRNetTorch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).float()


CASES = [
    # N, Cin, H,  W,  Cout, bias, relu, res,  ps
    (2, 64, 20, 40, 64, True, True, False, False),
    (1, 64, 8, 32, 64, True, False, True, False),
    (1, 7, 33, 45, 64, True, False, False, False),      # head: 7 bands padded to 16
    (1, 64, 16, 32, 7, True, False, False, False),      # tail: 7 bands in a 16-channel output group
    (1, 64, 12, 40, 256, True, False, False, True),     # upsampler + fused PixelShuffle
    (2, 32, 9, 31, 64, False, True, True, False),
]


@pytest.mark.parametrize("N,Cin,H,W,Cout,bias,relu,res,ps", CASES)
def test_conv3x3_bf16_vs_torch(N, Cin, H, W, Cout, bias, relu, res, ps, gpu_device):
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g) if bias else None
    cin_p, cout_p = (Cin + 15) // 16 * 16, (Cout + 15) // 16 * 16
    oshape = (N, Cout // 4, 2 * H, 2 * W) if ps else (N, Cout, H, W)
    r = torch.randn(*oshape, generator=g) if res else None
    ref = F.conv2d(_bf(x), _bf(w), b, padding=1)
    if relu:
        ref = F.relu(ref)
    if ps:
        ref = F.pixel_shuffle(ref, 2)
    if r is not None:
        ref = _bf(ref) + _bf(r)                     # the kernel rounds the conv result to bf16 before the residual add
    ref = _bf(ref)
    bp = None
    if b is not None:
        bp = torch.zeros(max(cout_p, 64))
        bp[:Cout] = b
        bp = bp.cuda()
    xh = to_nhwc_bf16(x.cuda(), cin_p)
    rh = None if r is None else to_nhwc_bf16(r.cuda(), cout_p)
    out = conv3x3_bf16(xh, pack_conv3x3_weight_bf16(w.cuda(), cin_p, cout_p), cout_p, bias=bp, relu=relu, residual=rh,
                       pixel_shuffle=ps)
    got = from_nhwc(out, oshape[1]).float().cpu()
    assert tuple(got.shape) == oshape
    err = (got - ref).abs()
    ulp = ref.abs().clamp_min(1e-3) * 2.0 ** -7      # one bf16 ulp is at most 2^-7 relative
    assert bool((err <= ulp).all()), f"max err {float(err.max()):.4g}"
    assert float((err == 0).float().mean()) > 0.97   # rounding ties aside, results are bit-identical
    if not ps and cout_p > Cout:                     # padded output channels are exactly the (zero-padded) bias: zero
        assert float(out[..., Cout:].float().abs().max()) == 0.0


def test_srnet_bf16_config4_shape(gpu_device):
    """7-band x4 SR through the bf16 executor vs the fp32 oracle (bf16 storage of 35 layers: ~1e-2 relative)."""
    torch.manual_seed(0)
    ref = SRNetTorch(7, 64, 4, 4).eval()
    net = kmsr_amd.SRNet(7, 64, 4, 4)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    x = torch.rand(1, 7, 40, 72, generator=torch.Generator().manual_seed(1)) * 80 + 20
    with torch.no_grad():
        want = ref(x)
    got = SRNetBF16(net)(x.cuda()).float().cpu()
    assert tuple(got.shape) == (1, 7, 160, 288)
    rel = float((got - want).abs().max() / want.abs().max())
    assert rel < 3e-2, f"rel err {rel:.3e}"
    assert kmsr_amd.psnr(got, want, float(want.abs().max())) > 40.0


# ------------------------------------------------------------------ backward legs (kmsr_conv3x3_{dgrad,wgrad}_bf16)
BWD_CASES = [  # N, Cin, H, W, Cout, relu
    (2, 64, 20, 40, 64, True),        # body layer (Winograd-domain weight gradient on the expanded planes)
    (1, 7, 33, 45, 64, False),        # head: 7 bands padded to 16, odd sizes (direct weight gradient)
    (1, 64, 16, 32, 7, False),        # tail: 7 bands in a 16-channel group (Cout <= 32 kernel)
    (2, 32, 10, 28, 48, True),
]


@pytest.mark.parametrize("N,Cin,H,W,Cout,relu", BWD_CASES)
def test_conv3x3_bf16_backward_vs_torch(N, Cin, H, W, Cout, relu, gpu_device):
    """Reference: float64 autograd of the conv on the bf16-ROUNDED x, w and (for the gradients) the bf16-rounded incoming
    gradient.  dx is a bf16 tensor -> within one bf16 ulp of the reference's value (plus the fp32 accumulation error);
    dw / db are fp32 sums of exact products -> 2e-5 of their scale."""
    from kmsr_amd.sr_bf16 import Conv3x3BF16Function
    g = torch.Generator().manual_seed(Cin * 11 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(N, Cout, H, W, generator=g)
    cin_p, cout_p = (Cin + 15) // 16 * 16, (Cout + 15) // 16 * 16
    xr, wr = _bf(x).double().requires_grad_(True), _bf(w).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, padding=1)
    ref = torch.relu(pre) if relu else pre
    # the kernel's mask is taken from ITS bf16 output; use the same mask in the reference (outputs within one ulp of 0 differ otherwise)
    xg = to_nhwc_bf16(x.cuda(), cin_p).requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = Conv3x3BF16Function.apply(xg, wg, bg, relu)
    gyb = to_nhwc_bf16(gy.cuda(), cout_p)
    y.backward(gyb)
    gy_ref = _bf(gy).double()
    if relu:
        mask = (from_nhwc(y.detach(), Cout).float().cpu() > 0).double()
        (pre * mask).backward(gy_ref)
    else:
        ref.backward(gy_ref)
    # forward sanity (one bf16 ulp)
    yf = from_nhwc(y.detach(), Cout).float().cpu()
    assert float(((yf - ref.detach().float()).abs() / (ref.detach().abs().float() * 2.0 ** -7 + 1e-3)).max()) <= 1.0
    dx = from_nhwc(xg.grad, Cin).float().cpu()
    dxr = xr.grad.float()
    tol = dxr.abs() * 2.0 ** -7 + 2e-5 * float(dxr.abs().max())
    assert bool(((dx - dxr).abs() <= tol).all()), f"dgrad max err {float((dx - dxr).abs().max()):.3e}"
    assert float(xg.grad[..., Cin:].abs().max()) == 0.0 if cin_p > Cin else True           # padding channels stay zero
    for got, want, nm in ((wg.grad.cpu(), wr.grad, "dw"), (bg.grad.cpu(), br.grad, "db")):
        err = float((got.double() - want).abs().max() / want.abs().max())
        assert err < 2e-5, f"{nm} rel err {err:.3e}"


def test_bf16_backward_rejects_unsupported(gpu_device):
    from kmsr_amd.sr_bf16 import conv3x3_dgrad_bf16, pack_conv3x3_weight_bf16_dgrad
    w = torch.randn(256, 64, 3, 3, device="cuda")
    dy = torch.zeros(1, 8, 32, 256, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):          # F -> 4F layer: the gradient conv would read 256 channels (kernel limit 64)
        conv3x3_dgrad_bf16(dy, pack_conv3x3_weight_bf16_dgrad(w, 64, 256), 64)


PS4_CASES = [  # N, F, H, W, bands
    (1, 64, 16, 32, 7),      # config 4's bands: 112 sub-pixel channels = a full 64-channel group + 48
    (2, 64, 13, 45, 3),      # ragged tiles, 48 channels in one group
    (1, 32, 9, 70, 1),       # two 16-channel chunks, one band, three column tiles
    (1, 64, 40, 36, 5),
]


@pytest.mark.parametrize("N,Fc,H,W,bands", PS4_CASES)
def test_conv5x5_ps4_bf16_vs_torch(N, Fc, H, W, bands, gpu_device):
    """kmsr_conv5x5_ps4_fwd_bf16 (the collapsed x4 upsampler of the bf16 path) against the fp32 conv of the bf16-rounded operands +
    PixelShuffle(4), rounded to bf16: within one bf16 ulp everywhere (K = 25 * F products accumulated in fp32 in another order), output
    pre-filled with NaN: every element written."""
    from kmsr_amd.sr_bf16 import conv5x5_ps4_bf16, pack_conv5x5_ps4_weight_bf16
    g = torch.Generator().manual_seed(Fc + 10 * bands + H)
    x = torch.randn(N, Fc, H, W, generator=g)
    w5 = torch.randn(16 * bands, Fc, 5, 5, generator=g) / (5.0 * Fc ** 0.5)
    b5 = torch.randn(16 * bands, generator=g)
    ref = _bf(F.pixel_shuffle(F.conv2d(_bf(x), _bf(w5), b5, padding=2), 4))
    out = torch.full((N, bands, 4 * H, 4 * W), float("nan"), device=gpu_device, dtype=torch.bfloat16)
    conv5x5_ps4_bf16(to_nhwc_bf16(x.cuda(), Fc), pack_conv5x5_ps4_weight_bf16(w5.cuda(), Fc), 16 * bands, bias=b5.cuda(), out=out)
    got = out.float().cpu()
    assert not bool(torch.isnan(got).any()), "unwritten output elements"
    err = (got - ref).abs()
    ulp = ref.abs().clamp_min(1e-3) * 2.0 ** -7
    assert bool((err <= ulp).all()), f"max err {float(err.max()):.4g}"
    assert float((err == 0).float().mean()) > 0.97


@pytest.mark.parametrize("colors,shape", [(7, (1, 7, 40, 72)), (3, (2, 3, 16, 24))])
def test_srnet_bf16_fused_upsampler(colors, shape, gpu_device):
    """SRNetBF16.fuse_upsampler: the outer 4-pixel HR frame (recomputed layer by layer on four LR border strips) is bit-identical to
    the layer-by-layer path; the interior (one conv5x5 + PixelShuffle(4) launch on the composed weights, no bf16 rounding of the two
    intermediate feature maps) agrees with it at the bf16-storage level and is at least as close to the fp32 oracle."""
    torch.manual_seed(3)
    ref = SRNetTorch(colors, 64, 2, 4).eval()
    net = kmsr_amd.SRNet(colors, 64, 2, 4)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    ex = SRNetBF16(net)
    x = torch.rand(*shape) * 2 - 1
    ex.fuse_upsampler = False
    lay = ex(x.cuda()).float().cpu()
    ex.fuse_upsampler = True
    fus = ex(x.cuda())
    assert fus.is_contiguous() and tuple(fus.shape) == (shape[0], colors, 4 * shape[2], 4 * shape[3])
    fus = fus.float().cpu()
    ring = torch.ones_like(lay, dtype=torch.bool)
    ring[:, :, 4:-4, 4:-4] = False
    assert torch.equal(fus[ring], lay[ring]), "the frame must be the layer-by-layer result"
    with torch.no_grad():
        want = ref(x)
    scale = float(want.abs().max())
    e_f, e_l = float((fus - want).abs().max()) / scale, float((lay - want).abs().max()) / scale
    assert e_f < 3e-2 and e_f <= 1.25 * e_l + 4e-3, f"fused {e_f:.3e} vs layer-wise {e_l:.3e} of scale from the fp32 oracle"
    assert float((fus - lay).abs().max()) < 3e-2 * scale


# ------------------------------------------------------------------ fused residual block (kmsr_resblock_fwd_bf16, round 4)
@pytest.mark.parametrize("N,H,W", [(1, 8, 16), (2, 24, 48), (1, 33, 45), (1, 5, 7), (3, 17, 16), (1, 64, 160), (1, 1, 1)])
def test_resblock_bf16_bit_identical_to_two_launches(N, H, W, gpu_device):
    """One launch (intermediate in LDS, conv1 recomputed on each tile's halo ring) must reproduce the two-launch path BIT FOR BIT:
    same MFMA accumulation order, the intermediate rounded to bf16 and zero outside the image, conv2 + bias rounded before the
    skip add.  Shapes: one tile, several tiles, ragged right / bottom tiles (H % 8, W % 16 != 0), images smaller than a tile,
    more tiles than workgroups' first items."""
    from kmsr_amd.sr_bf16 import resblock_bf16
    g = torch.Generator().manual_seed(H * 131 + W)
    x = to_nhwc_bf16((torch.randn(N, 64, H, W, generator=g) * 2).cuda())
    w1 = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    w2 = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b1 = torch.randn(64, generator=g).cuda()
    b2 = torch.randn(64, generator=g).cuda()
    p1, p2 = pack_conv3x3_weight_bf16(w1.cuda(), 64, 64), pack_conv3x3_weight_bf16(w2.cuda(), 64, 64)
    t = conv3x3_bf16(x, p1, 64, bias=b1, relu=True)
    want = conv3x3_bf16(t, p2, 64, bias=b2, residual=x)
    got = resblock_bf16(x, p1, b1, p2, b2)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), \
        f"{int((got.view(torch.int16) != want.view(torch.int16)).sum())} of {got.numel()} elements differ"
    # and against the fp32 reference of the bf16-rounded operands (the two-launch path's own bound: one bf16 ulp per rounding)
    ref_t = _bf(torch.relu(F.conv2d(_bf(from_nhwc(x, 64).float().cpu()), _bf(w1), b1.cpu(), padding=1)))
    ref = F.conv2d(ref_t, _bf(w2), b2.cpu(), padding=1) + from_nhwc(x, 64).float().cpu()
    err = (from_nhwc(got, 64).float().cpu() - ref).abs()
    assert float(err.max()) <= 0.05 * float(ref.abs().max()) and float((err / (ref.abs() + 1)).mean()) < 4e-3
    # no bias
    assert torch.equal(resblock_bf16(x, p1, None, p2, None).view(torch.int16),
                       conv3x3_bf16(conv3x3_bf16(x, p1, 64, relu=True), p2, 64, residual=x).view(torch.int16))


def test_srnet_bf16_fused_resblocks_equal_layerwise(gpu_device):
    """SRNetBF16 with one launch per residual block == the layer-by-layer body, bit for bit (whole network output)."""
    torch.manual_seed(3)
    net = kmsr_amd.SRNet(7, 64, 4, 4).cuda().eval()
    ex = SRNetBF16(net)
    x = torch.rand(1, 7, 40, 72, device="cuda") * 80 + 20
    a = ex(x)
    ex.fuse_resblocks = False
    b = ex(x)
    assert torch.equal(a, b)
