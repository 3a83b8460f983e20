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
