#!/usr/bin/env python3
"""Body layer (3x3 conv 64->64 @256x256, batch 16) on the F(4x4,3x3) kernel next to the F(2x2,3x3) one: relu and residual forms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
from kmsr_amd.sr import conv3x3, pack_for
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(0)
N, C, H, W = 16, 64, 256, 256
x = torch.randn(N, C, H, W, generator=g).cuda()
if os.environ.get("KMSR_ZERO_INPUT"):       # clock probe: same launch, same traffic, all-zero input
    x.zero_()
w = (torch.randn(C, C, 3, 3, generator=g) / 24).cuda()
b = torch.randn(C, generator=g).cuda()
out = torch.empty_like(x)
ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), padding=1)
r = torch.randn(N, C, H, W, generator=g).cuda()
if os.environ.get("KMSR_ZERO_INPUT"):
    r.zero_()
for algo in (sys.argv[1:] or ("winograd_vreg", "winograd4")):
    pw = pack_for(algo, w)
    for name, kw in (("relu", dict(relu=True)), ("residual", dict(residual=r)), ("relu", dict(relu=True))):
        for _ in range(30):
            conv3x3(x, pw, C, bias=b, algo=algo, out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            conv3x3(x, pw, C, bias=b, algo=algo, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        fl = 2 * 9 * N * C * C * H * W
        issued = fl / (4.0 if algo == "winograd4" else 2.25)
        print(f"{algo:14s} {name:9s} {ms*1e3:8.1f} us   algorithmic {fl/ms/1e9:7.1f} TF/s   issued {issued/ms/1e9:6.1f} TF/s = {issued/ms/1e9/157.3:.3f} of the fp32 MFMA peak")
    y = conv3x3(x[:1], pw, C, bias=b, algo=algo)
    print(f"{algo:14s} max err vs float64: {float((y.double()-ref).abs().max()/ref.abs().max()):.2e} of scale")
