#!/usr/bin/env python3
"""Quick correctness check of whatever library KMSR_LIB_OVERRIDE points at: F(4x4,3x3) body layer vs float64 on two images (relu / residual)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kmsr_amd.sr import conv3x3, pack_for
g = torch.Generator().manual_seed(1)
x = torch.randn(20, 64, 256, 256, generator=g).cuda(); w = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda(); b = torch.randn(64, generator=g).cuda()
r = torch.randn(20, 64, 256, 256, generator=g).cuda()
pw = pack_for("winograd4", w)
for kw, name in ((dict(relu=True), "relu"), (dict(residual=r), "residual")):
    y = conv3x3(x, pw, 64, bias=b, algo="winograd4", **kw)
    worst = 0.0
    for n in (0, 7, 19):
        ref = torch.nn.functional.conv2d(x[n:n+1].double(), w.double(), b.double(), padding=1)
        ref = ref.relu() if "relu" in kw else ref + r[n:n+1].double()
        worst = max(worst, float((y[n:n+1].double() - ref).abs().max() / ref.abs().max()))
    print(f"check {name}: max err {worst:.2e} of scale", "OK" if worst < 2e-5 else "FAIL")
