#!/usr/bin/env python3
"""Body conv 64->64 @256^2 x16 in its three epilogue forms (plain / ReLU / residual), per algorithm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_conv3x3_weight, pack_conv3x3_weight_wino, pack_conv3x3_weight_wino_vreg
torch.cuda.set_device(0)
N, Cin, H, W, Cout = 16, 64, 256, 256, 64
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
r = torch.randn(N, Cout, H, W, device="cuda"); out = torch.empty_like(r)
for algo, pack in (("winograd", pack_conv3x3_weight_wino), ("winograd_vreg", pack_conv3x3_weight_wino_vreg)):
    pw = pack(w)
    for name, kw in (("relu", dict(relu=True)), ("residual", dict(residual=r))):
        conv3x3(x, pw, Cout, bias=b, out=out, algo=algo, **kw); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): conv3x3(x, pw, Cout, bias=b, out=out, algo=algo, **kw)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
        print(f"{algo:13s} body + {name:8s}: {sorted(ts)[2]*1e3:7.1f} us")
