#!/usr/bin/env python3
"""One algorithm's body conv 64->64 @256^2 x16 (residual epilogue), a few launches: target command for rocprofv3 passes.
usage: one_vreg.py [algo] [n_launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_for
algo = sys.argv[1] if len(sys.argv) > 1 else "winograd_vreg"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "res"
torch.cuda.set_device(0)
x = torch.randn(16, 64, 256, 256, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
r = torch.randn(16, 64, 256, 256, device="cuda"); out = torch.empty_like(r)
pw = pack_for(algo, w)
for _ in range(n):
    conv3x3(x, pw, 64, bias=b, out=out, algo=algo, **(dict(residual=r) if mode == "res" else dict(relu=True)))
torch.cuda.synchronize()
