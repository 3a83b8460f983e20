#!/usr/bin/env python3
"""Run the SR body conv (64->64 @ [16,64,256,256]) a few times -- a profiling target."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_conv3x3_weight
torch.cuda.set_device(0)
N, Cin, H, W, Cout = 16, 64, 256, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
pw = pack_conv3x3_weight(w)
res = torch.randn(N, Cout, H, W, device="cuda")
for _ in range(6):
    out = conv3x3(x, pw, Cout, bias=b, residual=res)
torch.cuda.synchronize()
