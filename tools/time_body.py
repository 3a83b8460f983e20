#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_conv3x3_weight
torch.cuda.set_device(0)
N, Cin, H, W, Cout = 16, 64, 256, 256, 64
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
algo = os.environ.get("ALGO", "direct")
from kmsr_amd.sr import pack_conv3x3_weight_wino_dma
from kmsr_amd.sr import pack_conv3x3_weight_wino
pw = {"winograd": pack_conv3x3_weight_wino, "winograd_dma": pack_conv3x3_weight_wino_dma, "direct": pack_conv3x3_weight}[algo](w); out = conv3x3(x, pw, Cout, bias=b, algo=algo); torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): conv3x3(x, pw, Cout, bias=b, out=out, algo=algo)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
ms = sorted(ts)[len(ts)//2]; fl = 2.0*9*N*Cin*Cout*H*W
print(f"{os.environ.get('KMSR_LIB_OVERRIDE','default')[-22:]:24s} {algo} body 64->64: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
