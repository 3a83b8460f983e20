#!/usr/bin/env python3
"""Time single conv3x3 layers of the SR net at BASELINE config-2 shapes (TFLOP/s vs the 157.3 TF fp32 MFMA peak)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_conv3x3_weight

torch.cuda.set_device(0)
cases = [("body 64->64 @256^2 x16", 16, 64, 256, 256, 64, False), ("head 3->64 @256^2 x16", 16, 3, 256, 256, 64, False),
         ("up1 64->256 ps @256^2 x16", 16, 64, 256, 256, 256, True), ("up2 64->256 ps @512^2 x16", 16, 64, 512, 512, 256, True),
         ("tail 64->3 @1024^2 x16", 16, 64, 1024, 1024, 3, False)]
for name, N, Cin, H, W, Cout, ps in cases:
    x = torch.randn(N, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    pw = pack_conv3x3_weight(w)
    out = conv3x3(x, pw, Cout, bias=b, pixel_shuffle=ps)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            conv3x3(x, pw, Cout, bias=b, pixel_shuffle=ps, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[len(ts) // 2]
    fl = 2.0 * 9 * N * Cin * Cout * H * W
    print(f"{name:28s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3 * 100:5.1f}% of fp32 MFMA peak)")
