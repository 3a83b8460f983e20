#!/usr/bin/env python3
"""Stride-2 direct MFMA conv at the condition encoder's shapes (config 5, 512 patches): 32 -> 64 @256^2 and 64 -> 64 @128^2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_for
torch.cuda.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
DATA = sys.argv[2] if len(sys.argv) > 2 else "randn"       # "zeros": same instruction stream on zero operands (power / clock probe)
for name, (Cin, H, Cout) in (("32->64 @256^2", (32, 256, 64)), ("64->64 @128^2", (64, 128, 64))):
    x = torch.randn(N, Cin, H, H, device="cuda") if DATA == "randn" else torch.zeros(N, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    pw = pack_for("direct", w)
    out = conv3x3(x, pw, Cout, bias=b, relu=True, stride=2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): conv3x3(x, pw, Cout, bias=b, relu=True, stride=2, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * 9 * N * Cin * Cout * (H // 2) ** 2
    print(f"{name} x{N} ({DATA}): {ms:7.3f} ms  {fl/ms/1e9:6.1f} TFLOP/s = {fl/ms/1e9/157.3:5.3f} of the fp32 MFMA peak")
    if N <= 8:
        ref = torch.relu(torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2, padding=1))
        print("   max err vs float64:", float((out.double().cpu() - ref).abs().max()), "scale", float(ref.abs().max()))
