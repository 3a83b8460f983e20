#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, conv3x3_wgrad, pack_conv3x3_weight
torch.cuda.set_device(0)
for name, N, Cin, H, W, Cout in [("body 64->64 @256^2 x16", 16, 64, 256, 256, 64), ("up2 64->256 @512^2 x4", 4, 64, 512, 512, 256), ("tail 64->3 @1024^2 x4", 4, 64, 1024, 1024, 3)]:
    x = torch.randn(N, Cin, H, W, device="cuda"); dy = torch.randn(N, Cout, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda")
    fl = 2.0 * 9 * N * Cin * Cout * H * W
    for what, fn in (("wgrad", lambda: conv3x3_wgrad(x, dy)), ("dgrad", lambda: conv3x3(dy, pw, Cin))):
        pw = pack_conv3x3_weight(w, transpose_flip=True)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
        print(f"{name:26s} {what}: {ms:8.3f} ms  {fl/ms/1e9:7.1f} TF ({fl/ms/1e9/157.3*100:5.1f}%)")
