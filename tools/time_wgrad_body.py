#!/usr/bin/env python3
"""wgrad of the body layer only (64->64 @256^2 x16) -- for rocprofv3 / PMC runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3_wgrad
torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(B, 64, 256, 256, device="cuda"); dy = torch.randn(B, 64, 256, 256, device="cuda")
conv3x3_wgrad(x, dy); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): conv3x3_wgrad(x, dy)
e1.record(); torch.cuda.synchronize()
print(f"wgrad body: {e0.elapsed_time(e1)/10*1e3:.1f} us")
