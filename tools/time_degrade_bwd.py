#!/usr/bin/env python3
"""degrade_backward at the config-3 batch (HR [32,5,1024,1024]); run under rocprofv3 --kernel-trace --stats for the split."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.degradation import degrade_backward
torch.cuda.set_device(0)
x = torch.rand(32, 5, 1024, 1024, device="cuda") * 80 + 20
g = torch.randn(32, 5, 256, 256, device="cuda")
for ws in (True, False, True, False):
    degrade_backward(x, g, 13, 4, want_sigma=ws); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): degrade_backward(x, g, 13, 4, want_sigma=ws)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 50
    by = x.numel() * 4 + g.numel() * 4
    print(f"degrade_backward HR [32,5,1024,1024] sigma={ws}: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000:.2f} of HBM peak)")
