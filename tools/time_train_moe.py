#!/usr/bin/env python3
"""MoE GAN trainer at the reference's own size (train_gemini.py:164-165: 8 patches 5x256x256, real LR 5x64x64)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
torch.cuda.set_device(0)
torch.manual_seed(0)
for B in (8, 64):
    model = kmsr_amd.ContentAdaptiveDegradation(10, 5, 13).cuda()
    D = kmsr_amd.PatchDiscriminator(5).cuda()
    patches = torch.rand(B, 5, 256, 256, device="cuda") * 80 + 20
    real = torch.rand(B, 5, 64, 64, device="cuda") * 80 + 20
    kmsr_amd.train_moe(model, D, patches, real, total_iters=5, log=None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    kmsr_amd.train_moe(model, D, patches, real, total_iters=n, log=None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"train_moe B={B}: {dt*1e3:.2f} ms/iteration ({B/dt:.0f} patches/s)")
from kmsr_amd.degradation import degrade_backward
x = torch.rand(32, 5, 1024, 1024, device="cuda") * 80 + 20
g = torch.randn(32, 5, 256, 256, device="cuda")
for ws in (True, False):
    degrade_backward(x, g, 13, 4, want_sigma=ws); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): degrade_backward(x, g, 13, 4, want_sigma=ws)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
    by = x.numel() * 4 + g.numel() * 4
    print(f"degrade_backward HR [32,5,1024,1024] sigma={ws}: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000:.2f} of HBM peak)")
