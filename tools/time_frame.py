#!/usr/bin/env python3
"""kmsr_upsampler_frame_f32 on config 2's shape (16 x 64 x 256 x 256 -> 3 bands): launch time by HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
from kmsr_amd.sr import SRNet
torch.manual_seed(0)
net = SRNet(n_colors=3, n_feats=64, n_resblocks=1, scale=4).cuda().eval()
h = torch.randn(16, 64, 256, 256, device="cuda")
fu = net._fused_up
y = fu(net, h)
import kmsr_amd._lib as L
lib = L.load()
orig = lib.kmsr_upsampler_frame_f32
ev = []
def timed(*a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(*a); e1.record(); ev.append((e0, e1)); return r
lib.kmsr_upsampler_frame_f32 = timed
for _ in range(12):
    fu(net, h)
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
print(f"upsampler frame launch: median {ms[len(ms)//2]*1e3:.1f} us (min {ms[0]*1e3:.1f})")
