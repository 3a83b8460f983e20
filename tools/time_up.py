#!/usr/bin/env python3
"""Upsampler convs (64->256 + PixelShuffle at 256^2 and 512^2, batch 16) and the whole SRNet forward, per Winograd kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import SRNet, conv3x3, pack_for
torch.cuda.set_device(0)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, N, Cin, H, W, Cout in [("up1 64->256 ps @256", 16, 64, 256, 256, 256), ("up2 64->256 ps @512", 16, 64, 512, 512, 256)]:
    xx = torch.randn(N, Cin, H, W, device="cuda"); ww = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    for algo in ("winograd", "winograd_vreg"):
        pk = pack_for(algo, ww)
        print(f"  {name:22s} {algo:14s} {t(lambda: conv3x3(xx, pk, Cout, bias=b, pixel_shuffle=True, algo=algo)):8.3f} ms")
torch.manual_seed(0)
net = SRNet(3, 64, 16, 4).cuda().eval()
x = torch.rand(16, 3, 256, 256, device="cuda")
for fuse in (False, True):
    net.fuse_upsampler = fuse
    print(f"SRNet fwd batch 16 @256^2 (KMSR_WINO_ALGO={os.environ.get('KMSR_WINO_ALGO', 'default')}, fused upsampler {fuse}): {t(lambda: net(x)):8.3f} ms")
