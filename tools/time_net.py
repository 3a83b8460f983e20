#!/usr/bin/env python3
"""SRNet forward (config 2 shape) with the Winograd path on / off, and per-layer-type timings."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import SRNet, conv3x3, pack_conv3x3_weight, pack_conv3x3_weight_wino
torch.cuda.set_device(0)
torch.manual_seed(0)
net = SRNet(3, 64, 16, 4).cuda().eval()  # eval() -> fused inference path
x = torch.rand(16, 3, 256, 256, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for w in (False, True):
    net.use_winograd = w
    print(f"SRNet fwd batch 16 @256^2, winograd={w}: {t(lambda: net(x)):8.3f} ms")
for name, N, Cin, H, W, Cout, ps in [("body 64->64 @256", 16, 64, 256, 256, 64, False), ("up1 64->256 ps @256", 16, 64, 256, 256, 256, True), ("up2 64->256 ps @512", 16, 64, 512, 512, 256, True)]:
    xx = torch.randn(N, Cin, H, W, device="cuda"); ww = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    pd, pw = pack_conv3x3_weight(ww), pack_conv3x3_weight_wino(ww)
    fl = 2.0 * 9 * N * Cin * Cout * H * W
    for algo, pk in (("direct", pd), ("winograd", pw)):
        ms = t(lambda: conv3x3(xx, pk, Cout, bias=b, pixel_shuffle=ps, algo=algo))
        print(f"  {name:22s} {algo:9s} {ms:8.3f} ms  {fl/ms/1e9:7.1f} TF-equivalent")
xx = torch.randn(16, 64, 256, 256, device="cuda"); ww = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
rr = torch.randn(16, 64, 256, 256, device="cuda")
pd, pw = pack_conv3x3_weight(ww), pack_conv3x3_weight_wino(ww)
for algo, pk in (("direct", pd), ("winograd", pw)):
    print(f"  body + relu      {algo:9s} {t(lambda: conv3x3(xx, pk, 64, bias=b, relu=True, algo=algo)):8.3f} ms")
    print(f"  body + residual  {algo:9s} {t(lambda: conv3x3(xx, pk, 64, bias=b, residual=rr, algo=algo)):8.3f} ms")
import time
for w in (False, True):
    net.use_winograd = w
    net(x); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): net(x)
    torch.cuda.synchronize(); print(f"SRNet wall, winograd={w}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
