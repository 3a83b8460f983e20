#!/usr/bin/env python3
"""Collapsed upsampler (conv5x5 + PS4 + frame strips) vs the layer-by-layer up1/up2/tail, config-2 shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import SRNet, conv5x5_ps4
torch.cuda.set_device(0)
torch.manual_seed(0)
net = SRNet(3, 64, 16, 4).cuda().eval()
x = torch.rand(16, 3, 256, 256, device="cuda")
h = torch.randn(16, 64, 256, 256, device="cuda")
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def layerwise(h):
    for i, up in enumerate(net.up):
        h = net._conv(f"up{i}", up, h, pixel_shuffle=True)
    return net._conv("tail", net.tail, h)
a = layerwise(h); b = net._fused_up(net, h)
print("max |fused - layerwise| / max|layerwise| =", float((a - b).abs().max() / a.abs().max()))
print(f"layer-by-layer up1+up2+tail : {t(lambda: layerwise(h)):8.3f} ms")
print(f"collapsed (5x5+PS4 + frame) : {t(lambda: net._fused_up(net, h)):8.3f} ms")
fu = net._fused_up
ms = t(lambda: conv5x5_ps4(h, fu.w5, fu.cout5, bias=fu.b5))
fl = 2.0 * 25 * 64 * 48 * 16 * 256 * 256   # 48 real output channels: one 32-channel block (32x32x2 tiles) + 16 (16x16x4 tiles)
print(f"  conv5x5_ps4 kernel alone  : {ms:8.3f} ms  ({fl/ms/1e9:6.1f} TF issued = {fl/ms/1e9/157.3:4.2f} of the fp32 MFMA peak, no channel padding)")
for f in (False, True):
    net.fuse_upsampler = f
    print(f"SRNet fwd batch 16 @256^2, fuse_upsampler={f}: {t(lambda: net(x)):8.3f} ms")
