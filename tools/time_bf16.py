#!/usr/bin/env python3
"""BASELINE config 4: 7-band x4 SR, LR tile [1,7,1024,1024], bf16 -- per-layer and whole-net timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
from kmsr_amd.sr_bf16 import SRNetBF16, conv3x3_bf16, pack_conv3x3_weight_bf16
torch.cuda.set_device(0)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, H, W, Cin, Cout, ps, res in [("body 64->64 @1024^2", 1024, 1024, 64, 64, False, True), ("up1 64->256 ps @1024^2", 1024, 1024, 64, 256, True, False),
                                       ("up2 64->256 ps @2048^2", 2048, 2048, 64, 256, True, False), ("tail 64->16 @4096^2", 4096, 4096, 64, 16, False, False),
                                       ("head 16->64 @1024^2", 1024, 1024, 16, 64, False, False)]:
    x = torch.randn(1, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    pw = pack_conv3x3_weight_bf16(w, Cin, Cout); b = torch.zeros(max(Cout, 64), device="cuda")
    oshape = (1, 2 * H, 2 * W, Cout // 4) if ps else (1, H, W, Cout)
    r = torch.randn(*oshape, device="cuda").to(torch.bfloat16) if res else None
    out = torch.empty(oshape, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: conv3x3_bf16(x, pw, Cout, bias=b, residual=r, pixel_shuffle=ps, out=out))
    byt = 2.0 * (H * W * Cin + out.numel() * (2 if res else 1)); fl = 2.0 * 9 * Cin * Cout * H * W
    print(f"{name:26s} {ms:8.3f} ms   {byt/ms/1e6:8.1f} GB/s algorithmic ({byt/ms/1e6/8000*100:5.1f}% of HBM peak)   {fl/ms/1e9:8.1f} TFLOP/s")
torch.manual_seed(0)
net = kmsr_amd.SRNet(7, 64, 16, 4).cuda().eval()
ex = SRNetBF16(net)
x = torch.rand(1, 7, 1024, 1024, device="cuda") * 80 + 20
ms = t(lambda: ex(x), 3)
print(f"SRNet bf16 [1,7,1024,1024] -> [1,7,4096,4096]: {ms:8.2f} ms per tile ({16/ (ms/1e3):.1f} LR-256^2-patch equivalents/s)")
net.use_winograd = True
x32 = x.clone()
ms32 = t(lambda: net(x32), 2)
print(f"SRNet fp32 (Winograd) same tile: {ms32:8.2f} ms")
