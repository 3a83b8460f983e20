"""Weight-gradient kernels on the config-3 layer shapes: F(4x4,3x3) domain (default) vs F(2x2,3x3) (wino2) vs direct.
usage: python tools/time_wgrad.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmsr_amd.sr import conv3x3_wgrad

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ONLY = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. "body": that layer only, F(4x4) only
shapes = [("body 64->64 @256", B, 64, 256, 256, 64), ("up1 64->256 @256", B, 64, 256, 256, 256), ("up2 64->256 @512", B, 64, 512, 512, 256),
          ("head 5->64 @256", B, 5, 256, 256, 64)]
for name, N, Cin, H, W, Cout in shapes:
    if ONLY and not name.startswith(ONLY):
        continue
    x = torch.randn(N, Cin, H, W, device="cuda")
    dy = torch.randn(N, Cout, H, W, device="cuda") * 0.01
    ref = None
    for tag, kw in (("F(4x4)", {}), ("F(2x2)", {"wino2": True}), ("direct", {"direct": True})):
        if (tag == "direct" and Cout == 256 and H == 512) or (ONLY and tag != "F(4x4)"):
            continue
        for _ in range(2):
            dw = conv3x3_wgrad(x, dy, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dw = conv3x3_wgrad(x, dy, **kw)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = dw
        print(f"{name:18s} {tag:7s} {e0.elapsed_time(e1) / 5:8.3f} ms   max|diff vs F(4x4)| / max|dw| = {float((dw - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
