#!/usr/bin/env python3
"""Run the stamped F(4x4,3x3) build (KMSR_LIB_OVERRIDE=tools/diag/libkmsr_w4_stamped.so) on the body layer and print per-wave cycle sums."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kmsr_amd import _lib as L
from kmsr_amd.sr import conv3x3, pack_for
mode = sys.argv[1] if len(sys.argv) > 1 else "res"
x = torch.randn(16, 64, 256, 256, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
r = torch.randn(16, 64, 256, 256, device="cuda"); out = torch.empty_like(r)
pw = pack_for("winograd4", w)
kw = dict(residual=r) if mode == "res" else dict(relu=True)
for _ in range(30):
    conv3x3(x, pw, 64, bias=b, out=out, algo="winograd4", **kw)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (256 * 4 * 8))()
lib.kmsr_diag_w4.argtypes = [ctypes.c_void_p]
assert lib.kmsr_diag_w4(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 4, 8)
items = a[..., 5]
print("mode", mode, "items per workgroup", items.min(), items.max())
for i, nm in enumerate(["wait+barrier", "steps (not first)", "epilogues", "first steps", "kernel body total"]):
    v = a[..., i]
    print(f"{nm:20s} median {np.median(v):10.0f}  min {v.min():10.0f}  max {v.max():10.0f} cycles per wave")
tot = np.median(a[..., 4]); n = np.median(items)
print(f"per step: barrier {np.median(a[...,0])/(16*n):.0f}, step body {np.median(a[...,1])/(15*n):.0f}, first-step body {np.median(a[...,3])/n:.0f}; per item epilogue {np.median(a[...,2])/n:.0f}; "
      f"MFMA floor per step 2304; sum/total = {(np.median(a[...,0])+np.median(a[...,1])+np.median(a[...,2])+np.median(a[...,3]))/tot:.3f}")
