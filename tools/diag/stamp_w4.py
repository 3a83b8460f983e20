#!/usr/bin/env python3
"""Run the stamped F(4x4,3x3) build (KMSR_LIB_OVERRIDE=tools/diag/libkmsr_w4_stamped.so) on the body layer and print per-phase cycle sums."""
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
buf = (ctypes.c_ulonglong * (256 * 4 * 16))()
lib.kmsr_diag_w4.argtypes = [ctypes.c_void_p]
assert lib.kmsr_diag_w4(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 4, 16)
n = np.median(a[..., 11]); steps = 16 * n
names = ["wait + barrier", "MFMA pairs 0-15 (+ raw request, 12 patch reads, 8 A loads, cursors)", "column pass (36 pk VALU; waits for the patch)", "MFMA pairs 16-19 (+2 A loads)",
         "row pass 0-2 (21 pk)", "MFMA pairs 20-35 (+8 A loads)", "row pass 3-5 (21 pk)", "loop tail"]
print("mode", mode, "items per workgroup", n)
tot = 0
for i, nm in enumerate(names):
    v = np.median(a[..., i]) / steps; tot += v
    print(f"  {nm:48s} {v:7.0f} cycles per step")
print(f"  {'sum':48s} {tot:7.0f}   (MFMA floor 2304 = 1024 + 256 + 1024)")
print(f"  epilogue per item {np.median(a[..., 9]) / n:7.0f}; kernel body total per wave {np.median(a[..., 10]):9.0f} cycles")
