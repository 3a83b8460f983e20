#!/usr/bin/env python3
"""Run the stamped F(4x4,3x3) weight-gradient build (KMSR_LIB_OVERRIDE=tools/diag/libkmsr_wg4_stamped.so) on the body layer at batch 32."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kmsr_amd import _lib as L
from kmsr_amd.sr import conv3x3_wgrad
x = torch.randn(32, 64, 256, 256, device="cuda"); dy = torch.randn(32, 64, 256, 256, device="cuda") * 0.01
for _ in range(3):
    conv3x3_wgrad(x, dy)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (512 * 4 * 16))()
lib.kmsr_diag_wg4.argtypes = [ctypes.c_void_p]
assert lib.kmsr_diag_wg4(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(512, 4, 16)[:256]
n = np.median(a[..., 11])
names = ["Vhat0 + Ahat0 (59 VALU; waits for their raw reads)", "group (0,0): 18 MFMAs (+ 9 DMA in K-step 0)", "Vhat1 (39 VALU)", "group (0,1): 18 MFMAs (+ 7 DMA)", "Ahat1 (20 VALU)",
         "groups (1,0), (1,1): 36 MFMAs (+ next raw reads)", "-", "stage tail", "flush", "wait + barrier"]
print("stages per workgroup", n)
for ps in (0, 1):
    tot = 0
    print(f" point half {ps} (waves {ps}, {ps + 2}):")
    for i, nm in enumerate(names):
        v = np.median(a[:, ps::2, i]) / n; tot += v
        print(f"  {nm:44s} {v:7.0f} cycles per stage")
    print(f"  {'sum':44s} {tot:7.0f}   (MFMA floor 4608, + 236 packed VALU); kernel total per wave {np.median(a[:, ps::2, 10]):9.0f} cycles")
