#!/usr/bin/env python3
"""Run the stamped bf16 res-block build (KMSR_LIB_OVERRIDE=tools/diag/libkmsr_rb_stamped.so) at 1024 x 1024."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kmsr_amd import _lib as L
from kmsr_amd.sr_bf16 import pack_conv3x3_weight_bf16, resblock_bf16
x = torch.randn(1, 1024, 1024, 64, device="cuda").to(torch.bfloat16)
p1 = pack_conv3x3_weight_bf16(torch.randn(64, 64, 3, 3, device="cuda") / 24, 64, 64); p2 = pack_conv3x3_weight_bf16(torch.randn(64, 64, 3, 3, device="cuda") / 24, 64, 64)
b = torch.randn(64, device="cuda"); out = torch.empty_like(x)
for _ in range(3):
    resblock_bf16(x, p1, b, p2, b, out=out)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (256 * 4 * 16))()
lib.kmsr_diag_rb.argtypes = [ctypes.c_void_p]
assert lib.kmsr_diag_rb(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 4, 16)
n = np.median(a[..., 11])
names = ["wait + barrier A", "DMA issue (9 instr)", "conv1: 108 MFMAs (3 N-tiles)", "epilogue 1 -> t tile", "barrier B", "conv2: 72 MFMAs", "epilogue 2 -> staging", "barrier C + vmcnt(0)",
         "store phase", "loop top"]
print("tiles per workgroup", n)
tot = 0
for i, nm in enumerate(names):
    v = np.median(a[..., i]) / n; tot += v
    print(f"  {nm:34s} {v:7.0f} cycles per tile")
print(f"  {'sum':34s} {tot:7.0f}   (MFMA floor 180 x 33 = 5940); kernel total per wave {np.median(a[..., 10]):9.0f} cycles")
