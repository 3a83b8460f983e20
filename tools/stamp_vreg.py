#!/usr/bin/env python3
"""Diagnostic only: per-section s_memtime stamps of the V-in-registers Winograd kernel (needs a stamped build of the library,
KMSR_LIB_OVERRIDE=tools/diag/libkmsr_stamp.so; the product library contains no stamps).  Prints the median cycle count of
every section over all waves and the shares of a workgroup's lifetime."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd import _lib as L
from kmsr_amd.sr import conv3x3, pack_conv3x3_weight_wino_vreg
torch.cuda.set_device(0)
lib = L.load()
N, Cin, H, W, Cout = 16, 64, 256, 256, 64
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
r = torch.randn(N, Cout, H, W, device="cuda"); out = torch.empty_like(r)
pw = pack_conv3x3_weight_wino_vreg(w)
nblocks = N * (H // 4) * (W // 64)
dbg = torch.zeros(nblocks * 8 * 64, dtype=torch.int64, device="cuda")
for _ in range(3): conv3x3(x, pw, Cout, bias=b, out=out, algo="winograd_vreg", residual=r)
lib.kmsr_debug_set_buffer.argtypes = [C.c_void_p]
lib.kmsr_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
conv3x3(x, pw, Cout, bias=b, out=out, algo="winograd_vreg", residual=r)
torch.cuda.synchronize()
lib.kmsr_debug_set_buffer(C.c_void_p(0))
t = dbg.cpu().numpy().reshape(nblocks, 8, 64).astype(np.int64)
nch = Cin // 8
def d(a, b): return np.median(t[:, :, b] - t[:, :, a])
print(f"prologue: start->DMA wait done {d(0,1):8.0f}  barrier {d(1,2):8.0f}  transform(0) {d(2,3):8.0f}")
for c in range(nch):
    nxt = 4 + 4 * (c + 1) if c + 1 < nch else 40
    print(f"chunk {c}: vmcnt wait {d(4+4*c,5+4*c):7.0f}  barrier {d(5+4*c,6+4*c):7.0f}  body {d(6+4*c,nxt):7.0f}")
print(f"epilogue: col transform + residual read {d(40,41):7.0f}  barrier {d(41,42) - 0:7.0f}(incl. xch write)  sync {d(42,43):7.0f}  finalise+stores {d(43,44):7.0f}")
tot = np.median(t[:, :, 44] - t[:, :, 0])
print(f"total per wave {tot:8.0f} cycles; first stamp spread within kernel: {(t[:, :, 0].max() - t[:, :, 0].min())} ticks; kernel span {(t[:, :, 44].max() - t[:, :, 0].min())}")
body = sum(d(6 + 4 * c, (4 + 4 * (c + 1) if c + 1 < nch else 40)) for c in range(nch))
print(f"sum of chunk bodies {body:8.0f} = {body / tot:.2f} of the wave lifetime (pure MFMA issue would be {nch * 32 * 64 * 2} cycles at 2 waves/SIMD)")
