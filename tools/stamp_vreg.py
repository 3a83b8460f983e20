#!/usr/bin/env python3
"""Diagnostic only: per-section s_memtime stamps of the persistent V-in-registers Winograd kernel (needs a stamped build of the
library, KMSR_LIB_OVERRIDE=tools/diag/libkmsr_stamp.so; the product library contains no stamps)."""
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
use_res = (sys.argv[1] if len(sys.argv) > 1 else "res") == "res"
pw = pack_conv3x3_weight_wino_vreg(w)
nwg, WAVES, S = 256, 8, 256
dbg = torch.zeros(nwg * WAVES * S, dtype=torch.int64, device="cuda")
kw = dict(residual=r) if use_res else dict(relu=True)
for _ in range(3): conv3x3(x, pw, Cout, bias=b, out=out, algo="winograd_vreg", **kw)
lib.kmsr_debug_set_buffer.argtypes = [C.c_void_p]
lib.kmsr_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
conv3x3(x, pw, Cout, bias=b, out=out, algo="winograd_vreg", **kw)
torch.cuda.synchronize()
lib.kmsr_debug_set_buffer(C.c_void_p(0))
t = dbg.cpu().numpy().reshape(nwg, WAVES, S).astype(np.int64)
def d(a, b): return t[:, :, b] - t[:, :, a]
def fmt(v): return f"med {np.median(v):6.0f} mean {np.mean(v):6.0f} p90 {np.percentile(v, 90):6.0f}"
print("prologue (start -> raw(0) landed)", fmt(d(0, 1)), " barrier", fmt(d(1, 2)), " to loop", fmt(d(2, 8)))
G = 48
s0 = np.stack([d(8 + 5 * g, 9 + 5 * g) for g in range(G)]); wt = np.stack([d(9 + 5 * g, 10 + 5 * g) for g in range(G)])
br = np.stack([d(10 + 5 * g, 11 + 5 * g) for g in range(G)]); s1 = np.stack([d(11 + 5 * g, 12 + 5 * g) for g in range(G)])
ep = np.stack([d(12 + 5 * g, 13 + 5 * g) for g in range(G - 1)])
print("step 0      ", fmt(s0)); print("dma wait    ", fmt(wt)); print("barrier     ", fmt(br)); print("step 1      ", fmt(s1))
ends = [g for g in range(G - 1) if g % 8 == 7]
print("epilogue (last chunk of an item -> next chunk top)", fmt(ep[ends])); print("other chunk gaps", fmt(ep[[g for g in range(G - 1) if g % 8 != 7]]))
per_chunk = np.stack([d(8 + 5 * g, 8 + 5 * (g + 1)) for g in range(G - 1)])
print("whole chunk period", fmt(per_chunk), " (pure MFMA issue at 2 waves/SIMD: 4096)")
print("kernel span per workgroup", fmt(d(0, 7)))
w0 = t[3, 2]
print("raw stamps of block 3 wave 2, chunks 8..10 (relative):", [int(w0[8 + 6 * g + k] - w0[8 + 6 * 8]) for g in (8, 9, 10) for k in range(6)])
for k, nm in enumerate(["step0", "wait", "barrier", "resfetch", "step1", "tail/epilogue"]):
    v = np.stack([t[:, :, 9 + 6 * g + k] - t[:, :, 8 + 6 * g + k] for g in range(8, 32)])
    print(f"{nm:14s}", fmt(v))
