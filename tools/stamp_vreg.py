#!/usr/bin/env python3
"""Diagnostic only: per-section s_memtime stamps of the V-in-registers Winograd kernel (needs a stamped build of the library,
KMSR_LIB_OVERRIDE=tools/diag/libkmsr_stamp.so; the product library contains no stamps).  Prints the median cycle count of
every section over all waves, how many workgroups overlap in time on one CU, and the shares of a wave's lifetime."""
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
WAVES = 4
nblocks = N * (H // 2) * (W // 64)
dbg = torch.zeros(nblocks * WAVES * 64, dtype=torch.int64, device="cuda")
for _ in range(3): conv3x3(x, pw, Cout, bias=b, out=out, algo="winograd_vreg", residual=r)
lib.kmsr_debug_set_buffer.argtypes = [C.c_void_p]
lib.kmsr_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
conv3x3(x, pw, Cout, bias=b, out=out, algo="winograd_vreg", residual=r)
torch.cuda.synchronize()
lib.kmsr_debug_set_buffer(C.c_void_p(0))
t = dbg.cpu().numpy().reshape(nblocks, WAVES, 64).astype(np.int64)
nch = Cin // 4
def d(a, b): return np.median(t[:, :, b] - t[:, :, a])
print(f"prologue: start->first DMA wait {d(0,1):8.0f}  barrier {d(1,2):8.0f}  transform(0) + wait U(0) {d(2,3):8.0f}")
bar = [d(4 + 2 * c, 5 + 2 * c) for c in range(nch)]
body = [d(5 + 2 * c, (4 + 2 * (c + 1)) if c + 1 < nch else 40) for c in range(nch)]
print("per-chunk barrier wait:", " ".join(f"{v:5.0f}" for v in bar))
print("per-chunk body        :", " ".join(f"{v:5.0f}" for v in body))
print(f"epilogue: col transform + skip read {d(40,41):7.0f}  barrier {d(41,42):7.0f}  xch write {d(42,43):7.0f}  sync {d(43,44):7.0f}  finalise+stores {d(44,45):7.0f}")
tot = np.median(t[:, :, 45] - t[:, :, 0])
print(f"total per wave {tot:8.0f} cycles; kernel span {(t[:, :, 45].max() - t[:, :, 0].min())} ticks")
print(f"sum of chunk bodies {sum(body):8.0f} = {sum(body) / tot:.2f}, barriers {sum(bar):7.0f} = {sum(bar)/tot:.2f} of the wave lifetime (pure MFMA issue at 2 waves/SIMD: {nch * 16 * 64 * 2})")
# residency: group workgroups by CU = (XCC id, SE, SH, CU) and look at how their lifetimes overlap
hw, xcc = t[:, 0, 63], t[:, 0, 62] & 0xF
cu_key = (xcc << 8) | ((hw >> 8) & 0xFF)
import collections
by = collections.defaultdict(list)
for blk in range(nblocks):
    by[int(cu_key[blk])].append((int(t[blk, :, 0].min()), int(t[blk, :, 45].max()), int(np.median(t[blk, :, 3])), int(np.median(t[blk, :, 40])), blk))
co, both_body, any_body, life = [], 0, 0, 0
for k, lst in by.items():
    lst.sort()
    for (s0, e0, b0, b1, blk) in lst:
        mid = (s0 + e0) // 2
        co.append(sum(1 for (s2, e2, _, _, _) in lst if s2 <= mid <= e2))
    # fraction of the CU's busy span in which 0 / 1 / 2 workgroups are inside their MFMA bodies
    pts = sorted([(b0, 1) for (_, _, b0, b1, _) in lst] + [(b1, -1) for (_, _, b0, b1, _) in lst])
    cur, last = 0, pts[0][0]
    hist = collections.Counter()
    for (tt, dlt) in pts:
        hist[cur] += tt - last
        last, cur = tt, cur + dlt
    span = lst[-1][1] - lst[0][0]
    life += span
    both_body += hist[2] + hist[3] + hist[4]
    any_body += hist[1]
print(f"CUs seen {len(by)}; workgroups per CU {nblocks / len(by):.1f}; alive at a workgroup's midpoint on its CU: median {np.median(co):.0f}, mean {np.mean(co):.2f}")
print(f"of the CUs' busy span: two+ workgroups inside their MFMA bodies {both_body / life:.2f}, exactly one {any_body / life:.2f}, none {1 - (both_body + any_body) / life:.2f}")
k0 = sorted(by)[0]
print("first CU timeline (start, body start, body end, end) relative:", [(a - by[k0][0][0], b - by[k0][0][0], c - by[k0][0][0], e - by[k0][0][0]) for (a, e, b, c, _) in by[k0][:6]])
