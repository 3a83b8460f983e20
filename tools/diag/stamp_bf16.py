#!/usr/bin/env python3
"""Diagnostic: read the phase stamps of tools/diag/libkmsr_bf16_stamped.so (KMSR_LIB_OVERRIDE must point at it)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kmsr_amd import _lib as L
from kmsr_amd.sr_bf16 import conv3x3_bf16, pack_conv3x3_weight_bf16
torch.cuda.set_device(0)
H = W = 1024
x = torch.randn(1, H, W, 64, device="cuda").to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
pw = pack_conv3x3_weight_bf16(w, 64, 64); b = torch.zeros(64, device="cuda")
r = torch.randn(1, H, W, 64, device="cuda").to(torch.bfloat16)
out = torch.empty_like(r)
lib = L.load()
names = ["top(before barrier)", "after barrier", "after DMA issue", "after MFMA phase", "staging written", "after barrier 2", "after vmcnt(0)", "after stores"]
for res in (r, None):
    for _ in range(3): conv3x3_bf16(x, pw, 64, bias=b, residual=res, out=out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.kmsr_diag_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.kmsr_diag_stamps(buf, 64) == 0
    print(f"residual={res is not None}: s_memtime deltas (100 MHz ticks x 10 ns) per phase, items 2..9 of workgroup 0 wave 0")
    for it in range(8):
        s = [buf[it * 8 + k] for k in range(8)]
        nxt = buf[(it + 1) * 8] if it < 7 else None
        d = [s[k + 1] - s[k] for k in range(7)] + ([nxt - s[7]] if nxt else [])
        print("  item", it + 2, " ".join(f"{v:5d}" for v in d), " total", (nxt - s[0]) if nxt else "-")
print("columns:", " | ".join(f"{names[k]} -> {names[(k + 1) % 8]}" for k in range(8)))
