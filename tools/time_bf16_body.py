#!/usr/bin/env python3
"""bf16 body layer (64->64 + residual @1024^2) only; KMSR_LIB_OVERRIDE selects a diagnostic build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr_bf16 import conv3x3_bf16, pack_conv3x3_weight_bf16
torch.cuda.set_device(0)
H = W = 1024
x = torch.randn(1, H, W, 64, device="cuda").to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
pw = pack_conv3x3_weight_bf16(w, 64, 64); b = torch.zeros(64, device="cuda")
r = torch.randn(1, H, W, 64, device="cuda").to(torch.bfloat16)
out = torch.empty_like(r)
for res in (r, None):
    conv3x3_bf16(x, pw, 64, bias=b, residual=res, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): conv3x3_bf16(x, pw, 64, bias=b, residual=res, out=out)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
    print(f"{os.environ.get('KMSR_LIB_OVERRIDE','default')[-20:]:22s} body 64->64 res={res is not None}: {ms*1e3:7.1f} us")
