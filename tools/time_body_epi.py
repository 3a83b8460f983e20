#!/usr/bin/env python3
"""Body conv 64->64 @256^2 x16/x32 with the training-step epilogue variants of the winograd_vreg kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_for
torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(B, 64, 256, 256, device="cuda"); r = torch.randn_like(x); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
pw = pack_for("winograd_vreg", w)
def t(name, **kw):
    conv3x3(x, pw, 64, algo="winograd_vreg", **kw); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): conv3x3(x, pw, 64, algo="winograd_vreg", **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1)/20*1e3:8.1f} us")
t("plain"); t("residual", residual=r); t("relu_mask", relu_mask=r); t("plain + chan_sum", chan_sum=True)
t("residual + chan_sum", residual=r, chan_sum=True); t("relu_mask + chan_sum", relu_mask=r, chan_sum=True)
