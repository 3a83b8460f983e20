#!/usr/bin/env python3
"""Tail layer (64 -> 5 @1024^2 x32) weight gradient (small-Cout kernel) and forward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, conv3x3_wgrad, pack_for
torch.cuda.set_device(0)
x = torch.randn(32, 64, 1024, 1024, device="cuda"); dy = torch.randn(32, 5, 1024, 1024, device="cuda")
w = torch.randn(5, 64, 3, 3, device="cuda") * 0.05
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
gb = x.numel() * 4 / 1e9
ms = t(lambda: conv3x3_wgrad(x, dy)); print(f"tail wgrad 64->5 @1024^2 x32: {ms:7.3f} ms = {gb/ms:5.2f} TB/s = {gb/ms/8:4.2f} of HBM peak (reads x once)")
pw = pack_for("direct", w)
ms = t(lambda: conv3x3(x, pw, 5)); print(f"tail fwd   64->5 @1024^2 x32: {ms:7.3f} ms = {gb/ms:5.2f} TB/s = {gb/ms/8:4.2f} of HBM peak")
# the headline tail (config 2): 64 -> 3 @1024^2 x16
x3 = x[:16]; w3 = torch.randn(3, 64, 3, 3, device="cuda") * 0.05; pw3 = pack_for("direct", w3); gb3 = x3.numel() * 4 / 1e9
ms = t(lambda: conv3x3(x3, pw3, 3)); print(f"tail fwd   64->3 @1024^2 x16: {ms:7.3f} ms = {gb3/ms:5.2f} TB/s = {gb3/ms/8:4.2f} of HBM peak")
