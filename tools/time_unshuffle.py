#!/usr/bin/env python3
"""Tail data gradient 5 -> 64 @1024^2 x32 (config 3): plain small-Cin conv + pixel_unshuffle2_sum pass vs the un-shuffling conv."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_for, pixel_unshuffle2_sum, conv3x3_smallci_unshuffle
torch.cuda.set_device(0)
x = torch.randn(32, 5, 1024, 1024, device="cuda"); w = torch.randn(64, 5, 3, 3, device="cuda") * 0.1
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
out = torch.empty(32, 64, 1024, 1024, device="cuda")
pw = pack_for("smallci", w)
a = t(lambda: conv3x3(x, pw, 64, out=out, algo="smallci"))
b = t(lambda: pixel_unshuffle2_sum(out))
del out
torch.cuda.empty_cache()
c = t(lambda: conv3x3_smallci_unshuffle(x, w))
print(f"plain conv {a:.3f} ms + un-shuffle pass {b:.3f} ms = {a + b:.3f} ms;   un-shuffling conv {c:.3f} ms")
