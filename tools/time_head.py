#!/usr/bin/env python3
"""Few-input-channel 3x3 layers: SR head (3->64 @256^2 x16) and the tail's data gradient (5->64 @1024^2 x32): general direct
kernel vs the small-Cin kernel; rate against the output write (HBM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import conv3x3, pack_for
torch.cuda.set_device(0)
for name, (N, Cin, H, W, Cout) in (("head 3->64 @256^2 x16", (16, 3, 256, 256, 64)), ("head 5->64 @256^2 x32", (32, 5, 256, 256, 64)),
                                   ("tail dgrad 5->64 @1024^2 x32", (32, 5, 1024, 1024, 64))):
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1; b = torch.randn(Cout, device="cuda")
    out = torch.empty(N, Cout, H, W, device="cuda")
    for algo in ("direct", "smallci"):
        pw = pack_for(algo, w)
        conv3x3(x, pw, Cout, bias=b, out=out, algo=algo); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): conv3x3(x, pw, Cout, bias=b, out=out, algo=algo)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = (out.numel() + x.numel()) * 4 / 1e9
        print(f"{name:30s} {algo:8s} {ms*1e3:9.1f} us   {gb/ms:7.2f} TB/s algorithmic = {gb/ms/8:5.2f} of HBM peak")
