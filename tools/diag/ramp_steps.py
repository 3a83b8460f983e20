#!/usr/bin/env python3
"""Per-step HIP-event times of the headline workload from process start (is the first stretch of steps slower -- clock ramp -- and for how long?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl = bench.DegradeSRWorkload(dev, 0)
wl.net.fuse_upsampler = True
n = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
for i in range(n):
    ev[i].record()
    wl.step()
ev[n].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("per-step ms:", " ".join(f"{m:.2f}" for m in ms))
