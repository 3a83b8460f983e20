#!/usr/bin/env python3
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmsr_amd.sr import SRNet
torch.cuda.set_device(0)
net = SRNet(3, 64, 16, 4).cuda().eval()
x = torch.rand(16, 3, 256, 256, device="cuda")
net(x); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); net(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue (CPU) {1e3*(t1-t0):7.2f} ms, until GPU done {1e3*(t2-t0):7.2f} ms")
xs = torch.rand(1, 3, 16, 16, device="cuda")
net(xs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): net(xs)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"tiny input: CPU enqueue per forward {1e3*(t1-t0)/10:7.2f} ms ({1e3*(t1-t0)/10/36*1e3:6.1f} us per conv call)")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): net(xs)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
