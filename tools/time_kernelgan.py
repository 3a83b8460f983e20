#!/usr/bin/env python3
"""BASELINE config 5: batched KernelGAN kernel estimation, 512 GOCI-shaped patches [5,256,256] (one GPU's view)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
torch.cuda.set_device(0)
torch.manual_seed(0)
g = kmsr_amd.DynamicMultiBandLinearGenerator().cuda()
with torch.no_grad():
    for band in g.weight_bands:
        for w in band: w.mul_(30.0)
N = 512
x = torch.rand(N, 5, 256, 256, device="cuda")
BATCHES = tuple(int(a) for a in sys.argv[1:]) or (8, 64, 512)      # e.g. `time_kernelgan.py 512` under rocprofv3 for the bench shape alone
for bs in BATCHES:
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = [g.extract_effective_kernels(x[s:s + bs], reduce_batch=False) for s in range(0, N, bs)]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch {bs:4d}: {N} patches in {dt*1e3:8.2f} ms  -> {N/dt:10.1f} patches/s  (reference CPU path: 0.58 patches/s/core)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
sc = g.condition_encoder.scales_tensor(x)
co, ci, ks = g._layer_dims(); flat = g._flat_weights()
kmsr_amd.compose_kernels(flat, sc, co, ci, ks); e0.record()
for _ in range(5): kmsr_amd.compose_kernels(flat, sc, co, ci, ks)
e1.record(); torch.cuda.synchronize()
print(f"compose kernel alone (2560 workgroups): {e0.elapsed_time(e1)/5*1e3:.1f} us")
