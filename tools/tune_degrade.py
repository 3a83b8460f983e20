#!/usr/bin/env python3
"""Interleaved A/B timing of degrade_batch variants on the bench workload (one process, several rounds)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
from kmsr_amd import _lib as L
from bench import DegradeWorkload

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "vb"
wl = DegradeWorkload(dev, 0, batch=int(os.environ.get("KMSR_TUNE_B", 32)), C=int(os.environ.get("KMSR_TUNE_C", 5)))   # about config 2 (48 planes): B=10 C=5 (the bank has 5 bands)
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "4,8,16,32,64".split(","))]
def run(rows):
    if mode == "vb":
        kmsr_amd.degrade_batch(wl.hr, wl.bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=wl.kid,
                               sigma=wl.sigma, noise="gaussian", seed=1, out=wl.lr, workspace=wl.ws, taps_ready=True,
                               rows_per_chunk=rows, flags=L.FLAG_FORCE_FAST)
    else:
        f = 4 if mode == "va4" else 8
        kmsr_amd.degrade_batch(wl.hr, wl.bank, f, kernel_idx=wl.kid, rows_per_chunk=rows, flags=L.FLAG_FORCE_FAST)
wl.step()
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        run(v); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(v)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20)
byt = wl.bytes_per_patch * wl.B
for v in variants:
    t = sorted(res[v]); med = t[len(t)//2]
    print(f"mode={mode} rows_per_chunk={v:4d}  median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us   {byt/med/1e6:8.1f} GB/s (alg)")
