import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import kmsr_amd
from kmsr_amd import _lib as L
from bench import DegradeSRWorkload, DegradeWorkload
torch.cuda.set_device(0)
for WL in (DegradeSRWorkload, DegradeWorkload):
    wl = WL(torch.device("cuda", 0), 0)
    by = wl.deg_bytes_per_launch if hasattr(wl, "deg_bytes_per_launch") else wl.bytes_per_patch * wl.B
    for flags in (0, 0x100):
        for rows in (0, 4, 8, 16, 32):
            def run():
                kmsr_amd.degrade_batch(wl.hr, wl.bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=wl.kid, sigma=wl.sigma,
                                       noise="gaussian", seed=1, out=wl.lr, workspace=wl.ws, taps_ready=False, rows_per_chunk=rows,
                                       flags=L.FLAG_FORCE_FAST | flags)
            run(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
            ms = sorted(ts)[2]
            print(f"{WL.__name__:18s} NX={'1' if flags else 'auto'} rows={rows:2d}: {ms*1e3:7.1f} us  {by/ms/1e6:7.0f} GB/s  ({by/ms/1e6/8000:.3f})")
