"""Config-3 training step, A/B in one process: last upsampler conv + tail as one autograd node (un-shuffled tail data gradient) vs separate nodes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
wl = bench.TrainWorkload(dev, 0)
for fuse in (True, False, True, False):
    wl.net.fuse_up_tail_backward = fuse
    for _ in range(2):
        wl.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        wl.step()
    e1.record()
    torch.cuda.synchronize()
    print(f"fuse_up_tail_backward={fuse}: {e0.elapsed_time(e1) / 5:.2f} ms per step", flush=True)
