"""Does running the SR network over the 16-patch batch in SEQUENTIAL chunks keep the layer-to-layer activations in the 256 MiB
memory-side cache?  At batch 16 one 64-channel 256x256 fp32 feature map is 268 MB (in + out + skip = 805 MB per body layer); at
4 patches it is 67 MB (201 MB per layer).  Per-patch work is independent, so the chunked output must be bit-identical.
usage: python3 tools/diag/chunk_batch.py            (headline shape: LR [16,3,256,256], fused upsampler)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kmsr_amd  # noqa: F401
from kmsr_amd.sr import SRNet

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = SRNet(3, 64, 16, 4).to(dev).eval()
net.fuse_upsampler = True
B = 16
lr = torch.rand(B, 3, 256, 256, device=dev)


def run(chunk):
    with torch.no_grad():
        if chunk >= B:
            return net(lr)
        return torch.cat([net(lr[i:i + chunk]) for i in range(0, B, chunk)], dim=0)


def run_nocat(chunk):
    with torch.no_grad():
        for i in range(0, B, chunk):
            net(lr[i:i + chunk])


ref = run(16)
for chunk in (16, 8, 4, 2, 1, 16, 4):
    y = run(chunk)
    same = torch.equal(y, ref)
    for _ in range(2):
        run_nocat(chunk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        run_nocat(chunk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"chunk {chunk:2d}: {ms:7.3f} ms per 16 patches ({B / ms * 1e3:7.1f} patches/s)  bit-identical to the whole batch: {same}", flush=True)
