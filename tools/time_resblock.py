"""bf16 residual block at the config-4 tile size: fused launch (kmsr_resblock_fwd_bf16) vs two conv3x3_bf16 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmsr_amd.sr_bf16 import conv3x3_bf16, pack_conv3x3_weight_bf16, resblock_bf16
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(1, H, W, 64, device="cuda").to(torch.bfloat16)
if os.environ.get("KMSR_ZERO_INPUT"):       # clock probe: same launches, all-zero activations
    x.zero_()
w1, w2 = torch.randn(64, 64, 3, 3, device="cuda") / 24, torch.randn(64, 64, 3, 3, device="cuda") / 24
b1, b2 = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
p1, p2 = pack_conv3x3_weight_bf16(w1, 64, 64), pack_conv3x3_weight_bf16(w2, 64, 64)
t, y, z = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
def two():
    conv3x3_bf16(x, p1, 64, bias=b1, relu=True, out=t)
    conv3x3_bf16(t, p2, 64, bias=b2, residual=x, out=y)
def one():
    resblock_bf16(x, p1, b1, p2, b2, out=z)
for name, fn in (("two launches", two), ("fused", one)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:14s} {ms * 1e3:8.1f} us per block   ({2 * 2 * 9 * 64 * 64 * H * W / ms / 1e9:7.1f} TFLOP/s algorithmic, {2 * 64 * 2 * H * W / ms / 1e6:6.0f} GB/s of x-in + y-out)")
print("equal:", torch.equal(y, z))
