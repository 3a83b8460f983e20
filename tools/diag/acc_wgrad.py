"""Accuracy of the weight-gradient kernels at full config-3 sizes against float64 (tap-by-tap einsum on the GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kmsr_amd.sr import conv3x3_wgrad
torch.manual_seed(0)
for name, N, Cin, H, W, Cout in [("body", 32, 64, 256, 256, 64), ("up2", 8, 64, 512, 512, 256)]:
    x = torch.relu(torch.randn(N, Cin, H, W, device="cuda"))
    dy = torch.randn(N, Cout, H, W, device="cuda") * 0.01
    ref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device="cuda")
    for n in range(N):
        xp = torch.nn.functional.pad(x[n].double(), (1, 1, 1, 1))
        dd = dy[n].double().reshape(Cout, -1)
        for u in range(3):
            for v in range(3):
                ref[:, :, u, v] += dd @ xp[:, u:u + H, v:v + W].reshape(Cin, -1).t()
    sc = ref.abs().max()
    for tag, kw in (("F(4x4)", {}), ("F(2x2)", {"wino2": True}), ("direct", {"direct": True})):
        dw = conv3x3_wgrad(x, dy, **kw).double()
        print(f"{name} {tag}: max err / max|dw| = {float((dw - ref).abs().max() / sc):.3e}", flush=True)
