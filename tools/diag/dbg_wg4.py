import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from kmsr_amd.sr import conv3x3_wgrad
for (N, Cin, H, W, Cout) in [(1, 96, 13, 100, 33), (1, 64, 13, 100, 33), (1, 96, 13, 100, 34), (1, 96, 12, 96, 33), (1,32,8,32,33)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    dy = torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    dw = conv3x3_wgrad(x.float().cuda(), dy.float().cuda()).double().cpu()
    err = (dw - w.grad).abs() / w.grad.abs().max()
    eco = err.amax(dim=(1, 2, 3)); eci = err.amax(dim=(0, 2, 3))
    print((N, Cin, H, W, Cout), "max", float(err.max()), "bad co:", [i for i in range(Cout) if eco[i] > 1e-4][:20], "bad ci:", [i for i in range(Cin) if eci[i] > 1e-4][:40])
