import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kmsr_amd.sr_bf16 import conv3x3_bf16, pack_conv3x3_weight_bf16, resblock_bf16
torch.manual_seed(0)
H, W = 8, 16
x = (torch.randn(1, H, W, 64, device="cuda") * 2).to(torch.bfloat16)
w1 = torch.randn(64, 64, 3, 3, device="cuda") / 24; w2 = torch.zeros(64, 64, 3, 3, device="cuda")
for c in range(64): w2[c, c, 1, 1] = 1.0          # conv2 = identity: out = x + t + b2
b1 = torch.randn(64, device="cuda"); b2 = torch.zeros(64, device="cuda")
p1, p2 = pack_conv3x3_weight_bf16(w1, 64, 64), pack_conv3x3_weight_bf16(w2, 64, 64)
t = conv3x3_bf16(x, p1, 64, bias=b1, relu=True)
want = conv3x3_bf16(t, p2, 64, bias=b2, residual=x)
got = resblock_bf16(x, p1, b1, p2, b2)
d = (got.view(torch.int16) != want.view(torch.int16))[0]
print("differ", int(d.sum()), "of", d.numel())
print("per channel:", d.sum(dim=(0, 1)).tolist())
print("per row:", d.sum(dim=(1, 2)).tolist())
print("per col:", d.sum(dim=(0, 2)).tolist())
idx = d.nonzero()[:8]
for y, xx, c in idx.tolist():
    print((y, xx, c), "got", float(got[0, y, xx, c]), "want", float(want[0, y, xx, c]), "t", float(t[0, y, xx, c]), "x", float(x[0, y, xx, c]))
