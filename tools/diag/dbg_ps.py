import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from kmsr_amd.sr import conv3x3, pack_for
torch.manual_seed(0)
N, Cin, H, W, Cout = 1, 64, 24, 40, 256
x = torch.randn(N, Cin, H, W); w = torch.randn(Cout, Cin, 3, 3) * 0.05
ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), padding=1), 2)
out = torch.full((N, Cout // 4, 2 * H, 2 * W), 777.0, device="cuda")
conv3x3(x.cuda(), pack_for("winograd_vreg", w.cuda()), Cout, pixel_shuffle=True, algo="winograd_vreg", out=out)
bad = ((out.cpu().double() - ref).abs() > 1e-3)
print("bad", int(bad.sum()), "untouched", int((out == 777.0).sum()))
idx = bad.nonzero()
print("channels", idx[:, 1].unique().tolist()[:20]); print("rows", idx[:, 2].unique().tolist()[:60]); print("cols", idx[:, 3].unique().tolist()[:90])
o = out.cpu().double()
for k in range(0, min(len(idx), 400), 57):
    n, c, r, cc = idx[k].tolist()
    v = o[n, c, r, cc].item()
    m = (ref - v).abs() < 1e-4
    where = m.nonzero()[:3].tolist()
    print((c, r, cc), "got %.4f want %.4f" % (v, ref[n, c, r, cc].item()), "equals ref at", where)
