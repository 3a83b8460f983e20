import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kmsr_amd import _lib as L
from kmsr_amd.sr import conv3x3, pack_for
mode = sys.argv[1]
x = torch.randn(16, 64, 256, 256, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
r = torch.randn(16, 64, 256, 256, device="cuda"); out = torch.empty_like(r)
pw = pack_for("winograd4", w)
kw = dict(residual=r) if mode == "res" else dict(relu=True)
for _ in range(30): conv3x3(x, pw, 64, bias=b, out=out, algo="winograd4", **kw)
torch.cuda.synchronize()
lib = L.load(); buf = (ctypes.c_ulonglong * (256 * 4 * 16))()
lib.kmsr_diag_w4.argtypes = [ctypes.c_void_p]; assert lib.kmsr_diag_w4(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 4, 16); n = np.median(a[..., 11])
names = ["epilogue start: offsets, skip requests / waits", "sub_block transform (x8)", "skip rows into registers (x8)", "add + stores (x8)"]
print("mode", mode)
for i, nm in enumerate(names): print(f"  {nm:50s} {np.median(a[..., i]) / n:8.0f} cycles per item")
print(f"  kernel body per wave {np.median(a[..., 10]):9.0f} cycles")
