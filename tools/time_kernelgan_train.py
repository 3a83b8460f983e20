#!/usr/bin/env python3
"""KernelGAN trainer iteration at the reference's geometry (muti_kernel/train.py:199-213: batch 8 x [5,256,256] HR, 32x32 LR;
single_kernel/train.py:136-141: batch 16): wall time per iteration and the three convk legs of the 5x5 32->32 layer alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmsr_amd
torch.cuda.set_device(0)
torch.manual_seed(0)
dev = "cuda"


def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


model = kmsr_amd.DegradationModel(noise_init=0.3, noise_max=1.2).cuda()
D = kmsr_amd.PatchDiscriminator(5).cuda()
oD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999)); oG = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
x = torch.rand(8, 5, 256, 256, device=dev); r = torch.rand(8, 5, 32, 32, device=dev)
ts = torch.tensor([0.55, 0.72, 0.83, 0.63, 0.19], device=dev)
ms = timed(lambda: kmsr_amd.kernelgan_multi_step(model, D, oG, oD, x, r, ts, 20.0))
print(f"multi-kernel trainer iteration  (batch 8, 40 chains @256^2): {ms:8.2f} ms  -> {8 / ms * 1e3:7.1f} HR patches/s")
G = kmsr_amd.MultiBandLinearGenerator(5, 32).cuda(); D2 = kmsr_amd.PatchDiscriminator(5).cuda()
oD2 = torch.optim.Adam(D2.parameters(), lr=4e-4, betas=(0.5, 0.999)); oG2 = torch.optim.Adam(G.parameters(), lr=4e-4, betas=(0.5, 0.999))
x16 = torch.rand(16, 5, 256, 256, device=dev); r16 = torch.rand(16, 5, 32, 32, device=dev)
ms = timed(lambda: kmsr_amd.kernelgan_single_step(G, D2, oG2, oD2, x16, r16, 0.002, 20.0))
print(f"single-kernel trainer iteration (batch 16, 80 chains @256^2): {ms:8.2f} ms  -> {16 / ms * 1e3:7.1f} HR patches/s")
U = 40
h = torch.rand(U, 32, 256, 256, device=dev); w = torch.rand(U, 32, 25, 32, device=dev) * 0.01; g = torch.rand(U, 32, 256, 256, device=dev)
flops = 2.0 * U * 32 * 25 * 32 * 256 * 256
for name, fn, how in (("fwd  ", lambda: kmsr_amd.convk_forward(h, w, 5), "packed fp32 VALU, 2 px / thread"),
                      ("dgrad", lambda: kmsr_amd.convk_dgrad(g, w, 5), "padded-domain pass + fold"),
                      ("wgrad", lambda: kmsr_amd.convk_wgrad(h, g, 5), "fp32 MFMA")):
    ms = timed(fn, 10)
    print(f"convk {name} 5x5 32->32, 40 units @256^2: {ms:7.3f} ms  = {flops / ms / 1e9:6.1f} TFLOP/s ({how})")
