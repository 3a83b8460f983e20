#!/usr/bin/env python3
"""Diagnostic only: timing-only ablation builds of degrade_bwd.hip's persistent V-B kernel (results are WRONG by construction):
tools/diag/libkmsr_dbwd_<name>.so, loaded with KMSR_LIB_OVERRIDE.  Never part of the product library."""
import os, subprocess, sys
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
F = "degrade_bwd.hip"
src = open(f"{R}/{F}").read()
def build(name, edits):
    s = src
    for a, b in edits:
        assert a in s, (name, a[:60])
        s = s.replace(a, b)
    d = f"/tmp/diag_dbwd_{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/{F}", "w").write(s)
    for h in ("kmsr_common.h", "kmsr_noise.h"):
        open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
    subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c {F} -o x.o", shell=True)
    objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != F)
    subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_dbwd_{name}.so", shell=True)
    print("built", name)
NOLOAD = [("                const bool ok = c.live && cq[k] < 4 * VB_AQ && gy >= 0", "                const bool ok = c.live && p.H == 12345 && cq[k] < 4 * VB_AQ && gy >= 0")]
NOTAPS = [("            if (s < VB_S) {\n                const float* xr0", "            if (s < VB_S && p.H == 12345) {\n                const float* xr0")]
# keep the LDS reads, 4 multiply-adds per 4 samples instead of 16
NOFMA = [("                            for (int tt = 0; tt < 4; ++tt) acc[tt] = fmaf(gk[k], xv[4 * q + k + tt], acc[tt]);",
          "                            for (int tt = 0; tt < 1; ++tt) acc[k] += gk[k] * xv[4 * q + k + 3 * (q & 1)];")]
# keep the FMAs, 5 LDS reads per row instead of 19 (the other samples are copies)
NOLDS = [("                    for (int j = 0; j < BT + 3; ++j) xv[j] = xr[VB_F * j];", "                    for (int j = 0; j < BT + 3; ++j) xv[j] = xr[VB_F * (j & 3)];")]
NOSIGMA = [("            } else if (p.want_sigma && wv == 7) {", "            } else if (p.want_sigma && wv == 7 && p.H == 12345) {")]
ABL = {"noload": NOLOAD, "notaps": NOTAPS, "nofma": NOFMA, "nolds": NOLDS, "nosigma": NOSIGMA}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
