#!/usr/bin/env python3
"""Diagnostic only: timing-only ablation builds of conv3x3_wino_vreg.hip (results are WRONG by construction):
tools/diag/libkmsr_abl_<name>.so.  Never part of the product library."""
import os, subprocess, sys
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
src = open(f"{R}/conv3x3_wino_vreg.hip").read()
def build(name, edits):
    s = src
    for a, b in edits:
        assert a in s, (name, a[:50])
        s = s.replace(a, b)
    d = f"/tmp/diag_{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/conv3x3_wino_vreg.hip", "w").write(s)
    for h in ("kmsr_common.h", "kmsr_noise.h"):
        open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
    subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wino_vreg.hip -o conv3x3_wino_vreg.o", shell=True)
    objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wino_vreg.hip")
    subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/conv3x3_wino_vreg.o -o /root/repo/tools/diag/libkmsr_abl_{name}.so", shell=True)
    print("built", name)
ABL = {
    "nodma": [("        if (issue) issue_u(0, 2); ", "        if (false) issue_u(0, 2); "), ("        if (issue) issue_u(2, UINSTR);", "        if (false) issue_u(2, UINSTR);"),
              ("        if (issue) issue_raw();", "        if (false) issue_raw();"), ("        wait_dma<RINSTR>();  ", "        wait_dma<0>();  ")],
    "nostore": [("            store_outputs(ip, 0, Y0, rr0);", "            if (p.relu == 12345) store_outputs(ip, 0, Y0, rr0); else asm volatile(\"\" :: \"v\"(Y0[0][0]), \"v\"(Y0[1][3]));"),
                ("            store_outputs(ip, 1, Y1, rr1);", "            if (p.relu == 12345) store_outputs(ip, 1, Y1, rr1); else asm volatile(\"\" :: \"v\"(Y1[0][0]), \"v\"(Y1[1][3]));")],
    "nobarrier": [("        wait_dma<RINSTR>();                                  // outstanding: U(g+1), raw(g+3): everything but the newest raw chunk has landed\n        barrier_lds();",
                   "        wait_dma<RINSTR>();")],
    "noepi": [("            outputs_of(ip, 0, Y0);\n            fetch_residual(ip, 1, rr1);\n            store_outputs(ip, 0, Y0, rr0);\n            outputs_of(ip, 1, Y1);\n            store_outputs(ip, 1, Y1, rr1);",
               "            if (p.relu == 12345) { outputs_of(ip, 0, Y0); fetch_residual(ip, 1, rr1); store_outputs(ip, 0, Y0, rr0); outputs_of(ip, 1, Y1); store_outputs(ip, 1, Y1, rr1); }")],
}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
