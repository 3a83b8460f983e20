#!/usr/bin/env python3
"""Diagnostic only: timing-only ablation builds of conv3x3_wgrad_wino.hip (results are WRONG by construction):
tools/diag/libkmsr_wgw_<name>.so.  Never part of the product library."""
import os, subprocess, sys
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
F = "conv3x3_wgrad_wino.hip"
src = open(f"{R}/{F}").read()
def build(name, edits):
    s = src
    for a, b in edits:
        assert a in s, (name, a[:60])
        s = s.replace(a, b)
    d = f"/tmp/diag_wgw_{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/{F}", "w").write(s)
    for h in ("kmsr_common.h", "kmsr_noise.h"):
        open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
    subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c {F} -o x.o", shell=True)
    objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != F)
    subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_wgw_{name}.so", shell=True)
    print("built", name)
NOVALU = [("                    R[0][c] = e0 - e2;\n                    R[1][c] = e1 + e2;\n                    R[2][c] = e2 - e1;\n                    R[3][c] = e1 - e3;",
           "                    R[0][c] = e0; R[1][c] = e1; R[2][c] = e2; R[3][c] = e3;"),
          ("                    V[4 * i + 0] = R[i][0] - R[i][2];\n                    V[4 * i + 1] = R[i][1] + R[i][2];\n                    V[4 * i + 2] = R[i][2] - R[i][1];\n                    V[4 * i + 3] = R[i][1] - R[i][3];",
           "                    V[4 * i + 0] = R[i][0]; V[4 * i + 1] = R[i][1]; V[4 * i + 2] = R[i][2]; V[4 * i + 3] = R[i][3];"),
          ("                const float Tp[4] = {g0.x, g0.x + g1.x, g0.x - g1.x, g1.x};\n                const float Tq[4] = {g0.y, g0.y + g1.y, g0.y - g1.y, g1.y};",
           "                const float Tp[4] = {g0.x, g0.x, g1.x, g1.x};\n                const float Tq[4] = {g0.y, g0.y, g1.y, g1.y};"),
          ("                    A[4 * i + 1] = Tp[i] + Tq[i];\n                    A[4 * i + 2] = Tp[i] - Tq[i];", "                    A[4 * i + 1] = Tp[i];\n                    A[4 * i + 2] = Tq[i];")]
NOSTAGE = [("        if (more) load_stage(s + p.nblk);                    // in flight", "        if (more && p.N == 12345) load_stage(s + p.nblk);                    // in flight"),
           ("        if (more) store_stage(b ^ 1);  ", "        if (more && p.N == 12345) store_stage(b ^ 1);  ")]
NOBAR = NOSTAGE + [("        if (more && p.N == 12345) store_stage(b ^ 1);                        // the buffer every wave finished reading before the last barrier\n        __syncthreads();",
                    "        if (more && p.N == 12345) store_stage(b ^ 1);\n")]
NOLOAD = [NOSTAGE[0]]
NOSTORE = [("        if (more) store_stage(b ^ 1);  ", "        if (more) { if (p.N == 12345) store_stage(b ^ 1); else { float t = xh; for (int q = 0; q < 16; ++q) t += xr[q]; for (int q = 0; q < 8; ++q) t += dr[q]; asm volatile(\"\" :: \"v\"(t)); } }  ")]
ABL = {"noload": NOLOAD, "nostore": NOSTORE, "novalu": NOVALU, "nostage": NOSTAGE, "nobar": None, "novalu_nostage": NOVALU + NOSTAGE}
for n in (sys.argv[1:] or ABL):
    if n == "nobar":
        s2 = src
        for a, b in NOSTAGE: s2 = s2.replace(a, b)
        i = s2.index("        if (more && p.N == 12345) store_stage(b ^ 1);")
        j = s2.index("        __syncthreads();", i)
        src_save = src; src = s2[:j] + s2[j + len("        __syncthreads();"):]
        build(n, []); src = src_save
    else:
        build(n, ABL[n])
