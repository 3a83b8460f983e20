#!/usr/bin/env python3
"""Diagnostic only: timing-only ablation builds of conv3x3_wino4.hip (results are WRONG by construction):
tools/diag/libkmsr_abl_w4_<name>.so, loaded with KMSR_LIB_OVERRIDE.  Never part of the product library."""
import os, subprocess, sys
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
src = open(f"{R}/conv3x3_wino4.hip").read()
def build(name, edits):
    s = src
    for a, b in edits:
        assert a in s, (name, a[:60])
        s = s.replace(a, b)
    d = f"/tmp/diag_w4_{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/conv3x3_wino4.hip", "w").write(s)
    for h in ("kmsr_common.h", "kmsr_noise.h"):
        open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
    subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wino4.hip -o conv3x3_wino4.o", shell=True)
    objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wino4.hip")
    subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/conv3x3_wino4.o -o /root/repo/tools/diag/libkmsr_abl_w4_{name}.so", shell=True)
    print("built", name)
NOA = [("    auto load_a = [&](unsigned step_off, int g) {           // group g of the step at step_off -> slot g % QNA\n#pragma unroll\n        for (int cbi = 0; cbi < 2; ++cbi)",
        "    auto load_a = [&](unsigned step_off, int g) {\n#pragma unroll\n        for (int cbi = 0; cbi < 2 && (gA == 0 || p.relu == 12345); ++cbi)")]
NOBAR = [("        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();                                     // raw(g+1) visible to every wave; raw(g)'s slot free",
          "        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();")]
NORAW = [("        issue_raw();                                         // raw(g+3)", "        if (p.relu == 12345) issue_raw();")]
NOVALU = [("        cols_pass(P, t);                                     // 36 packed VALU", "        if (p.relu == 12345) cols_pass(P, t);"),
          ("        rows_pass(t, 0, 3, vn);                              // 21 packed VALU", "        if (p.relu == 12345) rows_pass(t, 0, 3, vn);"),
          ("        rows_pass(t, 3, 6, vn);                              // 21 packed VALU", "        if (p.relu == 12345) rows_pass(t, 3, 6, vn); else {\n#pragma unroll\n for (int i = 0; i < QPTS; ++i) vn[i] = vc[i] + P.mid[i % 6].x + P.e[i % 6].y; }")]
NOEPI = [("        epilogue(item);\n        item += istep;", "        if (p.relu == 12345 || m + 1 == n_my) epilogue(item);\n        item += istep;")]
ABL = {"noa": NOA, "nobar": NOBAR, "noraw": NORAW, "novalu": NOVALU, "noepi": NOEPI, "mfmaonly": NOA + NOBAR + NORAW + NOVALU + NOEPI,
       "noa_nobar": NOA + NOBAR, "novalu_noepi": NOVALU + NOEPI}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
