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
NOA = [("        for (int cbi = 0; cbi < 2; ++cbi)\n            A[g % QNA][cbi] =", "        for (int cbi = 0; cbi < 2 && (gN == 0 || p.relu == 12345); ++cbi)\n            A[g % QNA][cbi] =")]
NOBAR = [("        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();", "        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();")]
NORAW = [("        issue_raw();                                         // raw(g+3)", "        if (p.relu == 12345) issue_raw();")]
NOVALU = [("        cols_pass(P, t);                                     // 36 packed VALU", "        if (p.relu == 12345) cols_pass(P, t); else asm volatile(\"\" :: \"v\"(P.mid[0]), \"v\"(P.mid[5]), \"v\"(P.e[3]));"),
          ("        rows_pass(t, 0, 3, vn);                              // 21 packed VALU", "        if (p.relu == 12345) rows_pass(t, 0, 3, vn);"),
          ("        rows_pass(t, 3, 6, vn);                              // 21 packed VALU (points 18 .. 35: every MFMA of this step has issued)", "        if (p.relu == 12345) rows_pass(t, 3, 6, vn);")]
NOEPI = [("        epilogue();\n", "        if (p.relu == 12345 || m + 1 == n_my) epilogue();\n")]
NODRAIN = [("            if (have_prev && s_in_item < QSTG_SUB) drain(s_in_item);", "            if (p.relu == 12345 && have_prev && s_in_item < QSTG_SUB) drain(s_in_item);")]
RAWL2 = [("        xstep += (long long)QCK * HW;\n        ch_left -= QCK;", "        if (p.relu == 12345) { xstep += (long long)QCK * HW; ch_left -= QCK; }")]   # every raw request re-reads the same rows: L2 hits
ABL = {"noa": NOA, "nobar": NOBAR, "noraw": NORAW, "novalu": NOVALU, "noepi": NOEPI, "nodrain": NODRAIN, "rawl2": RAWL2,
       "mfmaonly": NOA + NOBAR + NORAW + NOVALU + NOEPI + NODRAIN, "noa_noraw": NOA + NORAW, "novalu_noepi": NOVALU + NOEPI}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
