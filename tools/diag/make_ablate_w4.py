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
import re
NOBAR = [("        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();", "        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();")]
NORAW = [(f"issue_dma1({i}); read_patch", f"if (p.relu == 12345) issue_dma1({i}); read_patch") for i in range(3)]
NOVALU = [("        cols_pass(P, t);                                     // 36 packed VALU", "        if (p.relu == 12345) cols_pass(P, t); else asm volatile(\"\" :: \"v\"(P.mid[0]), \"v\"(P.mid[5]), \"v\"(P.e[3]));"),
          ("        rows_pass(t, 0, 3, vn);                              // 21 packed VALU", "        if (p.relu == 12345) rows_pass(t, 0, 3, vn);"),
          ("        rows_pass(t, 3, 6, vn);                              // 21 packed VALU (points 18 .. 35: every MFMA of this step has issued)", "        if (p.relu == 12345) rows_pass(t, 3, 6, vn);")]
NOEPI = [("        epilogue();\n", "        if (p.relu == 12345 || m + 1 == n_my) epilogue();\n")]
# cursors frozen inside the steps (every request repeats the first one of the workgroup: L2 hits -- compare with "rawl2")
NOSALU = [("Q_P(1, 1); raw_next_a(); read_patch", "Q_P(1, 1); read_patch"), ("Q_P(3, 2); raw_next_b(); Q_SB();", "Q_P(3, 2);"),
          ("Q_P(4, 0); raw_next_c(); Q_SB();", "Q_P(4, 0);"), ("Q_P(4, 1); a_lookahead(); Q_SB();", "Q_P(4, 1);")]
RAWL2 = [("        xstep += (long long)QCK * HW;\n        ch_left -= QCK;", "        if (p.relu == 12345) { xstep += (long long)QCK * HW; ch_left -= QCK; }")]   # every raw request re-reads the same rows: L2 hits
NOLDS = [(f"read_patch(P, {k}); Q_SB();", "Q_SB();") for k in range(12)] + [("        Patch P;\n        TPairs t;\n        Q_SB();", "        Patch P = {};\n        TPairs t;\n        Q_SB();")]
NOSTORE = [("                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Y[a]), ro, vo_cur[a] + koff(k), 0, 0);",
            "                    if (p.relu == 12345) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Y[a]), ro, vo_cur[a] + koff(k), 0, 0); else asm volatile(\"\" :: \"v\"(Y[a]));")]
# stores of every item land on the workgroup's FIRST tile: L2-resident lines, no HBM write stream
STOREL2 = [("        if (m + 1 < n_my) set_item(item);", "        if (m + 1 < n_my && p.relu == 12345) set_item(item);")]
# A operands frozen at the prologue's load (no A requests inside the steps)
NOA = [("        A[g % QNA][cbi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, a_lane, step_off + (unsigned)((g * 2 + cbi) * 1024), 0));",
        "        if (gN == 0 || p.relu == 12345) A[g % QNA][cbi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, a_lane, step_off + (unsigned)((g * 2 + cbi) * 1024), 0));")]
# the raw rows are still requested, but no step waits for them (reads whatever is in the ring)
NORAWWAIT = [("        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();", "        q_barrier_lds();")]
# the skip tensor of every item is read from image 0 (16.8 MB: L2 / memory-side cache) instead of its own image
RESL2 = [("            cur_res = res + go;", "            cur_res = res + (p.relu == 12345 ? go : (long long)cg * QCO * HW);")]
ABL = {"resl2": RESL2, "norawwait": NORAWWAIT, "noa": NOA, "nostore": NOSTORE, "storel2": STOREL2, "nobar": NOBAR, "noraw": NORAW, "novalu": NOVALU, "noepi": NOEPI, "nosalu": NOSALU, "rawl2": RAWL2, "nolds_novalu": NOLDS + NOVALU,
       "rawl2_noepi": RAWL2 + NOEPI, "novalu_noepi": NOVALU + NOEPI,
       "mfmaonly": NOBAR + NORAW + NOVALU + NOEPI + NOSALU + NOLDS}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
