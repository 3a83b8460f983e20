#!/usr/bin/env python3
"""Diagnostic only: timing-only ablation builds of conv3x3_wgrad_wino4.hip (results are WRONG by construction):
tools/diag/libkmsr_abl_wg4_<name>.so, loaded with KMSR_LIB_OVERRIDE.  Never part of the product library."""
import os, subprocess, sys
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
src = open(f"{R}/conv3x3_wgrad_wino4.hip").read()
def build(name, edits):
    s = src
    for a, b in edits:
        assert a in s, (name, a[:70])
        s = s.replace(a, b)
    d = f"/tmp/diag_wg4_{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/conv3x3_wgrad_wino4.hip", "w").write(s)
    for h in ("kmsr_common.h", "kmsr_noise.h", "kmsr_wino4_xf.h"):
        open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
    subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wgrad_wino4.hip -o x.o", shell=True)
    objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wgrad_wino4.hip")
    subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_abl_wg4_{name}.so", shell=True)
    print("built", name)
NODMA = [("        if (di >= 0) dma_issue(di, nb); ", "        if (di >= 0 && p.N == 12345) dma_issue(di, nb); ")]
# every request re-reads the workgroup's first stage: L2 hits instead of the HBM stream
DMAL2 = [("        if (req_on) dma_setup();", "        if (req_on && p.N == 12345) dma_setup();")]
NOMFMA = [('asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc', 'if (p.N == 12345) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc'),
          ('asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc', 'if (p.N == 12345) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc')]
FAKEV = "{ for (int k = 0; k < HPW; ++k) %s[k] = %s; }"
NOVALU = [("            vhat(X0, Vop[0]);\n            ahat(D0, Aop[0]);", "            if (p.N == 12345) { vhat(X0, Vop[0]); ahat(D0, Aop[0]); } else { for (int k = 0; k < HPW; ++k) { Vop[0][k] = X0.mid[k % 5].x + X0.e[k % 5].y; Aop[0][k] = D0.g[k & 3].x; } }"),
          ("            vhat(X1, Vop[1]);", "            if (p.N == 12345) vhat(X1, Vop[1]); else { for (int k = 0; k < HPW; ++k) Vop[1][k] = X1.mid[k % 5].x + X1.e[k % 5].y; }"),
          ("            ahat(D1, Aop[1]);", "            if (p.N == 12345) ahat(D1, Aop[1]); else { for (int k = 0; k < HPW; ++k) Aop[1][k] = D1.g[k & 3].x; }")]
NOBAR = [("        __builtin_amdgcn_s_waitcnt(h_wait_imm(0, 0));       // next stage landed (requested a whole stage ago); my reads of this one are done\n        __builtin_amdgcn_s_barrier();",
          "        if (p.N == 12345) { __builtin_amdgcn_s_waitcnt(h_wait_imm(0, 0)); __builtin_amdgcn_s_barrier(); }")]
NOLDS = [("        if (k & 1) P.e[k >> 1] = f32x2{rp[0], rp[5]};\n        else P.mid[k >> 1] = *reinterpret_cast<const f32x4*>(rp + 1);", "        if (p.N == 12345) { if (k & 1) P.e[k >> 1] = f32x2{rp[0], rp[5]}; else P.mid[k >> 1] = *reinterpret_cast<const f32x4*>(rp + 1); } else { if (k & 1) P.e[k >> 1] = f32x2{1.f, 2.f}; else P.mid[k >> 1] = f32x4{1.f, 2.f, 3.f, 4.f}; }"),
         ("        D.g[r] = *reinterpret_cast<const f32x4*>(st + dbase + a * HDCH + 16 * ks + r * HDROW);", "        if (p.N == 12345) D.g[r] = *reinterpret_cast<const f32x4*>(st + dbase + a * HDCH + 16 * ks + r * HDROW); else D.g[r] = f32x4{1.f, 2.f, 3.f, 4.f};")]
NOFLUSH = [("    flush(acc, first_flush);", "    if (p.N == 12345 || s1 == s_end) flush(acc, first_flush);")]
DMAHALF = [("        if (di >= 0) dma_issue(di, nb); ", "        if (di >= 0 && (di & 1) == 0) dma_issue(di, nb); ")]
def lag(n):   # the second channel group of a stream starts n x 64 cycles late: its dy requests become true L2 hits instead of hits-on-miss
    return [("    // ---- prologue: stage s_beg -> buffer 0", "    if (grp & 1) __builtin_amdgcn_s_sleep(%d);\n    // ---- prologue: stage s_beg -> buffer 0" % n)]
ABL = {"lag32": lag(32), "lag64": lag(64), "lag127": lag(127), "dmahalf": DMAHALF, "nodma": NODMA, "dmal2": DMAL2, "nomfma": NOMFMA, "novalu": NOVALU, "nobar": NOBAR, "nolds": NOLDS, "noflush": NOFLUSH,
       "mfmaonly": NODMA + NOVALU + NOBAR + NOLDS, "nodma_nolds": NODMA + NOLDS, "nolds_novalu": NOLDS + NOVALU, "nodma_novalu": NODMA + NOVALU}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
