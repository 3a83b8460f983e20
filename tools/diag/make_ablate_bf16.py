#!/usr/bin/env python3
"""Diagnostic only: timing-only ablation builds of conv3x3_bf16.hip's register-weight kernel (results are WRONG by construction):
tools/diag/libkmsr_bf16_<name>.so, loaded with KMSR_LIB_OVERRIDE.  Never part of the product library."""
import os, subprocess, sys
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
F = "conv3x3_bf16.hip"
src = open(f"{R}/{F}").read()
def build(name, edits):
    s = src
    for a, b in edits:
        assert a in s, (name, a[:60])
        s = s.replace(a, b)
    d = f"/tmp/diag_bf16_{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/{F}", "w").write(s)
    for h in ("kmsr_common.h", "kmsr_noise.h"):
        open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
    subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c {F} -o x.o", shell=True)
    objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != F)
    subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_bf16_{name}.so", shell=True)
    print("built", name)
NOMFMA = [("        for (int g = 0; g < NG; ++g) {\n            bf16x8 nb0 = cb0, nb1 = cb1;\n            if (g + 1 < NG) fetch_b(g + 1, nb0, nb1);       // one group ahead",
           "        for (int g = 0; g < (p.H == 12345 ? NG : 1); ++g) {\n            bf16x8 nb0 = cb0, nb1 = cb1;\n            if (g + 1 < NG) fetch_b(g + 1, nb0, nb1);       // one group ahead")]
NODMA = [("        if (more) issue_tile(nxt, buf ^ 1);                 // that buffer's readers", "        if (more && p.H == 12345) issue_tile(nxt, buf ^ 1);                 // that buffer's readers")]
NOSTORE = [("                if (okt) *reinterpret_cast<u32x4*>(out + tile_base + go[t]) = val;", "                if (okt && (p.H == 12345 || val[0] == 0x12345678u)) *reinterpret_cast<u32x4*>(out + tile_base + go[t]) = val;")]
NORES = [("                if (res && more) residual_piece(nxt, nrb, npc, t);", "                if (res && more && p.H == 12345) residual_piece(nxt, nrb, npc, t);")]
ABL = {"nomfma": NOMFMA, "nodma": NODMA, "nostore": NOSTORE, "nores": NORES, "nodma_nostore_nores": NODMA + NOSTORE + NORES}
for n in (sys.argv[1:] or ABL):
    build(n, ABL[n])
