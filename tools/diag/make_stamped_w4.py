#!/usr/bin/env python3
"""Diagnostic only: conv3x3_wino4.hip with s_memtime stamps at three points per step (before the step's wait + barrier, after the
barrier, at the end of the step) and around the epilogue; per-wave sums land in a __device__ array read back by kmsr_diag_w4.
Builds tools/diag/libkmsr_w4_stamped.so (load with KMSR_LIB_OVERRIDE; run tools/diag/stamp_w4.py)."""
import os, subprocess
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
s = open(f"{R}/conv3x3_wino4.hip").read()
def rep(a, b):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, 1)
rep("struct Wino4Params {", "__device__ unsigned long long g_w4_diag[256 * 4 * 8];\nstruct Wino4Params {")
rep("    f32x4 acc[2][QPTS];", "    unsigned long long d_bar = 0, d_step = 0, d_epi = 0, d_first = 0, d_t0 = __builtin_readcyclecounter();\n    f32x4 acc[2][QPTS];")
rep("        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();",
    "        const unsigned long long T0 = __builtin_readcyclecounter();\n        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();\n        const unsigned long long T1 = __builtin_readcyclecounter();\n        d_bar += T1 - T0;")
rep("        advance_a();\n    };", "        advance_a();\n        const unsigned long long T2 = __builtin_readcyclecounter();\n        if (FIRST) d_first += T2 - T1; else d_step += T2 - T1;\n    };")
rep("        epilogue();\n", "        { const unsigned long long E0 = __builtin_readcyclecounter(); epilogue(); d_epi += __builtin_readcyclecounter() - E0; }\n")
rep("#undef Q_MFMA8\n}", "    if (lane == 0) { unsigned long long* o = g_w4_diag + (blockIdx.x * 4 + wv) * 8; o[0] = d_bar; o[1] = d_step; o[2] = d_epi; o[3] = d_first; o[4] = __builtin_readcyclecounter() - d_t0; o[5] = n_my; }\n#undef Q_MFMA8\n}")
s += '\nextern "C" int kmsr_diag_w4(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kmsr::g_w4_diag), sizeof(unsigned long long) * 256 * 4 * 8); }\n'
s = s.replace("namespace kmsr {\nnamespace {\n", "namespace kmsr {\n__device__ unsigned long long g_w4_diag[256 * 4 * 8];\nnamespace {\n", 1).replace("__device__ unsigned long long g_w4_diag[256 * 4 * 8];\nstruct Wino4Params {", "struct Wino4Params {", 1)
d = "/tmp/diag_w4_stamped"
os.makedirs(d, exist_ok=True)
open(f"{d}/conv3x3_wino4.hip", "w").write(s)
for h in ("kmsr_common.h", "kmsr_noise.h"):
    open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wino4.hip -o conv3x3_wino4.o", shell=True)
objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wino4.hip")
subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/conv3x3_wino4.o -o /root/repo/tools/diag/libkmsr_w4_stamped.so", shell=True)
print("built stamped")
