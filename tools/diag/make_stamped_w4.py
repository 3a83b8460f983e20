#!/usr/bin/env python3
"""Diagnostic only: conv3x3_wino4.hip with s_memtime stamps inside the step (9 phases) and around the epilogue; per-wave sums land in a
__device__ array read back by kmsr_diag_w4.  Builds tools/diag/libkmsr_w4_stamped.so (load with KMSR_LIB_OVERRIDE; run tools/diag/stamp_w4.py).
Each stamp waits for lgkmcnt(0) (s_memtime returns through the scalar cache), so LDS latencies are charged to the phase that ends at the stamp."""
import os, subprocess
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
s = open(f"{R}/conv3x3_wino4.hip").read()
def rep(a, b, count=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, count)
NPH = 8
rep("namespace kmsr {\nnamespace {\n", "namespace kmsr {\n__device__ unsigned long long g_w4_diag[256 * 4 * 16];\nnamespace {\n")
rep("    f32x4 acc[2][QPTS];", "    unsigned long long d_ph[%d] = {}, d_epi = 0, d_t0 = __builtin_readcyclecounter(), d_last = 0;\n    f32x4 acc[2][QPTS];" % NPH)
ST = lambda i: "{ const unsigned long long T_ = __builtin_readcyclecounter(); d_ph[%d] += T_ - d_last; d_last = T_; }" % i
rep("        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();", "        d_last = __builtin_readcyclecounter();\n        q_wait_vm<2 * QGROUPS * 2 + QRINSTR>();\n        q_barrier_lds();\n        " + ST(0))
rep("        cols_pass(P, t);                                     // 36 packed VALU\n        Q_SB();", ST(1) + " Q_SB();\n        cols_pass(P, t);\n        Q_SB();\n        " + ST(2) + " Q_SB();")
rep("        rows_pass(t, 0, 3, vn);                              // 21 packed VALU\n        Q_SB();", ST(3) + " Q_SB();\n        rows_pass(t, 0, 3, vn);\n        Q_SB();\n        " + ST(4) + " Q_SB();")
rep("        rows_pass(t, 3, 6, vn);                              // 21 packed VALU (points 18 .. 35: every MFMA of this step has issued)\n        Q_SB();\n        a_cur = a_nxt;\n        a_nxt = a_nn;",
    ST(5) + " Q_SB();\n        rows_pass(t, 3, 6, vn);\n        Q_SB();\n        " + ST(6) + "\n        a_cur = a_nxt;\n        a_nxt = a_nn;\n        " + ST(7))
rep("        epilogue();\n", "        { const unsigned long long E0 = __builtin_readcyclecounter(); epilogue(); d_epi += __builtin_readcyclecounter() - E0; }\n")
rep("#undef Q_MFMA2\n}", "    if (lane == 0) { unsigned long long* o = g_w4_diag + (blockIdx.x * 4 + wv) * 16; for (int i = 0; i < %d; ++i) o[i] = d_ph[i]; o[9] = d_epi; o[10] = __builtin_readcyclecounter() - d_t0; o[11] = n_my; }\n#undef Q_MFMA2\n}" % NPH)
s += '\nextern "C" int kmsr_diag_w4(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kmsr::g_w4_diag), sizeof(unsigned long long) * 256 * 4 * 16); }\n'
d = "/tmp/diag_w4_stamped"
os.makedirs(d, exist_ok=True)
open(f"{d}/conv3x3_wino4.hip", "w").write(s)
for h in ("kmsr_common.h", "kmsr_noise.h"):
    open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wino4.hip -o conv3x3_wino4.o", shell=True)
objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wino4.hip")
subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/conv3x3_wino4.o -o /root/repo/tools/diag/libkmsr_w4_stamped.so", shell=True)
print("built stamped")
