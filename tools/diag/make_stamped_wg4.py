#!/usr/bin/env python3
"""Diagnostic only: conv3x3_wgrad_wino4.hip with cycle stamps per phase of a K-step; per-wave sums land in a __device__ array read back
by kmsr_diag_wg4.  Builds tools/diag/libkmsr_wg4_stamped.so (KMSR_LIB_OVERRIDE; run tools/diag/stamp_wg4.py)."""
import os, subprocess
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
s = open(f"{R}/conv3x3_wgrad_wino4.hip").read()
def rep(a, b, count=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, count)
NPH = 10
rep("namespace kmsr {\nnamespace {\n", "namespace kmsr {\n__device__ unsigned long long g_wg4_diag[512 * 4 * 16];\nnamespace {\n")
rep("    bool first_flush = true;", "    unsigned long long d_ph[%d] = {}, d_t0 = __builtin_readcyclecounter(), d_last = d_t0; int d_stages = 0;\n    bool first_flush = true;" % NPH)
ST = lambda i: "{ const unsigned long long T_ = __builtin_readcyclecounter(); d_ph[%s] += T_ - d_last; d_last = T_; }" % i
# stage top (after previous barrier): phase 9 = barrier wait of the previous stage
rep("        const bool req_on = rs < s_end;", "        " + ST(9) + " ++d_stages;\n        const bool req_on = rs < s_end;")
rep("            ahat(D0, Aop[0]);\n            H_SB();", "            ahat(D0, Aop[0]);\n            H_SB();\n            " + ST(0) + "\n            H_SB();")
rep("            vhat(X1, Vop[1]);\n            H_SB();", "            " + ST(1) + " H_SB();\n            vhat(X1, Vop[1]);\n            H_SB();\n            " + ST(2) + " H_SB();")
rep("            ahat(D1, Aop[1]);\n            H_SB();", "            " + ST(3) + " H_SB();\n            ahat(D1, Aop[1]);\n            H_SB();\n            " + ST(4) + " H_SB();")
rep("#undef H_G\n#undef H_DMA\n        }", "            " + ST(5) + "\n#undef H_G\n#undef H_DMA\n        }")
rep("        if (req_on) dma_advance();", "        " + ST(7) + "\n        if (req_on) dma_advance();")
rep("    flush(acc, first_flush);", "    " + ST(7) + "\n    flush(acc, first_flush);\n    " + ST(8))
rep("    if (s_beg >= s_end) {                                    // an idle workgroup", "    if (lane == 0) { unsigned long long* o = g_wg4_diag + ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16; for (int i = 0; i < %d; ++i) o[i] = d_ph[i]; o[10] = __builtin_readcyclecounter() - d_t0; o[11] = d_stages; }\n    if (s_beg >= s_end) {                                    // an idle workgroup" % NPH)
s += '\nextern "C" int kmsr_diag_wg4(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kmsr::g_wg4_diag), sizeof(unsigned long long) * 512 * 4 * 16); }\n'
d = "/tmp/diag_wg4_stamped"
os.makedirs(d, exist_ok=True)
open(f"{d}/conv3x3_wgrad_wino4.hip", "w").write(s)
for h in ("kmsr_common.h", "kmsr_noise.h", "kmsr_wino4_xf.h"):
    open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wgrad_wino4.hip -o x.o", shell=True)
objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wgrad_wino4.hip")
subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_wg4_stamped.so", shell=True)
print("built stamped")
