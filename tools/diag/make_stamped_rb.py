#!/usr/bin/env python3
"""Diagnostic only: resblock_bf16.hip with cycle stamps per phase of a tile; sums per wave -> __device__ array read by kmsr_diag_rb."""
import os, subprocess
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
s = open(f"{R}/resblock_bf16.hip").read()
def rep(a, b, count=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, count)
rep("namespace kmsr {\nnamespace {\n", "namespace kmsr {\n__device__ unsigned long long g_rb_diag[256 * 4 * 16];\nnamespace {\n")
ST = lambda i: "{ const unsigned long long T_ = __builtin_readcyclecounter(); d_ph[%d] += T_ - d_last; d_last = T_; }" % i
rep("    int buf = 0;\n    Origin cur = tile_origin(item);", "    unsigned long long d_ph[12] = {}, d_t0 = __builtin_readcyclecounter(), d_last = d_t0; int d_n = 0;\n    int buf = 0;\n    Origin cur = tile_origin(item);")
rep("        const bool more = item + 1 < item_end;", "        " + ST(9) + " ++d_n;\n        const bool more = item + 1 < item_end;")
rep("        if (more) issue_tile(nxt, buf ^ 1);\n        const unsigned short* il = smem + buf * RIN;", "        " + ST(0) + "\n        if (more) issue_tile(nxt, buf ^ 1);\n        const unsigned short* il = smem + buf * RIN;\n        " + ST(1))
rep("            // epilogue 1: t = bf16(relu(acc + b1))", "            " + ST(2) + "\n            // epilogue 1: t = bf16(relu(acc + b1))")
rep("        __builtin_amdgcn_s_barrier();                        // B: t complete", "        " + ST(3) + "\n        __builtin_amdgcn_s_barrier();                        // B: t complete\n        " + ST(4))
rep("            // epilogue 2: c = bf16(acc + b2)", "            " + ST(5) + "\n            // epilogue 2: c = bf16(acc + b2)")
rep("        __builtin_amdgcn_s_barrier();                        // C: staging complete", "        " + ST(6) + "\n        __builtin_amdgcn_s_barrier();                        // C: staging complete")
rep("        __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0): the next tile", "        __builtin_amdgcn_s_waitcnt(0x0070);\n        " + ST(7) + " //")
rep("        cur = nxt;\n    }", "        " + ST(8) + "\n        cur = nxt;\n    }\n    if (lane == 0) { unsigned long long* o = g_rb_diag + (blockIdx.x * 4 + wv) * 16; for (int i = 0; i < 10; ++i) o[i] = d_ph[i]; o[10] = __builtin_readcyclecounter() - d_t0; o[11] = d_n; }")
s += '\nextern "C" int kmsr_diag_rb(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kmsr::g_rb_diag), sizeof(unsigned long long) * 256 * 4 * 16); }\n'
d = "/tmp/diag_rb_stamped"
os.makedirs(d, exist_ok=True)
open(f"{d}/resblock_bf16.hip", "w").write(s)
for h in ("kmsr_common.h", "kmsr_noise.h"):
    open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c resblock_bf16.hip -o x.o", shell=True)
objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "resblock_bf16.hip")
subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_rb_stamped.so", shell=True)
print("built stamped")
