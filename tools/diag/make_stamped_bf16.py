#!/usr/bin/env python3
"""Diagnostic only: cycle-stamped build of conv3x3_bf16.hip's register-weight kernel (s_memtime at the phase boundaries of wave 0 of
workgroup 0, items 2..9) -> tools/diag/libkmsr_bf16_stamped.so with an extra export kmsr_diag_stamps(unsigned long long* host, int n)."""
import os, subprocess
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
F = "conv3x3_bf16.hip"
s = open(f"{R}/{F}").read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a[:70], s.count(a))
    s = s.replace(a, b)
STAMP = lambda k: f'if (blockIdx.x == 0 && tid == 0 && item - item0 >= 2 && item - item0 < 10) g_stamps[(item - item0 - 2) * 8 + {k}] = __builtin_amdgcn_s_memtime();\n'
rep("struct Origin { int n, ty0, tx0, cg; };", "__device__ unsigned long long g_stamps[64];\nstruct Origin { int n, ty0, tx0, cg; };")
rep("    Origin cur = tile_origin(item), nxt_carry = cur;", "    const int item0 = item;\n    Origin cur = tile_origin(item), nxt_carry = cur;")
rep("        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        if (more) issue_tile(nxt, buf ^ 1);",
    "        " + STAMP(0) + "        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        " + STAMP(1) + "        if (more) issue_tile(nxt, buf ^ 1);\n        " + STAMP(2))
a = s.index("template <bool PS, int KS, bool RELU, int MT, int NW>")
k = s[a:]
k = k.replace("        // ---- epilogue into the staging tile (its own LDS region", "        " + STAMP(3) + "        // ---- epilogue into the staging tile (its own LDS region", 1)
k = k.replace("        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): staging complete\n        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        __builtin_amdgcn_s_waitcnt(0x0070);",
              "        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): staging complete\n        " + STAMP(4) + "        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        " + STAMP(5) + "        __builtin_amdgcn_s_waitcnt(0x0070);\n        " + STAMP(6), 1)
pat = "                if (okt) *reinterpret_cast<u32x4*>(out + tile_base + go[t]) = val[t];\n            }\n        }\n"
assert k.count(pat) == 1
k = k.replace(pat, pat + "        " + STAMP(7), 1)
s = s[:a] + k
s += '''
extern "C" int kmsr_diag_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kmsr::g_stamps_alias), sizeof(unsigned long long) * n);
}
'''
# expose the symbol outside the anonymous namespace
s = s.replace("__device__ unsigned long long g_stamps[64];", "}\n__device__ unsigned long long g_stamps_alias[64];\nnamespace {\n#define g_stamps g_stamps_alias")
d = "/tmp/diag_bf16_stamped"
os.makedirs(d, exist_ok=True)
open(f"{d}/{F}", "w").write(s)
for h in ("kmsr_common.h", "kmsr_noise.h"):
    open(f"{d}/{h}", "w").write(open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"'))
subprocess.check_call(f"cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c {F} -o x.o", shell=True)
objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != F)
subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} {d}/x.o -o /root/repo/tools/diag/libkmsr_bf16_stamped.so", shell=True)
print("built stamped")
