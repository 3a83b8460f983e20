#!/usr/bin/env python3
"""Diagnostic only: writes a stamped copy of conv3x3_wino_vreg.hip to /tmp/diag and links tools/diag/libkmsr_stamp.so
(s_memtime stamps per chunk section + a debug-buffer hook).  The product library never contains stamps."""
import os, subprocess
R = "/root/repo/kernel-modeling-super-resolution_amd/csrc"
s = open(f"{R}/conv3x3_wino_vreg.hip").read()
def rep(a, b, n=1):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, n)
rep('__device__ __attribute__((aligned(16))) float g_zero_page[4];', '''__device__ __attribute__((aligned(16))) float g_zero_page[4];
__device__ unsigned long long* g_dbg;
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (lane == 0 && dbg && (k) < 256) dbg[(k)] = t_; } while (0)
''')
rep('    const int hh = wv & 1, tq = wv >> 1,', '''    unsigned long long* dbg = g_dbg ? g_dbg + ((long long)blockIdx.x * 8 + wv) * 256 : nullptr;
    STAMP(0);
    const int hh = wv & 1, tq = wv >> 1,''')
rep('''    wait_dma<3 * RINSTR + 2 * UINSTR>();                     // raw(0) has landed (this wave's part)
    barrier_lds();''', '''    wait_dma<3 * RINSTR + 2 * UINSTR>();                     // raw(0) has landed (this wave's part)
    STAMP(1);
    barrier_lds();
    STAMP(2);''')
rep('''        step(g, 0, vA, vB, g, 1, g, 1, false);
        wait_dma<RINSTR>();''', '''        STAMP(8 + 6 * g);
        step(g, 0, vA, vB, g, 1, g, 1, false);
        STAMP(9 + 6 * g);
        wait_dma<RINSTR>();''')
rep('''        barrier_lds();
        f32x2 rr0[2][4];''', '''        STAMP(10 + 6 * g);
        barrier_lds();
        STAMP(11 + 6 * g);
        f32x2 rr0[2][4];''')
rep('''        step(g, 1, vB, vA, g + 1, 0, g + 1, 0, true);
        if (++c == nch) {''', '''        STAMP(12 + 6 * g);
        step(g, 1, vB, vA, g + 1, 0, g + 1, 0, true);
        STAMP(13 + 6 * g);
        if (++c == nch) {''')
rep('''            ++item;
        }
    }
#undef KMSR_MFMA8''', '''            ++item;
        }
    }
    STAMP(7);
#undef KMSR_MFMA8''')
rep('extern "C" size_t kmsr_conv3x3_winov_packed_weight_floats', 'extern "C" int kmsr_debug_set_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &p, sizeof(p)); }\nextern "C" size_t kmsr_conv3x3_winov_packed_weight_floats')
os.makedirs("/tmp/diag", exist_ok=True)
open("/tmp/diag/conv3x3_wino_vreg.hip", "w").write(s)
for h in ("kmsr_common.h", "kmsr_noise.h"):
    t = open(f"{R}/{h}").read().replace('"../../include/kmsr.h"', '"/root/repo/include/kmsr.h"')
    open(f"/tmp/diag/{h}", "w").write(t)
subprocess.check_call("cd /tmp/diag && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wino_vreg.hip -o conv3x3_wino_vreg.o", shell=True)
objs = " ".join(f"{R}/{o[:-4]}.o" for o in sorted(os.listdir(R)) if o.endswith(".hip") and o != "conv3x3_wino_vreg.hip")
subprocess.check_call(f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 {objs} /tmp/diag/conv3x3_wino_vreg.o -o /root/repo/tools/diag/libkmsr_stamp.so", shell=True)
print("built tools/diag/libkmsr_stamp.so")
