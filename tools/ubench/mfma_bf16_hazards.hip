// Microbenchmark (round 4): does v_mfma_f32_32x32x16_bf16 on gfx950 need software wait states
//   (a) between the MFMA and a VALU WRITE to its SrcB registers (write-after-read), and
//   (b) between the MFMA and a VALU READ of its first / last accumulator register,
//   (c) the same read behind v_mfma_f32_16x16x4_f32,  (d)/(e) between a VALU write of SrcC / SrcB and the MFMA that reads it,
// beyond what the instruction stream provides?  The whole probe is ONE inline-asm block on fixed physical registers (B = v[100:103],
// accumulator = v[104:119]), so neither hipcc's scheduler nor its hazard recognizer touches it.
// Per round:  acc = 0;  CHAIN dependent MFMAs acc += A x B;  [s_nop GAP];  probe;  drain (96 cycles of s_nop);  compare with a fully drained reference.
//   (a) probe = v_mov_b32 v100..v103, NaN pattern  -> the accumulator must still equal the reference
//   (b) probe = v_mov_b32 t0, v104 ; t15, v119     -> must equal the drained accumulator
// One wave per SIMD (256 threads, 1 block per CU), all 4 SIMDs of every CU running the same code.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_bf16_hazards.hip -o mfma_bf16_hazards
#include <hip/hip_runtime.h>
#include <cstdio>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define DRAIN "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define MFMA "v_mfma_f32_32x32x16_bf16 v[104:119], %[a], v[100:103], v[104:119]\n\t"
#define SETB "v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[b1]\n\tv_mov_b32 v102, %[b2]\n\tv_mov_b32 v103, %[b3]\n\t"
#define ZACC "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t" \
             "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\t"
#define KILLB "v_mov_b32 v100, 0x7fc07fc0\n\tv_mov_b32 v101, 0x7fc07fc0\n\tv_mov_b32 v102, 0x7fc07fc0\n\tv_mov_b32 v103, 0x7fc07fc0\n\t"
#define READ "v_mov_b32 %[t0], v104\n\tv_mov_b32 %[t15], v119\n\t"
#define OUT "v_mov_b32 %[o0], v104\n\tv_mov_b32 %[o1], v105\n\tv_mov_b32 %[o2], v111\n\tv_mov_b32 %[o3], v119\n\t"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151"

// one probe: returns 4 sampled accumulator registers (0, 1, 7, 15) after the drain, and the two early reads
#define PROBE(NAME, CHAINSTR, GAPSTR, PROBESTR)                                                                                  \
    __device__ __forceinline__ void NAME(u32x4 a, u32x4 b, float (&o)[4], float& t0, float& t15) {                                \
        asm volatile(SETB ZACC "s_nop 7\n\t" CHAINSTR GAPSTR PROBESTR DRAIN OUT                                                  \
                     : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [t0] "=&v"(t0), [t15] "=&v"(t15)   \
                     : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3])                                 \
                     : CLOB, "memory");                                                                                           \
    }
#define NOREAD "v_mov_b32 %[t0], 0\n\tv_mov_b32 %[t15], 0\n\t"
PROBE(ref1, MFMA, DRAIN, NOREAD)
PROBE(ref3, MFMA MFMA MFMA, DRAIN, NOREAD)
PROBE(a1_none, MFMA, "", KILLB NOREAD)
PROBE(a1_n0, MFMA, "s_nop 0\n\t", KILLB NOREAD)
PROBE(a1_n1, MFMA, "s_nop 1\n\t", KILLB NOREAD)
PROBE(a1_n3, MFMA, "s_nop 3\n\t", KILLB NOREAD)
PROBE(a1_n7, MFMA, "s_nop 7\n\t", KILLB NOREAD)
PROBE(a3_none, MFMA MFMA MFMA, "", KILLB NOREAD)
PROBE(b1_none, MFMA, "", READ)
PROBE(b1_n1, MFMA, "s_nop 1\n\t", READ)
PROBE(b1_n3, MFMA, "s_nop 3\n\t", READ)
PROBE(b1_n7, MFMA, "s_nop 7\n\t", READ)
PROBE(b1_n15, MFMA, "s_nop 15\n\t", READ)
PROBE(b1_n31, MFMA, "s_nop 15\n\ts_nop 15\n\t", READ)
PROBE(b1_n9, MFMA, "s_nop 9\n\t", READ)
PROBE(b1_n10, MFMA, "s_nop 10\n\t", READ)
PROBE(b1_n11, MFMA, "s_nop 11\n\t", READ)
PROBE(b4i_n10, "v_mfma_f32_32x32x16_bf16 v[120:135], %[a], v[100:103], v[120:135]\n\tv_mfma_f32_32x32x16_bf16 v[136:151], %[a], v[100:103], v[136:151]\n\t" MFMA, "s_nop 10\n\t", READ)
PROBE(b4i_n11, "v_mfma_f32_32x32x16_bf16 v[120:135], %[a], v[100:103], v[120:135]\n\tv_mfma_f32_32x32x16_bf16 v[136:151], %[a], v[100:103], v[136:151]\n\t" MFMA, "s_nop 11\n\t", READ)
PROBE(b4i_n7, "v_mfma_f32_32x32x16_bf16 v[120:135], %[a], v[100:103], v[120:135]\n\tv_mfma_f32_32x32x16_bf16 v[136:151], %[a], v[100:103], v[136:151]\n\t" MFMA, "s_nop 7\n\t", READ)
PROBE(b1_n13, MFMA, "s_nop 13\n\t", READ)
PROBE(b3_none, MFMA MFMA MFMA, "", READ)
PROBE(b3_n15, MFMA MFMA MFMA, "s_nop 15\n\t", READ)

// (c) the same read probe behind v_mfma_f32_16x16x4_f32 (8 passes): acc = v[104:107], A = v100, B = v101; reads v104 and v107
#define MFMA4 "v_mfma_f32_16x16x4_f32 v[104:107], v100, v101, v[104:107]\n\t"
#define READ4 "v_mov_b32 %[t0], v104\n\tv_mov_b32 %[t15], v107\n\t"
#define OUT4 "v_mov_b32 %[o0], v104\n\tv_mov_b32 %[o1], v105\n\tv_mov_b32 %[o2], v106\n\tv_mov_b32 %[o3], v107\n\t"
#define PROBE4(NAME, GAPSTR, PROBESTR)                                                                                            \
    __device__ __forceinline__ void NAME(u32x4 a, u32x4 b, float (&o)[4], float& t0, float& t15) {                                \
        asm volatile(SETB ZACC "s_nop 7\n\t" MFMA4 GAPSTR PROBESTR DRAIN OUT4                                                    \
                     : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [t0] "=&v"(t0), [t15] "=&v"(t15)   \
                     : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3])                                 \
                     : CLOB, "memory");                                                                                           \
    }
PROBE4(c_ref, DRAIN, NOREAD)
PROBE4(c_none, "", READ4)
PROBE4(c_n1, "s_nop 1\n\t", READ4)
PROBE4(c_n3, "s_nop 3\n\t", READ4)
PROBE4(c_n5, "s_nop 5\n\t", READ4)
PROBE4(c_n7, "s_nop 7\n\t", READ4)
PROBE4(c_n9, "s_nop 9\n\t", READ4)
PROBE4(c_n11, "s_nop 11\n\t", READ4)

// (d) VALU write of the accumulator (SrcC) right in front of the MFMA; (e) VALU write of SrcB right in front of the MFMA
#define ONEACC "v_mov_b32 v104, 1.0\n\tv_mov_b32 v105, 1.0\n\tv_mov_b32 v106, 1.0\n\tv_mov_b32 v107, 1.0\n\tv_mov_b32 v108, 1.0\n\tv_mov_b32 v109, 1.0\n\tv_mov_b32 v110, 1.0\n\tv_mov_b32 v111, 1.0\n\t" \
               "v_mov_b32 v112, 1.0\n\tv_mov_b32 v113, 1.0\n\tv_mov_b32 v114, 1.0\n\tv_mov_b32 v115, 1.0\n\tv_mov_b32 v116, 1.0\n\tv_mov_b32 v117, 1.0\n\tv_mov_b32 v118, 1.0\n\tv_mov_b32 v119, 1.0\n\t"
#define PROBE_D(NAME, PRE)                                                                                                        \
    __device__ __forceinline__ void NAME(u32x4 a, u32x4 b, float (&o)[4], float& t0, float& t15) {                                \
        asm volatile(PRE MFMA NOREAD DRAIN OUT                                                                                    \
                     : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [t0] "=&v"(t0), [t15] "=&v"(t15)   \
                     : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3])                                 \
                     : CLOB, "memory");                                                                                           \
    }
PROBE_D(d_ref, SETB ONEACC "s_nop 7\n\t")
PROBE_D(d_none, SETB "s_nop 7\n\t" ONEACC)
PROBE_D(e_none, ZACC "s_nop 7\n\t" SETB)

template <int V>
__global__ void __launch_bounds__(256, 1) probe(unsigned* bad, int iters) {
    const int lane = threadIdx.x & 63;
    const u32x4 a = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};      // bf16 1.0
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 b;
        for (int q = 0; q < 4; ++q) {       // 1, 1.5, 2, 2.5 ... : exact in bf16, sums exact in fp32
            const unsigned lo = 0x3F80u + (((lane + it + 2 * q) & 3) << 6), hi = 0x4000u + (((lane + 3 * it + q) & 1) << 6);
            b[q] = lo | (hi << 16);
        }
        float r[4], o[4], t0, t15, u0, u15;
        if constexpr (V == 7 || V == 14 || V == 15) ref3(a, b, r, u0, u15);
        else if constexpr (V == 30) d_ref(a, b, r, u0, u15);
        else if constexpr (V >= 20 && V < 30) c_ref(a, b, r, u0, u15);
        else ref1(a, b, r, u0, u15);
        if constexpr (V == 2) a1_none(a, b, o, t0, t15);
        if constexpr (V == 3) a1_n0(a, b, o, t0, t15);
        if constexpr (V == 4) a1_n1(a, b, o, t0, t15);
        if constexpr (V == 5) a1_n3(a, b, o, t0, t15);
        if constexpr (V == 6) a1_n7(a, b, o, t0, t15);
        if constexpr (V == 7) a3_none(a, b, o, t0, t15);
        if constexpr (V == 8) b1_none(a, b, o, t0, t15);
        if constexpr (V == 9) b1_n1(a, b, o, t0, t15);
        if constexpr (V == 10) b1_n3(a, b, o, t0, t15);
        if constexpr (V == 11) b1_n7(a, b, o, t0, t15);
        if constexpr (V == 12) b1_n15(a, b, o, t0, t15);
        if constexpr (V == 13) b1_n31(a, b, o, t0, t15);
        if constexpr (V == 14) b3_none(a, b, o, t0, t15);
        if constexpr (V == 15) b3_n15(a, b, o, t0, t15);
        if constexpr (V == 19) b1_n10(a, b, o, t0, t15);
        if constexpr (V == 40) b4i_n7(a, b, o, t0, t15);
        if constexpr (V == 41) b4i_n10(a, b, o, t0, t15);
        if constexpr (V == 42) b4i_n11(a, b, o, t0, t15);
        if constexpr (V == 16) b1_n9(a, b, o, t0, t15);
        if constexpr (V == 17) b1_n11(a, b, o, t0, t15);
        if constexpr (V == 18) b1_n13(a, b, o, t0, t15);
        if constexpr (V == 20) c_none(a, b, o, t0, t15);
        if constexpr (V == 21) c_n1(a, b, o, t0, t15);
        if constexpr (V == 22) c_n3(a, b, o, t0, t15);
        if constexpr (V == 23) c_n5(a, b, o, t0, t15);
        if constexpr (V == 24) c_n7(a, b, o, t0, t15);
        if constexpr (V == 25) c_n9(a, b, o, t0, t15);
        if constexpr (V == 26) c_n11(a, b, o, t0, t15);
        if constexpr (V == 30) d_none(a, b, o, t0, t15);
        if constexpr (V == 31) e_none(a, b, o, t0, t15);
        for (int k = 0; k < 4; ++k) nbad += (__builtin_bit_cast(unsigned, o[k]) != __builtin_bit_cast(unsigned, r[k])) ? 1u : 0u;
        if constexpr (V >= 8 && (V < 30 || V >= 40)) {
            nbad += (__builtin_bit_cast(unsigned, t0) != __builtin_bit_cast(unsigned, r[0])) ? 1u : 0u;
            nbad += (__builtin_bit_cast(unsigned, t15) != __builtin_bit_cast(unsigned, r[3])) ? 1u : 0u;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int V>
void run(unsigned* d_bad, const char* what) {
    (void)hipMemset(d_bad, 0, 4);
    hipLaunchKernelGGL((probe<V>), dim3(256), dim3(256), 0, 0, d_bad, 4000);
    (void)hipDeviceSynchronize();
    unsigned h = 0;
    (void)hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("%-86s -> %u mismatches (of 256 CUs x 256 lanes x 4000 rounds x %d checks)\n", what, h, (V >= 8 && (V < 30 || V >= 40)) ? 6 : 4);
}

int main() {
    unsigned* d_bad;
    (void)hipMalloc(&d_bad, 4);
    run<2>(d_bad, "(a) 1 MFMA, SrcB overwritten by the NEXT instruction");
    run<3>(d_bad, "(a) 1 MFMA, s_nop 0, SrcB overwritten");
    run<4>(d_bad, "(a) 1 MFMA, s_nop 1, SrcB overwritten");
    run<5>(d_bad, "(a) 1 MFMA, s_nop 3, SrcB overwritten");
    run<6>(d_bad, "(a) 1 MFMA, s_nop 7, SrcB overwritten");
    run<7>(d_bad, "(a) 3 dependent MFMAs, SrcB overwritten by the next instruction");
    run<8>(d_bad, "(b) 1 MFMA, acc[0] / acc[15] read by the NEXT instruction");
    run<9>(d_bad, "(b) 1 MFMA, s_nop 1, read");
    run<10>(d_bad, "(b) 1 MFMA, s_nop 3, read");
    run<11>(d_bad, "(b) 1 MFMA, s_nop 7, read");
    run<16>(d_bad, "(b) 1 MFMA, s_nop 9, read");
    run<19>(d_bad, "(b) 1 MFMA, s_nop 10, read   [= the 11 wait states hipcc's hazard table inserts for an 8-pass XDL op]");
    run<17>(d_bad, "(b) 1 MFMA, s_nop 11, read");
    run<18>(d_bad, "(b) 1 MFMA, s_nop 13, read");
    run<12>(d_bad, "(b) 1 MFMA, s_nop 15, read");
    run<13>(d_bad, "(b) 1 MFMA, 2 x s_nop 15, read");
    run<14>(d_bad, "(b) 3 dependent MFMAs, read by the next instruction");
    run<15>(d_bad, "(b) 3 dependent MFMAs, s_nop 15, read");
    run<40>(d_bad, "(b) 2 independent MFMAs in front, then the MFMA, s_nop 7, read");
    run<41>(d_bad, "(b) 2 independent MFMAs in front, then the MFMA, s_nop 10, read");
    run<42>(d_bad, "(b) 2 independent MFMAs in front, then the MFMA, s_nop 11, read");
    run<20>(d_bad, "(c) 1 v_mfma_f32_16x16x4_f32, acc[0] / acc[3] read by the NEXT instruction");
    run<21>(d_bad, "(c) f32 16x16x4, s_nop 1, read");
    run<22>(d_bad, "(c) f32 16x16x4, s_nop 3, read");
    run<23>(d_bad, "(c) f32 16x16x4, s_nop 5, read");
    run<24>(d_bad, "(c) f32 16x16x4, s_nop 7, read");
    run<25>(d_bad, "(c) f32 16x16x4, s_nop 9, read");
    run<26>(d_bad, "(c) f32 16x16x4, s_nop 11, read");
    run<30>(d_bad, "(d) accumulator (SrcC) written by v_mov right in front of the bf16 MFMA");
    run<31>(d_bad, "(e) SrcB written by v_mov right in front of the bf16 MFMA");
    return 0;
}
