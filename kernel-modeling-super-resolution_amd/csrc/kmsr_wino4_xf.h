// Packed-fp32 Winograd F(4x4,3x3) transform helpers shared by the forward / data-gradient kernel (conv3x3_wino4.hip) and the
// weight-gradient kernel (conv3x3_wgrad_wino4.hip).  Transform points {0, +-1, +-2, inf}:
//   B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
//   A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
//   G   = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
#pragma once

namespace kmsr {
namespace wino4xf {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// ---- packed-fp32 transform helpers.  v_pk_fma_f32 / v_pk_add_f32 cost the SIMD the same ~5 cycles as a scalar v_fma_f32 when
// clustered between f32 MFMAs (tools/ubench/valu_cluster.hip) but do two lanes of work, and op_sel / neg_hi pick and negate halves
// for free -- the transforms below are written on register PAIRS throughout.
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// 1-D input transform B^T d on a pair of independent lines (12 packed VALU for 2 x 6 outputs):
//   B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
__device__ __forceinline__ void bt6p(f32x2 d0, f32x2 d1, f32x2 d2, f32x2 d3, f32x2 d4, f32x2 d5, f32x2& o0, f32x2& o1, f32x2& o2,
                                     f32x2& o3, f32x2& o4, f32x2& o5) {
    const f32x2 a = fma2(splat2(-4.f), d2, d4), b = fma2(splat2(-4.f), d1, d3);
    const f32x2 c = d4 - d2, e = d3 - d1;
    o0 = fma2(splat2(4.f), d0, fma2(splat2(-5.f), d2, d4));
    o1 = a + b;
    o2 = a - b;
    o3 = fma2(splat2(2.f), e, c);
    o4 = fma2(splat2(-2.f), e, c);
    o5 = fma2(splat2(4.f), d1, fma2(splat2(-5.f), d3, d5));
}
// (x.y + x.x, x.y - x.x)
__device__ __forceinline__ f32x2 pk_hi_pm_lo(f32x2 x) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x));
    return r;
}
// (x.x + x.y, x.x - x.y)
__device__ __forceinline__ f32x2 pk_lo_pm_hi(f32x2 x) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x));
    return r;
}
// (x.y + 2 x.x, x.y - 2 x.x)
__device__ __forceinline__ f32x2 pk_hi_pm_2lo(f32x2 x) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(x));
    return r;
}
// (x.y * k.x + w.x, x.x * k.y + w.y): the halves of x swapped
__device__ __forceinline__ f32x2 pk_fma_swap(f32x2 x, f32x2 k, f32x2 w) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(k), "v"(w));
    return r;
}
// 1-D input transform of ONE line held as the pairs P0 = (t0,t5), P1 = (t1,t2), P2 = (t3,t4) (7 packed VALU):
//   returns O05 = (o0,o5), O12 = (o1,o2), O34 = (o3,o4).  K41 = (4,1), K54 = (-5,4), K15 = (1,-5).
__device__ __forceinline__ void bt6row(f32x2 P0, f32x2 P1, f32x2 P2, f32x2 K41, f32x2 K54, f32x2 K15, f32x2& O05, f32x2& O12, f32x2& O34) {
    const f32x2 BA = fma2(splat2(-4.f), P1, P2);            // (t3 - 4 t1, t4 - 4 t2) = (b, a)
    const f32x2 EC = P2 - P1;                               // (t3 - t1, t4 - t2)     = (e, c)
    O12 = pk_hi_pm_lo(BA);                                  // (a + b, a - b)
    O34 = pk_hi_pm_2lo(EC);                                 // (c + 2e, c - 2e)
    f32x2 w = P0 * K41;                                     // (4 t0, t5)
    w = pk_fma_swap(P1, K54, w);                            // (4 t0 - 5 t2, t5 + 4 t1)
    O05 = pk_fma_swap(P2, K15, w);                          // (4 t0 - 5 t2 + t4, t5 + 4 t1 - 5 t3)
}
// 1-D output transform A^T m on a pair of independent lines (10 packed VALU for 2 x 4 outputs):
//   A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
__device__ __forceinline__ void at6p(f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2& y0, f32x2& y1, f32x2& y2,
                                     f32x2& y3) {
    const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = fma2(splat2(2.f), d2, d1);
    y2 = fma2(splat2(4.f), s2, s1);
    y3 = fma2(splat2(8.f), d2, d1) + m5;
}
// (x.x + 2 x.y, x.x - 2 x.y)
__device__ __forceinline__ f32x2 pk_lo_pm_2hi(f32x2 x) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(x));
    return r;
}

}  // namespace wino4xf
}  // namespace kmsr
