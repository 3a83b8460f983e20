// Counter-based Gaussian noise stream shared by the degradation forward and backward kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace kmsr {

// ------------------------------------------------------------------------------------------------
// Counter-based Gaussian stream (see oracle/kmsr_oracle.c: oracle_eps -- the two must stay in sync)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ static void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: the integer multiplier is a
        // quarter-rate unit, and the 40 + 40 multiplies of the ten rounds were more VALU time than the 13 x 13 fold they feed
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four normals for rows 4*iblk .. 4*iblk+3 of column j in `plane`
__device__ __forceinline__ static void eps4(unsigned long long seed, unsigned offset, unsigned plane, unsigned iblk,
                                     unsigned j, float z[4]) {
    unsigned r[4];
    philox4x32_10(j, iblk, plane, offset, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const float u1 = ((float)(r[2 * pr] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float u2 = (float)(r[2 * pr + 1] >> 8) * (1.0f / 16777216.0f);
        // Box-Muller on the transcendental unit: v_log_f32 (log2; u1 >= 2^-24 is a normal number), v_sqrt_f32, v_sin_f32 / v_cos_f32
        // (argument in revolutions = u2 itself).  The oracle evaluates the same formula in double; tests hold |eps - oracle| <= 3e-6.
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // -2 ln 2 * log2(u1)
        z[2 * pr] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * pr + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
}

__device__ __forceinline__ static float pick4(const float z[4], int k) {
    return k == 0 ? z[0] : (k == 1 ? z[1] : (k == 2 ? z[2] : z[3]));
}


}  // namespace kmsr
