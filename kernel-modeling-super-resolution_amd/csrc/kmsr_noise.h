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
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
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
        const float rad = sqrtf(-2.0f * logf(u1));
        float s, c;
        sincospif(2.0f * u2, &s, &c);
        z[2 * pr] = rad * c;
        z[2 * pr + 1] = rad * s;
    }
}

__device__ __forceinline__ static float pick4(const float z[4], int k) {
    return k == 0 ? z[0] : (k == 1 ? z[1] : (k == 2 ? z[2] : z[3]));
}


}  // namespace kmsr
