// Microbenchmark (gfx950, ONE wave per SIMD): what a CLUSTER of K non-MFMA instructions between two groups of 8 independent
// v_mfma_f32_16x16x4_f32 costs the SIMD, per instruction, by instruction kind -- the regime of conv3x3_wino4.hip (transform VALU is
// clustered between MFMA groups because the f32 MFMA and the VALU do not co-issue).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_cluster.hip -o /tmp/valu_cluster && /tmp/valu_cluster
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

enum { K_FMA = 0, K_PKFMA, K_PKADD, K_FMA_DEP, K_SALU, K_DSREAD128, K_ACCREAD, K_PKFMA_OPSEL, K_ADD };

template <int KIND, int K>
__global__ void __launch_bounds__(256, 1) bench(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a = 1.0f + lane, b = 0.5f * lane;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[16];
    f32x2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = i + lane; p[i] = f32x2{(float)i, (float)lane}; }
    int s = iters;
    f32x4 l4 = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int r = k & 15;
            if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(a), "v"(b));
            else if constexpr (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(a));
            else if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(p[(r + 1) & 15]), "v"(p[(r + 2) & 15]));
            else if constexpr (KIND == K_PKFMA_OPSEL) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p[r]) : "v"(p[(r + 1) & 15]), "v"(p[(r + 2) & 15]));
            else if constexpr (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 15]));
            else if constexpr (KIND == K_FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 1]) : "v"(a), "v"(b));
            else if constexpr (KIND == K_SALU) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
            else if constexpr (KIND == K_DSREAD128) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((lane * 16 + k * 1024) & 16383)); l4 += t; }
            else if constexpr (KIND == K_ACCREAD) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(v[(r + 3) & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = (float)s + l4.x + l4.y;
    for (int i = 0; i < 8; ++i) sum += acc[i][0];
    for (int i = 0; i < 16; ++i) sum += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

template <int KIND, int K>
double run(const char* name, float* out, unsigned long long* cyc, double base) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((bench<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipMemset(cyc, 0, blocks * 8);
    hipLaunchKernelGGL((bench<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    const double per_iter = m / iters;
    if (K) printf("%-34s K=%2d : %7.1f clk per (8 MFMA + cluster)  -> %5.2f clk per clustered instruction\n", name, K, per_iter, (per_iter - base) / K);
    else printf("%-34s      : %7.1f clk per 8 MFMAs (floor 256)\n", name, per_iter);
    return per_iter;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const double base = run<K_FMA, 0>("8 x v_mfma_f32_16x16x4_f32 only", out, cyc, 0);
    run<K_FMA, 16>("v_fma_f32 (independent)", out, cyc, base);     run<K_FMA, 48>("v_fma_f32 (independent)", out, cyc, base);
    run<K_ADD, 48>("v_add_f32 (independent)", out, cyc, base);
    run<K_FMA_DEP, 48>("v_fma_f32 (2 dependent chains)", out, cyc, base);
    run<K_PKFMA, 16>("v_pk_fma_f32", out, cyc, base);              run<K_PKFMA, 48>("v_pk_fma_f32", out, cyc, base);
    run<K_PKFMA_OPSEL, 48>("v_pk_fma_f32 op_sel mixed halves", out, cyc, base);
    run<K_PKADD, 48>("v_pk_add_f32", out, cyc, base);
    run<K_ACCREAD, 48>("v_mov_b32", out, cyc, base);
    run<K_SALU, 16>("s_add_i32", out, cyc, base);                  run<K_SALU, 48>("s_add_i32", out, cyc, base);
    run<K_DSREAD128, 16>("ds_read_b128 (conflict-free)", out, cyc, base);
    return 0;
}
