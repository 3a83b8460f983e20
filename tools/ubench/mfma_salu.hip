// Microbenchmark: do scalar-ALU instructions of the SAME wave hide in the shadow of an fp32 MFMA on gfx950 (one wave per SIMD)?
//   spread : MFMA, k x s_add_i32 (dependent chain), MFMA, ...      cluster: 8 MFMAs, then 8 k x s_add_i32
// Prints cycles per MFMA (floor 32 for v_mfma_f32_16x16x4_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KS, bool CLUSTER>
__global__ void __launch_bounds__(256, 1) bench(float* out, unsigned long long* cyc, int iters, int seed) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane, b = 0.5f * lane;
    f32x4 acc4[8];
    for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0, 0, 0, 0};
    int s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 + 1;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!CLUSTER) {
#pragma unroll
                for (int k = 0; k < KS; ++k) asm volatile("s_add_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (CLUSTER) {
#pragma unroll
            for (int k = 0; k < 8 * KS; ++k) asm volatile("s_add_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)s0;
    for (int i = 0; i < 8; ++i) s += acc4[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}
template <int KS, bool CLUSTER>
void run(float* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((bench<KS, CLUSTER>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 3);
    hipMemset(cyc, 0, blocks * 8);
    hipLaunchKernelGGL((bench<KS, CLUSTER>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 3);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("%s  SALU/MFMA %d : %.1f clk per MFMA\n", CLUSTER ? "cluster" : "spread ", KS, m / (iters * 8.0));
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, false>(out, cyc); run<1, false>(out, cyc); run<2, false>(out, cyc); run<3, false>(out, cyc); run<4, false>(out, cyc); run<6, false>(out, cyc);
    run<1, true>(out, cyc); run<2, true>(out, cyc); run<4, true>(out, cyc);
    return 0;
}
