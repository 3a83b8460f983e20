// Microbenchmark: how many independent VALU / LDS-read instructions fit "for free" between fp32 MFMAs on gfx950,
// at 1 and 2 waves per SIMD.  Prints cycles per MFMA for k filler instructions per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int SHAPE, int KV, int KL>
__global__ void __launch_bounds__(512, 2) bench(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a = 1.0f + lane, b = 0.5f * lane;
    f32x4 acc4[8];
    f32x16 acc16[4];
    for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f;
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float l[4] = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if constexpr (SHAPE == 16) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m], 0, 0, 0);
            else acc16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KV; ++k) { v[(m + k) & 7] = v[(m + k) & 7] * 1.0001f + 0.5f; asm volatile("" : "+v"(v[(m + k) & 7])); }
#pragma unroll
            for (int k = 0; k < KL; ++k) { l[k & 3] += lds[(lane * 2 + 64 * k + m) & 4095]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc4[i][0] + v[i];
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);   // slowest wave of the block
}

template <int SHAPE, int KV, int KL>
void run(int waves_per_simd, float* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256;
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((bench<SHAPE, KV, KL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipMemset(cyc, 0, blocks * 8);
    hipLaunchKernelGGL((bench<SHAPE, KV, KL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("shape %2d  waves/SIMD %d  VALU/MFMA %d  LDS/MFMA %d : %.1f clk per MFMA per wave (%.1f per SIMD MFMA)\n", SHAPE, waves_per_simd, KV, KL,
           m / (iters * 8.0), m / (iters * 8.0) / waves_per_simd);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<16, 0, 0>(w, out, cyc); run<16, 1, 0>(w, out, cyc); run<16, 2, 0>(w, out, cyc); run<16, 3, 0>(w, out, cyc); run<16, 4, 0>(w, out, cyc); run<16, 6, 0>(w, out, cyc);
        run<16, 0, 1>(w, out, cyc); run<16, 2, 1>(w, out, cyc);
        run<32, 0, 0>(w, out, cyc); run<32, 2, 0>(w, out, cyc); run<32, 4, 0>(w, out, cyc); run<32, 8, 0>(w, out, cyc); run<32, 12, 0>(w, out, cyc); run<32, 4, 2>(w, out, cyc);
    }
    return 0;
}
