// Microbenchmark: what does it cost an MFMA-bound wave to bring 1 KB per instruction from L2 into LDS
//   (a) as LDS-DMA (global_load_lds, 16 B per lane), or
//   (b) through registers (buffer_load_dwordx4 now, ds_write_b128 one iteration later)?
// 8 waves per workgroup (2 per SIMD), 256 workgroups; per iteration a wave issues NL loads and 32 independent fp32 MFMAs
// (the winograd kernel's ratio is 6 loads per 64 MFMAs).  Prints cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((address_space(3))) void lds_void_t;

template <int MODE, int NL>   // MODE 0: no loads, 1: LDS-DMA, 2: registers
__global__ void __launch_bounds__(512, 2) bench(const float* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 8 * 4 * 256];      // 2 slots x 8 waves x NL(<=4) x 1 KB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a = 1.0f + lane, b = 0.5f * lane;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float* base = src + (size_t)(blockIdx.x & 7) * 8192 + wv * 1024 + lane * 4;     // 32 KB regions, L2 resident
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 8u * 8192u * 4u, 0x00020000);
    const unsigned voff = (unsigned)((blockIdx.x & 7) * 8192 + wv * 1024 + lane * 4) * 4u;
    f32x4 regs[4];
    for (int i = 0; i < 4; ++i) regs[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float* slot = lds + ((it & 1) * 8 + wv) * 4 * 256;
        if constexpr (MODE == 2) {        // last iteration's loads -> LDS
#pragma unroll
            for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4*>(slot + i * 256 + lane * 4) = regs[i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(base + i * 256 + (it & 3) * 2048), (lds_void_t*)(slot + i * 256), 16, 0, 0);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                regs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)(i * 256 + (it & 3) * 2048) * 4u, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (NL & 15));     // vmcnt <= NL: the previous iteration's DMA landed
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    float s = lds[threadIdx.x];
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 4; ++i) s += regs[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

template <int MODE, int NL>
void run(const float* src, float* out, unsigned long long* cyc, const char* name) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((bench<MODE, NL>), dim3(blocks), dim3(512), 0, 0, src, out, cyc, iters);
    (void)hipMemset(cyc, 0, blocks * 8);
    hipLaunchKernelGGL((bench<MODE, NL>), dim3(blocks), dim3(512), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("%-28s %d x 1 KB per wave and 32 MFMAs: %8.1f clk per iteration (MFMA-only floor 2 waves x 32 x 32 = 2048)\n", name, NL, m / iters);
}
int main() {
    float *src, *out; unsigned long long* cyc;
    (void)hipMalloc(&src, 8 * 8192 * 4 + 65536); (void)hipMemset(src, 0, 8 * 8192 * 4 + 65536);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    run<0, 0>(src, out, cyc, "no loads");
    run<1, 1>(src, out, cyc, "LDS-DMA"); run<1, 2>(src, out, cyc, "LDS-DMA"); run<1, 3>(src, out, cyc, "LDS-DMA"); run<1, 4>(src, out, cyc, "LDS-DMA");
    run<2, 1>(src, out, cyc, "buffer_load + ds_write"); run<2, 2>(src, out, cyc, "buffer_load + ds_write");
    run<2, 3>(src, out, cyc, "buffer_load + ds_write"); run<2, 4>(src, out, cyc, "buffer_load + ds_write");
    return 0;
}
