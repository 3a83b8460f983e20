// Microbenchmark (gfx950, one wave per SIMD, 256 workgroups): what does a buffer store cost the wave that issues it?
//   body per iteration: FILL independent packed-fp32 VALU instructions (or MFMAs), then one store of WIDTH dwords per lane into an
//   8 KB per-wave region that stays in L2 (no HBM stream), with 1 .. 256 workgroups running.  Reported: cycles per iteration minus the same loop without the store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int WIDTH, int FILL, bool MFMA>   // WIDTH 0: no store
__global__ void __launch_bounds__(256, 1) bench(float* big, unsigned long long* cyc, float* sink, int iters, long long stride) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* mine = big + ((long long)blockIdx.x * 4 + wv) * stride;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 1u << 16, 0x00020000);
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)i, (float)lane};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x4 d = {1, 2, 3, (float)lane};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < FILL; ++k) {
            if (MFMA) acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(d.x, d.w, acc[k & 7], 0, 0, 0);
            else { v[k & 7] = v[k & 7] * f32x2{1.0001f, 0.9999f} + f32x2{0.5f, 0.25f}; asm volatile("" : "+v"(v[k & 7])); }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned off = (unsigned)(lane * 4 * WIDTH + (it & 7) * 256 * WIDTH);
        if (WIDTH == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), rb, off, 0, 0);
        if (WIDTH == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{d.x, d.w}), rb, off, 0, 0);
        if (WIDTH == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d.w), rb, off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i].x + acc[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}
template <int WIDTH, int FILL, bool MFMA>
double run(float* big, unsigned long long* cyc, float* sink, long long stride, int blocks) {
    const int iters = 4000;
    hipLaunchKernelGGL((bench<WIDTH, FILL, MFMA>), dim3(blocks), dim3(256), 0, 0, big, cyc, sink, iters, stride);
    (void)hipMemset(cyc, 0, blocks * 8);
    hipLaunchKernelGGL((bench<WIDTH, FILL, MFMA>), dim3(blocks), dim3(256), 0, 0, big, cyc, sink, iters, stride);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    return m / iters;
}
template <int FILL, bool MFMA>
void row(float* big, unsigned long long* cyc, float* sink, long long stride, int blocks) {
    const double b = run<0, FILL, MFMA>(big, cyc, sink, stride, blocks);
    const double s4 = run<4, FILL, MFMA>(big, cyc, sink, stride, blocks), s2 = run<2, FILL, MFMA>(big, cyc, sink, stride, blocks), s1 = run<1, FILL, MFMA>(big, cyc, sink, stride, blocks);
    printf("%3d workgroups, %2d %s per store: loop %.0f clk;  + dwordx4 store %.0f   + dwordx2 store %.0f   + dword store %.0f\n", blocks, FILL, MFMA ? "MFMAs   " : "pk VALU ", b, s4 - b, s2 - b, s1 - b);
}
int main() {
    float *big, *sink; unsigned long long* cyc;
    const long long stride = 4096;     // floats per wave: 16 KB regions
    (void)hipMalloc(&big, 1024 * stride * 4); (void)hipMalloc(&sink, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    for (int blocks : {1, 8, 64, 256}) {
        row<8, false>(big, cyc, sink, stride, blocks);
        row<32, false>(big, cyc, sink, stride, blocks);
        row<8, true>(big, cyc, sink, stride, blocks);
        row<32, true>(big, cyc, sink, stride, blocks);
    }
    return 0;
}
