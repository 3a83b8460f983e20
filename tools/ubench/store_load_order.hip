// Microbenchmark (gfx950): does a wave's wait for a LOAD also wait for an OLDER STORE (one in-order vmcnt)?  And how long does a store take
// to be acknowledged?  One wave per CU.  Modes: load only | store only | store, then load, wait for the load's data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int MODE>   // 0: load, 1: store, 2: store + load
__global__ void __launch_bounds__(64, 1) bench(const float* small, float* big, unsigned long long* cyc, float* sink, int iters, long long stride) {
    const int lane = threadIdx.x;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0, 1u << 20, 0x00020000);
    float* mybig = big + (long long)blockIdx.x * stride;
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long total = 0;
    for (int it = 0; it < iters; ++it) {
        const auto rb = __builtin_amdgcn_make_buffer_rsrc(mybig + (long long)(it & 1023) * 4096, 0, 1u << 16, 0x00020000);
        f32x4 v = acc + (float)it;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE >= 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rb, lane * 16, 0, 0);
        f32x4 l = {0, 0, 0, 0};
        if (MODE != 1) l = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((it * 64 + lane) * 16) & 0xFFFF0, 0, 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        acc += l;
        total += t1 - t0;
    }
    if (lane == 0) cyc[blockIdx.x] = total;
    sink[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE>
void run(const char* name, const float* small, float* big, unsigned long long* cyc, float* sink, int blocks, long long stride) {
    const int iters = 4000;
    hipLaunchKernelGGL((bench<MODE>), dim3(blocks), dim3(64), 0, 0, small, big, cyc, sink, iters, stride);
    hipLaunchKernelGGL((bench<MODE>), dim3(blocks), dim3(64), 0, 0, small, big, cyc, sink, iters, stride);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("%-44s %3d waves: %.0f cycles until vmcnt(0)\n", name, blocks, m / iters);
}
int main() {
    float *small, *big, *sink; unsigned long long* cyc;
    const long long stride = 1024LL * 4096 + 4096;     // floats per block: 16 MB apart
    (void)hipMalloc(&small, 1 << 20); (void)hipMalloc(&big, 256 * stride * 4); (void)hipMalloc(&sink, 256 * 64 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(small, 0, 1 << 20);
    for (int blocks : {1, 256}) {
        run<0>("load (L2-resident 1 MB buffer)", small, big, cyc, sink, blocks, stride);
        run<1>("16-byte-per-lane store (fresh 16 KB apart)", small, big, cyc, sink, blocks, stride);
        run<2>("store, then load: wait for both", small, big, cyc, sink, blocks, stride);
    }
    return 0;
}
