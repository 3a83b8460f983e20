// Microbenchmark (gfx950, one wave per SIMD): cost of vector-memory instructions issued BETWEEN fp32 MFMAs of the same wave.
//   every PER MFMAs: nothing | one 16-byte-per-lane buffer store | one 16-byte-per-lane buffer load (L2-resident, result unused until the end)
//   | one ds_read_b128 | one 16-byte-per-lane buffer load straight into LDS (LDS-DMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int MODE, int PER, int SPAN>   // MODE 0 none, 1 store, 2 load; one memory instruction per PER MFMAs; stores cycle through SPAN x 8 KB per wave
__global__ void __launch_bounds__(256, 1) bench(float* big, const float* small, unsigned long long* cyc, float* sink, int iters, long long stride) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 8 * 256];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4 * 8 * 256; i += 256) lds[i] = (float)i;
    __syncthreads();
    float a = 1.0f + lane, b = 0.5f * lane;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float* mine = big + ((long long)blockIdx.x * 4 + wv) * stride;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0, 1u << 20, 0x00020000);
    f32x4 l[8];
    for (int i = 0; i < 8; ++i) l[i] = f32x4{0, 0, 0, 0};
    f32x4 d = {1, 2, 3, 4};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const auto rb = __builtin_amdgcn_make_buffer_rsrc(mine + (long long)(it & (SPAN - 1)) * 2048, 0, 1u << 16, 0x00020000);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (PER <= 8 ? (m % PER == PER - 1) : (m == 7 && (it & 1))) {
                if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), rb, lane * 16 + (m / PER) * 1024, 0, 0);
                if (MODE == 2) l[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((it * 8 + m) * 1024 + lane * 16) & 0xFFFF0, 0, 0));   // overwritten a loop later: no wait inside the loop body except the WAW one
                if (MODE == 3) l[m] = *reinterpret_cast<const f32x4*>(lds + (wv * 8 + m) * 256 + lane * 4);
                if (MODE == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wv * 8 + m) * 256), 16, ((it * 8 + m) * 1024 + lane * 16) & 0xFFFF0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = lds[(threadIdx.x * 7) & 8191];
    for (int i = 0; i < 8; ++i) s += l[i].x;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}
template <int MODE, int PER, int SPAN>
void run(const char* name, float* big, const float* small, unsigned long long* cyc, float* sink, long long stride) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((bench<MODE, PER, SPAN>), dim3(blocks), dim3(256), 0, 0, big, small, cyc, sink, iters, stride);
    (void)hipMemset(cyc, 0, blocks * 8);
    hipLaunchKernelGGL((bench<MODE, PER, SPAN>), dim3(blocks), dim3(256), 0, 0, big, small, cyc, sink, iters, stride);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= blocks;
    printf("%-52s %.1f clk per MFMA\n", name, m / (iters * 8.0));
}
int main() {
    float *small, *big, *sink; unsigned long long* cyc;
    const long long stride = 256LL * 2048 + 2048;     // floats per wave: 2 MB regions
    (void)hipMalloc(&small, 1 << 20); (void)hipMalloc(&big, 1024 * stride * 4); (void)hipMalloc(&sink, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(small, 0, 1 << 20);
    run<0, 8, 1>("MFMAs only", big, small, cyc, sink, stride);
    run<1, 8, 1>("store per 8 MFMAs, 8 KB per wave rewritten (L2)", big, small, cyc, sink, stride);
    run<1, 4, 1>("store per 4 MFMAs, L2", big, small, cyc, sink, stride);
    run<1, 2, 1>("store per 2 MFMAs, L2", big, small, cyc, sink, stride);
    run<1, 8, 256>("store per 8 MFMAs, 2 MB per wave (HBM stream)", big, small, cyc, sink, stride);
    run<1, 16, 256>("store per 16 MFMAs, HBM stream", big, small, cyc, sink, stride);
    run<2, 8, 1>("load per 8 MFMAs (L2-resident), no consumer", big, small, cyc, sink, stride);
    run<2, 4, 1>("load per 4 MFMAs", big, small, cyc, sink, stride);
    run<2, 2, 1>("load per 2 MFMAs", big, small, cyc, sink, stride);
    run<3, 8, 1>("ds_read_b128 per 8 MFMAs, no consumer", big, small, cyc, sink, stride);
    run<3, 4, 1>("ds_read_b128 per 4 MFMAs", big, small, cyc, sink, stride);
    run<3, 2, 1>("ds_read_b128 per 2 MFMAs", big, small, cyc, sink, stride);
    run<4, 8, 1>("16-byte LDS-DMA load per 8 MFMAs (L2-resident)", big, small, cyc, sink, stride);
    run<4, 4, 1>("16-byte LDS-DMA load per 4 MFMAs", big, small, cyc, sink, stride);
    return 0;
}
