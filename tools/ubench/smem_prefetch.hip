// Microbenchmark: is the SCALAR memory path (s_load -> scalar cache -> L2) a second road into L2 next to the per-CU vector L1
// (whose ~64 outstanding 128-byte misses bound every streaming kernel here at ~20 GB/s per CU)?
//   A. line-touch rates over 2 GiB of cold HBM: one s_load_dword per 128-byte line (16 in flight per wave) against one
//      global_load_dword lane per line and against a coalesced float4 read; 4 .. 32 waves per CU.
//   B. a row-streaming consumer shaped like degrade_rowstream_kernel (wave-private 3-slot LDS-DMA ring of 2 KB rows, 4 waves per
//      workgroup, 6 workgroups per CU) with a 5th wave per workgroup that touches the consumers' rows AHEAD rows before they are
//      requested, through the scalar path only (progress counters in LDS; the consumers never wait for it).
// Build: hipcc -O3 --offload-arch=gfx950 smem_prefetch.hip -o smem_prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((address_space(3))) void lds_void_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 16 lines (2 KB) behind p: one dword of each, all in flight together, waited for; returns the xor (keeps the loads alive)
__device__ __forceinline__ unsigned touch16(const float* p) {
    unsigned r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    asm volatile(
        "s_load_dword %0, %16, 0x0\n s_load_dword %1, %16, 0x80\n s_load_dword %2, %16, 0x100\n s_load_dword %3, %16, 0x180\n"
        "s_load_dword %4, %16, 0x200\n s_load_dword %5, %16, 0x280\n s_load_dword %6, %16, 0x300\n s_load_dword %7, %16, 0x380\n"
        "s_load_dword %8, %16, 0x400\n s_load_dword %9, %16, 0x480\n s_load_dword %10, %16, 0x500\n s_load_dword %11, %16, 0x580\n"
        "s_load_dword %12, %16, 0x600\n s_load_dword %13, %16, 0x680\n s_load_dword %14, %16, 0x700\n s_load_dword %15, %16, 0x780\n"
        "s_waitcnt lgkmcnt(0)\n"
        : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4), "=&s"(r5), "=&s"(r6), "=&s"(r7), "=&s"(r8), "=&s"(r9), "=&s"(r10),
          "=&s"(r11), "=&s"(r12), "=&s"(r13), "=&s"(r14), "=&s"(r15)
        : "s"(p)
        : "memory");
    return r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ r8 ^ r9 ^ r10 ^ r11 ^ r12 ^ r13 ^ r14 ^ r15;
}

// ---- A: MODE 0 scalar touch, 1 vector touch (one lane per line: 64 lines = 8 KB per instruction), 2 coalesced float4 read
template <int MODE>
__global__ void __launch_bounds__(256) touch_kernel(const float* __restrict__ src, size_t bytes, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    unsigned acc = 0;
    if constexpr (MODE == 0) {
        const size_t chunks = bytes / 2048;
        for (size_t c = wave; c < chunks; c += nwaves) {
            const unsigned long long a = (unsigned long long)src + c * 2048ull;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            acc ^= touch16((const float*)(((unsigned long long)hi << 32) | lo));
        }
    } else if constexpr (MODE == 1) {
        const size_t chunks = bytes / 8192;                   // 64 lines per wave instruction, 4 instructions in flight
        for (size_t c = wave * 4; c + 3 < chunks; c += nwaves * 4) {
            unsigned v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load((const unsigned*)src + ((c + k) * 8192 + (size_t)lane * 128) / 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= v[k];
        }
    } else {
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
        const size_t chunks = bytes / 1024;                   // 1 KB per wave instruction, 8 in flight
        for (size_t c = wave * 8; c + 7 < chunks; c += nwaves * 8) {
            u4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *((const u4*)src + (c + k) * 64 + lane);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;                     // never true for the zero-filled buffer: keeps the loads
}

// ---- B: 4 consumer waves + 1 scalar prefetch wave per workgroup
constexpr int ROWF = 512;                                     // floats per row (2 KB = 2 LDS-DMA instructions of 1 KB)
template <int AHEAD>                                          // 0: the 5th wave exits at once
__global__ void __launch_bounds__(320) stream_kernel(const float* __restrict__ src, int rows_per_wave, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 3 * ROWF + 16];   // ONE LDS object (a second one costs a vmcnt(0) before every ds_read behind an LDS-DMA)
    int* prog = reinterpret_cast<int*>(lds + 4 * 3 * ROWF);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 4) prog[threadIdx.x] = 0;
    __syncthreads();                                          // the only block barrier (before any wave diverges)
    const int n = rows_per_wave;
    const size_t blk_base = (size_t)blockIdx.x * 4 * n * ROWF;
    if (wv < 4) {
        const float* base = src + blk_base + (size_t)wv * n * ROWF;
        float* my = lds + wv * 3 * ROWF;
        const unsigned paddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) int*)(prog + wv);   // all lanes store the same value
        auto issue = [&](int r) {
            const float* rp = base + (size_t)r * ROWF;
            float* slot = my + (r % 3) * ROWF;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(rp + 4 * lane), (lds_void_t*)(slot), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(rp + 256 + 4 * lane), (lds_void_t*)(slot + 256), 16, 0, 0);
        };
        issue(0);
        if (n > 1) issue(1);
        float s = 0.f;
        for (int r = 0; r < n; ++r) {
            if (r + 2 < n) issue(r + 2);
            const int left = min(2, n - 1 - r);
            if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(my + (r % 3) * ROWF + 4 * lane);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(my + (r % 3) * ROWF + 256 + 4 * lane);
            s += v0.x + v0.w + v1.y + v1.z;                   // (uses the values: the LDS reads have retired before the slot is requested again)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (AHEAD > 0) asm volatile("ds_write_b32 %0, %1" ::"v"(paddr), "v"(r + 1) : "memory");   // (inline asm: a compiler-visible LDS store behind an LDS-DMA gets a vmcnt(0))
        }
        if (AHEAD > 0) asm volatile("ds_write_b32 %0, %1" ::"v"(paddr), "v"(n) : "memory");
        if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if (AHEAD > 0) {
        int pf[4] = {2, 2, 2, 2};                            // rows 0, 1 are requested by the consumers' prologues
        unsigned acc = 0;
        for (int guard = 0; guard < (1 << 24); ++guard) {     // bounded even if a counter were never published
            bool done = true, any = false;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int pr = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&prog[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                pf[w] = max(pf[w], min(pr + 3, n));          // never behind what the consumer has already requested
                const int lim = min(pr + 2 + AHEAD, n);
                if (pf[w] < lim) {
                    acc ^= touch16(src + blk_base + ((size_t)w * n + pf[w]) * ROWF);
                    ++pf[w];
                    any = true;
                }
                if (pf[w] < n) done = false;
            }
            if (done) break;
            if (!any) __builtin_amdgcn_s_sleep(4);
        }
        if (acc == 0x12345678u) out[0] = (float)acc;
    }
}


// ---- C: the MEMORY SIDE of degrade_rowstream_kernel alone, feature by feature.  Image-shaped units: plane = 1024 x 1024 floats, unit =
//      (plane, strip of 512 HR columns, chunk of CR HR rows), one wave per unit, 4 waves per workgroup, 3-slot LDS-DMA ring, one
//      ds_read_b128 pair per row instead of the fold.  flags: 1 = 9 halo rows per chunk (TY - F), 2 = segment starts 8 floats left of the
//      strip and carries 8 more on the right (a third, 4-lane DMA instruction; 17 lines per row instead of 16), 4 = odd chunks walk
//      bottom-up, 8 = XCD-grouped unit order, 16 = whole-row strips (1024 columns, 4 DMA instructions per row, no column halo at all).
struct CParams { int planes, chunk_rows, flags, n_units, n_blocks; };
__global__ void __launch_bounds__(256) image_stream_kernel(const float* __restrict__ src, CParams p, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 3 * 1056];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int blk = blockIdx.x;
    if (p.flags & 8) {
        const int nb = p.n_blocks, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
        blk = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    }
    const int unit = blk * 4 + wv;
    if (unit >= p.n_units) return;
    const bool wide = (p.flags & 16) != 0;
    const int n_chunks = 1024 / p.chunk_rows, n_strips = wide ? 1 : 2;
    const int chunk = unit % n_chunks, strip = (unit / n_chunks) % n_strips, plane = unit / (n_chunks * n_strips);
    const bool halo = (p.flags & 1) != 0, mis = (p.flags & 2) != 0 && !wide, up = (p.flags & 4) != 0 && (chunk & 1);
    const int r0 = chunk * p.chunk_rows - (halo ? 6 : 0), n = p.chunk_rows + (halo ? 9 : 0);
    const int ninstr = wide ? 4 : (mis ? 3 : 2);
    const int rowf = wide ? 1024 : 528;
    const float* img = src + (size_t)plane * 1024 * 1024;
    const int col0 = strip * 512 - (mis ? 8 : 0);
    float* my = lds + wv * 3 * 1056;
    auto issue = [&](int k) {
        int r = up ? r0 + n - 1 - k : r0 + k;
        r = min(max(r, 0), 1023);
        const float* rp = img + (size_t)r * 1024 + col0;
        float* slot = my + (k % 3) * rowf;
        for (int t = 0; t < ninstr; ++t) {
            const int c = col0 + 4 * (lane + 64 * t);
            const bool in = c >= 0 && c + 3 < 1024 && (4 * (lane + 64 * t) < (mis ? 528 : rowf));
            if (in) __builtin_amdgcn_global_load_lds((glb_void_t*)(rp + 4 * (lane + 64 * t)), (lds_void_t*)(slot + 256 * t), 16, 0, 0);
        }
    };
    issue(0);
    issue(1);
    float s = 0.f;
    for (int k = 0; k < n; ++k) {
        if (k + 2 < n) issue(k + 2);
        const int left = min(2, n - 1 - k) * ninstr;
        switch (left) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(my + (k % 3) * rowf + 4 * lane);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(my + (k % 3) * rowf + 256 + 4 * lane);
        s += v0.x + v0.w + v1.y + v1.z;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (s == 123.456f) out[unit] = s;
}

template <typename F>
float time_ms(F&& launch, int reps = 3) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    return best;
}

int main() {
    const size_t bytes = 2ull << 30;
    float* src; float* out;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMemset(src, 0, bytes));
    CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipDeviceSynchronize());
    printf("A. line-touch / read rates over %zu MiB (each line once per launch; bytes counted = lines x 128)\n", bytes >> 20);
    for (int wpc : {4, 8, 16, 32}) {
        const int blocks = 256 * wpc / 4;
        const float t0 = time_ms([&] { hipLaunchKernelGGL(touch_kernel<0>, dim3(blocks), dim3(256), 0, 0, src, bytes, (unsigned*)out); });
        const float t1 = time_ms([&] { hipLaunchKernelGGL(touch_kernel<1>, dim3(blocks), dim3(256), 0, 0, src, bytes, (unsigned*)out); });
        const float t2 = time_ms([&] { hipLaunchKernelGGL(touch_kernel<2>, dim3(blocks), dim3(256), 0, 0, src, bytes, (unsigned*)out); });
        printf("  %2d waves/CU: scalar touch %7.3f ms = %6.0f GB/s | vector touch %7.3f ms = %6.0f GB/s | float4 read %7.3f ms = %6.0f GB/s\n", wpc,
               t0, bytes / t0 * 1e-6, t1, bytes / t1 * 1e-6, t2, bytes / t2 * 1e-6);
        fflush(stdout);
    }
    printf("B. row-streaming consumers (LDS-DMA ring, 24 waves/CU) with a scalar prefetch wave per workgroup\n");
    const int blocks = 256 * 6, rows = (int)(bytes / ((size_t)blocks * 4 * ROWF * 4));
    const size_t moved = (size_t)blocks * 4 * rows * ROWF * 4;
    const float b0 = time_ms([&] { hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(320), 0, 0, src, rows, out); });
    printf("  no prefetch        %7.3f ms = %6.0f GB/s (%d rows of 2 KB per wave)\n", b0, moved / b0 * 1e-6, rows); fflush(stdout);
    const float b4 = time_ms([&] { hipLaunchKernelGGL(stream_kernel<4>, dim3(blocks), dim3(320), 0, 0, src, rows, out); });
    printf("  prefetch 4 ahead   %7.3f ms = %6.0f GB/s\n", b4, moved / b4 * 1e-6); fflush(stdout);
    const float b8 = time_ms([&] { hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(320), 0, 0, src, rows, out); });
    printf("  prefetch 8 ahead   %7.3f ms = %6.0f GB/s\n", b8, moved / b8 * 1e-6); fflush(stdout);
    const float b16 = time_ms([&] { hipLaunchKernelGGL(stream_kernel<16>, dim3(blocks), dim3(320), 0, 0, src, rows, out); });
    printf("  prefetch 16 ahead  %7.3f ms = %6.0f GB/s\n", b16, moved / b16 * 1e-6);
    printf("C. memory side of the degradation kernel alone (GB/s of UNIQUE image bytes = planes x 4 MiB)\n");
    for (int planes : {160, 48}) {
        const int cr = planes == 160 ? 64 : 16;
        struct { int flags; const char* what; } V[] = {
            {0, "partition, aligned 2 KB segments"}, {1, "+ 9 halo rows per chunk"}, {1 | 2, "+ misaligned 2112-byte segments (3 DMA instr)"},
            {1 | 2 | 4, "+ odd chunks bottom-up"}, {1 | 2 | 4 | 8, "+ XCD-grouped units (= degrade)"}, {1 | 4 | 8, "aligned segments, halo rows, mirrored, XCD"},
            {1 | 4 | 8 | 16, "whole-row strips (4 KB, 4 instr), halo rows, mirrored, XCD"}};
        for (auto& v : V) {
            CParams cp;
            cp.planes = planes; cp.chunk_rows = cr; cp.flags = v.flags;
            cp.n_units = planes * ((v.flags & 16) ? 1 : 2) * (1024 / cr);
            cp.n_blocks = (cp.n_units + 3) / 4;
            const float t = time_ms([&] { hipLaunchKernelGGL(image_stream_kernel, dim3(cp.n_blocks), dim3(256), 0, 0, src, cp, out); }, 5);
            printf("  %3d planes, %2d-row chunks: %-62s %7.1f us = %6.0f GB/s\n", planes, cr, v.what, t * 1e3, planes * 4194304.0 / t * 1e-6);
            fflush(stdout);
        }
    }
    return 0;
}
