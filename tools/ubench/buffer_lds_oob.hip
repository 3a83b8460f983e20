// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS to its LDS slot (gfx950)?  LDS is pre-filled with 7.0f.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/buffer_lds_oob.hip -o /tmp/buffer_lds_oob && /tmp/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const float* a, float* o) {
    __shared__ __attribute__((aligned(16))) float l[256];
    for (int i = threadIdx.x; i < 256; i += 64) l[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 512, 0x00020000);   // 512 bytes = 32 lanes x 16 B
    const unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;                       // odd lanes: far out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)l, 16, off, 0, 0, 0);                 // lanes 32..63 (even): past num_records
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) o[i] = l[i];
}
int main() {
    float h[256], *a, *o;
    for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
    hipMalloc(&a, 4096); hipMalloc(&o, 1024);
    hipMemcpy(a, h, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, o);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane += 1)
        if (lane < 6 || lane == 31 || lane == 32 || lane == 33 || lane == 62 || lane == 63)
            printf("lane %2d: LDS slot = %g %g %g %g\n", lane, h[4 * lane], h[4 * lane + 1], h[4 * lane + 2], h[4 * lane + 3]);
    return 0;
}
