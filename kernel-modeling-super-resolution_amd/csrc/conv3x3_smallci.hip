// 3x3 convolution with FEW INPUT channels (Cin <= 8: the SR head bands -> F, and the data gradient of the tail F -> bands),
// stride 1, zero padding 1, bias / ReLU epilogue (semantics: include/kmsr.h, kmsr_conv3x3_smallci_fwd_f32).
//
// These layers write 64 channels per pixel from 3..8 channels read: they are bound by the output write and by the matrix pipe
// in equal parts, so the kernel is built to pay for nothing else:
//   * persistent workgroups (2 per CU) walk (image, cout group, 8 x 64-pixel tile) items; the layer's weights live in REGISTERS
//     for the whole kernel (A operands of v_mfma_f32_32x32x2_f32: lane (co = l & 31, k = l >> 5) holds w[co][2q + k][tap] for the
//     9 x ceil(Cin/2) (tap, channel pair) steps of both 32-channel blocks) -- no weight traffic, no packing pass;
//   * K = (tap, channel pair): Cin is padded to even (5 -> 6: 83 % useful MFMA work; the 8-channel chunks of the general
//     kernel give 62 %); B operands are ds_read_b32 with compile-time offsets from one lane base (pixel + k * plane);
//   * the input halo tile (Cin planes x 10 x 66) is double buffered in LDS; the next tile arrives by 16-byte buffer loads (range
//     check = zero padding) issued before the products, written after them; one barrier per tile;
//   * stores are buffer stores on one descriptor per (image, cout group): 32 consecutive pixels of one channel per half wave.
#include "kmsr_common.h"

#include <algorithm>

namespace kmsr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int QTH = 8, QTW = 64;                 // output tile
constexpr int QROWS = QTH + 2;                   // 10 input rows
constexpr int QPITCH = 72;                       // row pitch: halo column -1 at LDS column 3, interior at 4 .. 67 (16-byte aligned), +64 at 68
constexpr int QPLANE = QROWS * QPITCH + 16;      // 736 floats = 32 mod 64 banks: the two k halves of a B read use disjoint banks
constexpr int QQ = QTW / 4;                      // 16 interior 16-byte pieces per row

struct SmallCiParams {
    int N, Cin, H, W, Cout;
    int tiles_x, tiles, cgs, n_items, relu;
};

// UNSH (round 4): the output is stored PIXEL-UNSHUFFLED -- out[n, 4 co + 2 (y & 1) + (x & 1), y >> 1, x >> 1] -- and every lane keeps running
// sums of what it stores per (column parity, channel): the data gradient of the SR tail then arrives in the layout the F -> 4F layer below
// wants as its output gradient, with that layer's bias gradient, and no pass re-reads the HR-sized map (kmsr_conv3x3_smallci_unshuffle_fwd_f32).
// The MFMA's N index is free to name any 32 pixels: here lane l32 of column half ch multiplies pixel x = 2 l32 + ch (the B operand is a
// stride-2 ds_read_b32: 32 distinct banks), so one accumulator register holds the 32 pixels of ONE sub-pixel plane at consecutive
// positions and an un-shuffled store is a full 128-byte line, like the plain store (the first form kept lane = x and wrote every line as two
// 64-byte halves from different instructions: 2.5 ms slower per training step than a separate un-shuffle pass).  Wave wv owns rows wv and
// wv + 4 of the tile, so the row parity is the wave's and the sums need the same 64 registers as before.
// NCB (round 4): 32-channel output blocks per item.  Cout <= 32 (the KernelGAN condition encoder's 5 -> 32 layer: 4.3 GB of output per 512
// patches) runs ONE block -- with two, half of the MFMAs multiplied zero weights and the layer was bound by that (2.2 ms where the write takes 1.1).
// SUMS: the un-shuffled form also keeps the per-lane channel sums (64 more live registers: with them the 2-workgroups-per-CU register budget
// spills 188 VGPRs and the layer takes 4.7 instead of 2.5 ms at config 3 -- the caller that can get the sums elsewhere passes chan_sums = NULL).
template <int CP2, bool UNSH, int NCB = 2, bool SUMS = UNSH>      // channel pairs: Cin padded to 2 * CP2
__global__ void __launch_bounds__(256, 2) conv3x3_smallci_kernel(const float* __restrict__ xg, const float* __restrict__ wg,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 float* __restrict__ usum, SmallCiParams p) {
    constexpr int CP = 2 * CP2, TILE_F = CP * QPLANE;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_F + 64];
    float* sbias = lds + 2 * TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    const unsigned HWb = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;

    // ---- staging map: interior pieces e = tid + 256 j < CP * 10 * 16 -> (plane, row, piece); halo floats e < CP * 10 * 2
    constexpr int NPIECE = CP * QROWS * QQ, NJ = (NPIECE + 255) / 256, NHALO = CP * QROWS * 2;
    f32x4 xr[NJ];
    float xh = 0.f;
    unsigned vo[NJ], voh = OOB;
    __amdgpu_buffer_rsrc_t rx;
    auto item_of = [&](int it, int& n, int& cg, int& y0, int& x0) {
        const int tile = it % p.tiles, rest = it / p.tiles;
        cg = rest % p.cgs;
        n = rest / p.cgs;
        y0 = (tile / p.tiles_x) * QTH;
        x0 = (tile % p.tiles_x) * QTW;
    };
    auto stage_setup = [&](int it) {
        int n, cg, y0, x0;
        item_of(it, n, cg, y0, x0);
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + (long long)n * p.Cin * HW), 0, (unsigned)p.Cin * HWb, 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + 256 * j;
            const int pl = e / (QROWS * QQ), rem = e - pl * (QROWS * QQ), r = rem / QQ, q = rem - r * QQ;
            const int y = y0 - 1 + r, x = x0 + 4 * q;
            vo[j] = (e < NPIECE && y >= 0 && y < H && x < W) ? (unsigned)(y * W + x) * 4u + (unsigned)pl * HWb : OOB;   // planes >= Cin: out of range -> 0
        }
        {
            const int pl = tid / (QROWS * 2), rem = tid - pl * (QROWS * 2), r = rem >> 1, side = rem & 1;
            const int y = y0 - 1 + r, x = side ? x0 + QTW : x0 - 1;
            voh = (tid < NHALO && y >= 0 && y < H && x >= 0 && x < W) ? (unsigned)(y * W + x) * 4u + (unsigned)pl * HWb : OOB;
        }
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo[j], 0, 0));
        xh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voh, 0, 0));
    };
    auto store_stage = [&](int b) {
        float* xl = lds + b * TILE_F;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + 256 * j;
            const int pl = e / (QROWS * QQ), rem = e - pl * (QROWS * QQ), r = rem / QQ, q = rem - r * QQ;
            if (e < NPIECE) *reinterpret_cast<f32x4*>(xl + pl * QPLANE + r * QPITCH + 4 + 4 * q) = xr[j];
        }
        if (tid < NHALO) {
            const int pl = tid / (QROWS * 2), rem = tid - pl * (QROWS * 2), r = rem >> 1, side = rem & 1;
            xl[pl * QPLANE + r * QPITCH + (side ? 4 + QTW : 3)] = xh;
        }
    };

    // ---- weights -> registers, once per (workgroup, cout group); re-read when the item's cout group changes
    float areg[NCB][9 * CP2];
    int cg_loaded = -1;
    auto load_weights = [&](int cg) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int co = cg * 64 + cb * 32 + l32;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < CP2; ++q) {
                    const int ci = 2 * q + half;
                    const bool ok = co < p.Cout && ci < p.Cin;
                    const float v = wg[ok ? ((long long)co * p.Cin + ci) * 9 + t : 0];
                    areg[cb][t * CP2 + q] = ok ? v : 0.f;
                }
        }
        if (tid < 64) sbias[tid] = (bias && cg * 64 + tid < p.Cout) ? bias[cg * 64 + tid] : 0.f;
        cg_loaded = cg;
    };

    int it = blockIdx.x, b = 0;
    if (it >= p.n_items) return;                               // block-uniform
    stage_setup(it);
    issue_loads();
    {
        int n, cg, y0, x0;
        item_of(it, n, cg, y0, x0);
        load_weights(cg);
    }
    store_stage(0);
    __syncthreads();
    static_assert(!UNSH || NCB == 2, "the un-shuffled form stores both channel blocks");
    static_assert(!SUMS || UNSH, "channel sums belong to the un-shuffled form");
    float cs[SUMS ? 2 : 1][2][16];                              // UNSH: running sums per (row parity, co block, register) of this lane's stores
#pragma unroll
    for (int a = 0; a < (SUMS ? 2 : 1); ++a)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[a][cb][r] = 0.f;
    const int lbase = half * QPLANE + 3 + (UNSH ? 2 * l32 : l32);   // B operand: k = half picks the plane of the pair, pixel l32 (UNSH: 2 l32 + ch); + taps / group offsets
    for (; it < p.n_items; it += gridDim.x) {
        const int nxt = it + gridDim.x;
        const bool more = nxt < p.n_items;
        int n, cg, y0, x0;
        item_of(it, n, cg, y0, x0);
        if (more) {
            stage_setup(nxt);
            issue_loads();                                       // in flight under the products
        }
        const float* st = lds + b * TILE_F + lbase;
        const auto ro = __builtin_amdgcn_make_buffer_rsrc(out + ((long long)n * p.Cout + cg * 64) * HW, 0,
                                                          (unsigned)min(32 * NCB, p.Cout - cg * 64) * HWb, 0x00020000);
        // wave wv owns output rows 2 wv, 2 wv + 1 of the tile = 4 groups of 32 pixels; two groups are multiplied together
        // (4 independent accumulator tiles keep the 64-cycle MFMA chain busy)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {                        // gp = row of the pair
            f32x16 acc[2][NCB];                                  // [column half][co block]
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {             // accumulators start at the bias: D rows co = (r & 3) + 8 (r >> 2) + 4 half
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + cb * 32 + 8 * g4 + 4 * half);
                        acc[ch][cb][4 * g4 + 0] = b4.x; acc[ch][cb][4 * g4 + 1] = b4.y;
                        acc[ch][cb][4 * g4 + 2] = b4.z; acc[ch][cb][4 * g4 + 3] = b4.w;
                    }
            const int row = UNSH ? wv + 4 * gp : 2 * wv + gp;    // output row inside the tile; input rows row .. row + 2
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < CP2; ++q) {
                    const int off = (2 * q) * QPLANE + (row + t / 3) * QPITCH + t % 3;
                    const float b0 = st[off], b1 = st[off + (UNSH ? 1 : 32)];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[cb][t * CP2 + q], b0, acc[0][cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[1][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[cb][t * CP2 + q], b1, acc[1][cb], 0, 0, 0);
                }
            const int y = y0 + row;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int x = UNSH ? x0 + 2 * l32 + ch : x0 + 32 * ch + l32;
                const bool inimg = y < H && x < W;
                unsigned vbase;
                if constexpr (UNSH)     // the 4 sub-pixel planes of a channel occupy the channel's HW floats: same channel stride
                    vbase = inimg ? (unsigned)((2 * (y & 1) + (x & 1)) * (int)(HW >> 2) + (y >> 1) * (W >> 1) + (x >> 1)) * 4u : OOB;
                else
                    vbase = inimg ? (unsigned)(y * W + x) * 4u : OOB;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;      // channel within the group (past Cout: dropped by the range check)
                        float v = acc[ch][cb][r];
                        if (p.relu) v = fmaxf(v, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vbase + (unsigned)col * HWb, 0, 0);
                        if constexpr (SUMS) cs[ch][cb][r] += inimg ? v : 0.f;
                    }
            }
        }
        if (more) {
            store_stage(b ^ 1);                                  // the other buffer: every wave finished reading it before the last barrier
            int n2, cg2, y2, x2;
            item_of(nxt, n2, cg2, y2, x2);
            if (cg2 != cg_loaded) {                              // block-uniform; sbias / areg of the old group are no longer needed after the barrier
                __syncthreads();
                load_weights(cg2);
            }
        }
        __syncthreads();
        b ^= 1;
    }
    if constexpr (SUMS) {       // per-lane partial sums [block][wave][lane][column parity][co block][register] (row parity = wave & 1); reduced in a fixed order below
        float* o = usum + ((long long)(blockIdx.x * 4 + wv) * 64 + lane) * 64;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(o + a * 32 + cb * 16 + 4 * q) = f32x4{cs[a][cb][4 * q], cs[a][cb][4 * q + 1], cs[a][cb][4 * q + 2], cs[a][cb][4 * q + 3]};
    }
}

// sums[4 co + 2 yp + xp] = sum over blocks, the two waves of row parity yp and the 32 lanes of the channel's k half of their partials (fixed
// order: bit-reproducible)
__global__ void __launch_bounds__(256) smallci_unshuffle_sum_kernel(const float* __restrict__ usum, int nblocks, float* __restrict__ sums) {
    const int k = blockIdx.x, co = k >> 2, yp = (k >> 1) & 1, xp = k & 1;
    const int cb = co >> 5, c5 = co & 31, half = (c5 >> 2) & 1, r = (c5 & 3) + 4 * (c5 >> 3);
    const int total = nblocks * 2 * 32;                       // (block, wave of the parity, lane of the half)
    float a = 0.f;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int j = e & 31, w2 = (e >> 5) & 1, blk = e >> 6, wave = 2 * w2 + yp, lane = half * 32 + j;
        a += usum[((long long)(blk * 4 + wave) * 64 + lane) * 64 + xp * 32 + cb * 16 + r];
    }
    __shared__ float red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[k] = red[0];
}

}  // namespace
}  // namespace kmsr

using namespace kmsr;

extern "C" int kmsr_conv3x3_smallci_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* w, const float* bias,
                                            float* out, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (!x || !w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (d->stride != 1 || d->pixel_shuffle || d->flags) return fail(KMSR_E_UNSUPPORTED, "small-Cin conv: stride 1, no pixel_shuffle, no flags");
    if (d->Cin > 8) return fail(KMSR_E_UNSUPPORTED, "small-Cin conv serves Cin <= 8 (Cin=%d)", d->Cin);
    if (d->W % 4 || reinterpret_cast<uintptr_t>(x) % 16) return fail(KMSR_E_UNSUPPORTED, "small-Cin conv needs W %% 4 == 0 and a 16-byte aligned input");
    if ((long long)d->H * d->W * 4 * 64 >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "small-Cin conv: H*W exceeds 2^23 pixels per plane");
    SmallCiParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.relu = d->relu;
    p.tiles_x = ceil_div(d->W, QTW);
    p.tiles = p.tiles_x * ceil_div(d->H, QTH);
    p.cgs = ceil_div(d->Cout, 64);
    const long long items = (long long)p.tiles * p.cgs * d->N;
    if (items > (1ll << 30)) return fail(KMSR_E_UNSUPPORTED, "conv grid too large");
    p.n_items = (int)items;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)std::min<long long>(items, 2ll * cus);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (d->Cout <= 32) {        // one 32-channel block per item (cgs == 1)
        switch ((d->Cin + 1) / 2) {
            case 1: conv3x3_smallci_kernel<1, false, 1><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            case 2: conv3x3_smallci_kernel<2, false, 1><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            case 3: conv3x3_smallci_kernel<3, false, 1><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            default: conv3x3_smallci_kernel<4, false, 1><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
        }
    } else {
        switch ((d->Cin + 1) / 2) {
            case 1: conv3x3_smallci_kernel<1, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            case 2: conv3x3_smallci_kernel<2, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            case 3: conv3x3_smallci_kernel<3, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            default: conv3x3_smallci_kernel<4, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
        }
    }
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}

static int smallci_grid(const kmsr_conv3x3_desc* d) {
    const long long items = (long long)ceil_div(d->W, QTW) * ceil_div(d->H, QTH) * d->N;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return (int)std::min<long long>(items, 2ll * cus);
}

extern "C" size_t kmsr_conv3x3_smallci_unshuffle_workspace_bytes(const kmsr_conv3x3_desc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0) return 0;
    return (size_t)smallci_grid(d) * 4 * 64 * 64 * sizeof(float);
}

extern "C" int kmsr_conv3x3_smallci_unshuffle_fwd_f32(const kmsr_conv3x3_desc* d, const float* x, const float* w, const float* bias,
                                                      float* out, float* chan_sums, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!d) return fail(KMSR_E_BADARG, "null descriptor");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return fail(KMSR_E_BADARG, "bad conv shape");
    if (!x || !w || !out) return fail(KMSR_E_BADARG, "null tensor pointer");
    if (chan_sums && !workspace) return fail(KMSR_E_BADARG, "small-Cin unshuffle conv: channel sums need the workspace");
    if (d->stride != 1 || d->pixel_shuffle || d->flags) return fail(KMSR_E_UNSUPPORTED, "small-Cin conv: stride 1, no pixel_shuffle, no flags");
    if (d->Cin > 8 || d->Cout > 64) return fail(KMSR_E_UNSUPPORTED, "small-Cin unshuffle conv serves Cin <= 8, Cout <= 64 (Cin=%d, Cout=%d)", d->Cin, d->Cout);
    if (d->H % 2 || d->W % 4 || reinterpret_cast<uintptr_t>(x) % 16) return fail(KMSR_E_UNSUPPORTED, "small-Cin unshuffle conv needs even H, W %% 4 == 0 and a 16-byte aligned input");
    if ((long long)d->H * d->W * 4 * 64 >= (1ll << 31)) return fail(KMSR_E_UNSUPPORTED, "small-Cin conv: H*W exceeds 2^23 pixels per plane");
    if (chan_sums && workspace_bytes < kmsr_conv3x3_smallci_unshuffle_workspace_bytes(d)) return fail(KMSR_E_NOMEM, "small-Cin unshuffle conv: workspace too small");
    SmallCiParams p{};
    p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.relu = d->relu;
    p.tiles_x = ceil_div(d->W, QTW);
    p.tiles = p.tiles_x * ceil_div(d->H, QTH);
    p.cgs = 1;
    p.n_items = p.tiles * d->N;
    const int grid = smallci_grid(d);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* ws = static_cast<float*>(workspace);
    if (!chan_sums) {           // store only: the caller has the sums from elsewhere (sr.UpTailFunction derives them from the input's border sums)
        switch ((d->Cin + 1) / 2) {
            case 1: conv3x3_smallci_kernel<1, true, 2, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            case 2: conv3x3_smallci_kernel<2, true, 2, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            case 3: conv3x3_smallci_kernel<3, true, 2, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
            default: conv3x3_smallci_kernel<4, true, 2, false><<<grid, 256, 0, stream>>>(x, w, bias, out, nullptr, p); break;
        }
        KMSR_HIP_CHECK(hipGetLastError());
        return KMSR_OK;
    }
    switch ((d->Cin + 1) / 2) {
        case 1: conv3x3_smallci_kernel<1, true><<<grid, 256, 0, stream>>>(x, w, bias, out, ws, p); break;
        case 2: conv3x3_smallci_kernel<2, true><<<grid, 256, 0, stream>>>(x, w, bias, out, ws, p); break;
        case 3: conv3x3_smallci_kernel<3, true><<<grid, 256, 0, stream>>>(x, w, bias, out, ws, p); break;
        default: conv3x3_smallci_kernel<4, true><<<grid, 256, 0, stream>>>(x, w, bias, out, ws, p); break;
    }
    KMSR_HIP_CHECK(hipGetLastError());
    smallci_unshuffle_sum_kernel<<<4 * d->Cout, 256, 0, stream>>>(ws, grid, chan_sums);
    KMSR_HIP_CHECK(hipGetLastError());
    return KMSR_OK;
}
