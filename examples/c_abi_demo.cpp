// Torch-free use of the C ABI (include/kmsr.h): plain HIP allocations, raw pointers, a HIP stream.
// Runs the V-A degradation (C_30apply_kernel_to_landsat.py:68-124 semantics) on a synthetic 5 x 96 x 80 patch and checks
// it against a straightforward host loop written in the reference's order of operations (normalise, replicate pad,
// full-resolution cross-correlation, cascade of 2x2 average pools).  Build: make -C kernel-modeling-super-resolution_amd/csrc demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/kmsr.h"

#define CHECK_HIP(e)                                                                     \
    do {                                                                                 \
        hipError_t _e = (e);                                                             \
        if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } \
    } while (0)

int main() {
    const int C = 5, H = 96, W = 80, K = 13, f = 4, P = K / 2;
    std::vector<float> img((size_t)C * H * W), ker((size_t)C * K * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto& v : img) v = 20.f + 80.f * rnd();
    for (auto& v : ker) v = 0.01f + rnd();

    // ---- host restatement (reference order)
    const int s_levels = 2, F = 4, h = H / F, w = W / F;
    std::vector<float> want((size_t)C * h * w);
    for (int c = 0; c < C; ++c) {
        float sum = 0.f;
        for (int i = 0; i < K * K; ++i) sum += ker[(size_t)c * K * K + i];
        std::vector<float> kn(K * K);
        for (int i = 0; i < K * K; ++i) kn[i] = sum > 0.f ? ker[(size_t)c * K * K + i] / sum : ker[(size_t)c * K * K + i];
        std::vector<float> blur((size_t)H * W);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float acc = 0.f;
                for (int u = 0; u < K; ++u)
                    for (int v = 0; v < K; ++v) {
                        int yy = y + u - P, xx = x + v - P;
                        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                        acc += kn[u * K + v] * img[((size_t)c * H + yy) * W + xx];
                    }
                blur[(size_t)y * W + x] = acc;
            }
        int ch = H, cw = W;
        for (int l = 0; l < s_levels; ++l) {
            const int nh = ch / 2, nw = cw / 2;
            std::vector<float> nxt((size_t)nh * nw);
            for (int y = 0; y < nh; ++y)
                for (int x = 0; x < nw; ++x)
                    nxt[(size_t)y * nw + x] = 0.25f * (blur[(size_t)(2 * y) * cw + 2 * x] + blur[(size_t)(2 * y) * cw + 2 * x + 1] +
                                                       blur[(size_t)(2 * y + 1) * cw + 2 * x] + blur[(size_t)(2 * y + 1) * cw + 2 * x + 1]);
            blur.swap(nxt);
            ch = nh; cw = nw;
        }
        for (int i = 0; i < h * w; ++i) want[(size_t)c * h * w + i] = blur[i];
    }

    // ---- the library
    if (kmsr_device_count() < 1) { fprintf(stderr, "no GPU: %s\n", kmsr_last_error()); return 3; }
    kmsr_degrade_desc d = {};
    d.N = 1; d.C = C; d.H = H; d.W = W;
    d.hr_stride_n = (int64_t)C * H * W; d.hr_stride_c = (int64_t)H * W; d.hr_stride_h = W;
    d.n_kernels = 1; d.kh = K; d.kw = K; d.factor = f;
    d.down_mode = KMSR_DOWN_AVGPOOL; d.pad_mode = KMSR_PAD_REPLICATE; d.normalize = 1; d.noise_mode = KMSR_NOISE_NONE;
    int32_t oh = 0, ow = 0;
    if (kmsr_degrade_out_shape(&d, &oh, &ow) != KMSR_OK || oh != h || ow != w) { fprintf(stderr, "out shape: %s\n", kmsr_last_error()); return 4; }
    const size_t ws_bytes = kmsr_degrade_workspace_bytes(&d);
    float *d_img = nullptr, *d_ker = nullptr, *d_out = nullptr;
    void* d_ws = nullptr;
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc(&d_img, img.size() * 4));
    CHECK_HIP(hipMalloc(&d_ker, ker.size() * 4));
    CHECK_HIP(hipMalloc(&d_out, want.size() * 4));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 4));
    CHECK_HIP(hipMemcpyAsync(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(d_ker, ker.data(), ker.size() * 4, hipMemcpyHostToDevice, stream));
    const int rc = kmsr_degrade_f32(&d, d_img, d_ker, nullptr, nullptr, nullptr, nullptr, d_out, d_ws, ws_bytes, stream);
    if (rc != KMSR_OK) { fprintf(stderr, "kmsr_degrade_f32 -> %d: %s\n", rc, kmsr_last_error()); return 5; }
    std::vector<float> got(want.size());
    CHECK_HIP(hipMemcpyAsync(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    double worst = 0.0;
    for (size_t i = 0; i < got.size(); ++i) worst = std::fmax(worst, std::fabs((double)got[i] - want[i]) / std::fabs((double)want[i]));
    // error convention: a bad argument is reported as a code + message, never an abort
    d.kh = 12;
    const int rc_bad = kmsr_degrade_f32(&d, d_img, d_ker, nullptr, nullptr, nullptr, nullptr, d_out, d_ws, ws_bytes, stream);
    printf("kmsr %d: V-A [%d,%d,%d] f=%d -> [%d,%d,%d], max rel err vs host loop %.3e; bad descriptor -> rc %d (\"%s\")\n",
           kmsr_version(), C, H, W, f, C, oh, ow, worst, rc_bad, kmsr_last_error());
    (void)hipFree(d_img); (void)hipFree(d_ker); (void)hipFree(d_out); (void)hipFree(d_ws); (void)hipStreamDestroy(stream);
    return worst < 1e-5 ? 0 : 1;
}
