// Microbenchmark of lm_front2_kernel (csrc/k_front2.h: conv_stem + blocks.0.0 in one launch), timing only -- parity is the engine's
// tests.  For reference the two launches it replaces take 0.094 + 0.160 ms per 256 crops (profiles/r06_run1_kernel_table.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifndef FRONT2_ABL
#define FRONT2_ABL 0
#endif
#include "k_front2.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = 256, W = 256, OH = 128, OW = 128;
    std::vector<unsigned char> img((size_t)B * H * W * 3);
    unsigned s = 1; for (auto& v : img) { s = s * 1664525u + 1013904223u; v = (unsigned char)(s >> 24); }
    std::vector<_Float16> ws(16 * 64); for (auto& v : ws) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xff) / 256.0f - 0.5f); }
    std::vector<float> f(16 * 16 + 9 * 16 + 64); for (auto& v : f) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    unsigned char* d_img; _Float16* d_ws; float *d_f, *d_out;
    CK(hipMalloc(&d_img, img.size())); CK(hipMalloc(&d_ws, ws.size() * 2)); CK(hipMalloc(&d_f, f.size() * 4)); CK(hipMalloc(&d_out, (size_t)B * OH * OW * 16 * 4));
    CK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_ws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_f, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    Front2Args a{};
    a.in = d_img; a.out = d_out; a.w_u8 = (const pf_half*)d_ws; a.w_f32 = (const pf_half*)d_ws; a.b_stem = d_f; a.s_u8 = 1.f / 255; a.s_f32 = 1.f;
    a.w_dw = d_f + 16; a.b_dw = d_f + 16 + 144; a.w_pw = d_f + 16 + 160; a.b_pw = d_f + 32 + 160 + 256 - 16;
    a.B = B; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.outLd = 16; a.act_stem = PF_ACT_HSWISH; a.tilesX = OW / 32; a.range_slot = nullptr;
    const dim3 grid(a.tilesX * (OH / 8), B);
    auto launch = [&]() { hipLaunchKernelGGL(lm_front2_kernel<false>, grid, dim3(256), 0, 0, a); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10);
    }
    printf("lm_front2_kernel: %d crops, %d workgroups: %.4f ms (writes %.0f MB: %.2f TB/s)\n", B, grid.x * grid.y, best, B * OH * OW * 64 / 1e6, B * OH * OW * 64.0 / best / 1e9);
    return 0;
}
