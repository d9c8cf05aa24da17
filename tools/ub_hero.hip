// Microbenchmark of the hero conv (csrc/k_hero.h: 3x3, 128 -> 128 on 64 x 64 maps): the shipped three-product instance against the
// opt-in one-product one (ONEPROD), timing + a sanity comparison of the outputs (one product: ~1e-3 of the output range per element on
// random data).  TOOL, not product (see tools/ub_sepup.hip; build: tools/build_ub.sh ub_hero).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "k_hero.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = 64, W = 64, C = 128, N = 128;
    unsigned seed = 3;
    std::vector<float> x((size_t)B * H * W * C), bias(N);
    for (auto& v : x) v = frand(seed) * 4.f;
    for (auto& v : bias) v = frand(seed);
    // weights [N][36 K steps (tap * 4 + chunk)][hi 32 | lo 32] f16: a genuine split of random f32 weights scaled to ~2^13
    std::vector<_Float16> wt((size_t)N * 36 * 64);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < 36; ++k)
            for (int e = 0; e < 32; ++e) {
                const float w = frand(seed) * 16384.f;
                const _Float16 hi = (_Float16)w;
                wt[((size_t)n * 36 + k) * 64 + e] = hi;
                wt[((size_t)n * 36 + k) * 64 + 32 + e] = (_Float16)(w - (float)hi);
            }
    float *dx, *db, *do3, *do1; _Float16* dw;
    const size_t oe = (size_t)B * H * W * N;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&do3, oe * 4)); CK(hipMalloc(&do1, oe * 4)); CK(hipMalloc(&dw, wt.size() * 2));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, wt.data(), wt.size() * 2, hipMemcpyHostToDevice));
    ConvGemmArgs a{};
    a.in = dx; a.wt = dw; a.bias = db; a.out = do3; a.B = B; a.inH = H; a.inW = W; a.inC = C; a.inLd = C; a.outH = H; a.outW = W; a.N = N; a.Npad = N; a.outLd = N; a.outCs = 1;
    a.outCpad = N; a.KH = a.KW = 3; a.stride = 1; a.pad = 1; a.dil = 1; a.Cpad = C; a.act = PF_ACT_RELU; a.store_out = 1; a.acc_scale = 1.f / 16384.f;
    const dim3 grid((unsigned)((size_t)B * H * W / 128));
    auto run = [&](const char* name, auto kernel, float* out) {
        a.out = out;
        auto launch = [&]() { hipLaunchKernelGGL(kernel, grid, dim3(512), 0, 0, a); };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10);
        }
        printf("  %-40s %.4f ms per %d faces\n", name, best, B);
    };
    run("three products (shipped)", conv3x3_hero_kernel<4, true, false>, do3);
    run("ONE product", conv3x3_hero_kernel<4, true, true>, do1);
    std::vector<float> o3(1 << 20), o1(1 << 20);
    CK(hipMemcpy(o3.data(), do3, o3.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), do1, o1.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < o3.size(); ++i) { worst = std::max(worst, (double)fabsf(o3[i] - o1[i])); scale = std::max(scale, (double)fabsf(o3[i])); }
    printf("  max |one - three| %.3g of range %.3g (%.2g)\n", worst, scale, worst / scale);
    return 0;
}
