// Microbenchmark of the pooled-vector FC kernels (csrc/k_layers.h): fc2_kernel (one launch per FC pair) against two fc_kernel launches,
// on the Student's ten pairs, 256 faces.  TOOL, not product (see tools/ub_sepup.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "pf_common.h"
#include "k_layers.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
template <typename F> static float time_us(F&& launch, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1000.f;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const int shapes[][3] = {{72, 24, 72}, {120, 32, 120}, {480, 120, 480}, {672, 168, 672}, {960, 240, 960}, {160, 64, 256}, {256, 64, 256}};
    float tot2 = 0, tot1 = 0;
    const int mult[] = {1, 2, 1, 2, 2, 1, 1};
    for (int si = 0; si < 7; ++si) {
        const int K = shapes[si][0], R = shapes[si][1], N = shapes[si][2];
        unsigned seed = 7;
        std::vector<float> x((size_t)B * K), w1((size_t)K * R), w2((size_t)R * N), b1(R), b2(N);
        for (auto& v : x) v = frand(seed); for (auto& v : w1) v = frand(seed) * 0.1f; for (auto& v : w2) v = frand(seed) * 0.1f;
        for (auto& v : b1) v = frand(seed); for (auto& v : b2) v = frand(seed);
        float *dx, *dw1, *dw2, *db1, *db2, *dh, *dy, *dy2;
        CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dw1, w1.size() * 4)); CK(hipMalloc(&dw2, w2.size() * 4)); CK(hipMalloc(&db1, R * 4)); CK(hipMalloc(&db2, N * 4));
        CK(hipMalloc(&dh, (size_t)B * R * 4)); CK(hipMalloc(&dy, (size_t)B * N * 4)); CK(hipMalloc(&dy2, (size_t)B * N * 4));
        CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, b1.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), N * 4, hipMemcpyHostToDevice));
        Fc2Args a{}; a.x = dx; a.w1 = dw1; a.b1 = db1; a.w2 = dw2; a.b2 = db2; a.y = dy2; a.B = B; a.K = K; a.R = R; a.N = N; a.act1 = PF_ACT_RELU; a.act1b = 0; a.act2 = PF_ACT_HSIGMOID; a.nparts = 1; a.xscale = 1.f;
        FcArgs f1{}; f1.x = dx; f1.wt = dw1; f1.bias = db1; f1.y = dh; f1.B = B; f1.K = K; f1.N = R; f1.act = PF_ACT_RELU;
        FcArgs f2{}; f2.x = dh; f2.wt = dw2; f2.bias = db2; f2.y = dy; f2.B = B; f2.K = R; f2.N = N; f2.act = PF_ACT_HSIGMOID;
        auto pair = [&]() {
            hipLaunchKernelGGL(fc_kernel<true>, dim3((R + 63) / 64, (B + 7) / 8), dim3(256), 0, 0, f1);
            hipLaunchKernelGGL(fc_kernel<true>, dim3((N + 63) / 64, (B + 7) / 8), dim3(256), 0, 0, f2);
        };
        auto fused = [&]() { hipLaunchKernelGGL(fc2_kernel, dim3((B + PF_FC2_FB - 1) / PF_FC2_FB), dim3(1024), 0, 0, a); };
        pair(); fused(); CK(hipDeviceSynchronize());
        std::vector<float> y((size_t)B * N), y2((size_t)B * N);
        CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y2.data(), dy2, y.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0; for (size_t i = 0; i < y.size(); ++i) worst = std::max(worst, (double)fabsf(y[i] - y2[i]));
        const float t1 = time_us(pair, 50), t2 = time_us(fused, 50);
        printf("%4d -> %3d -> %4d : two fc launches %6.2f us, fc2 %6.2f us, max |d| %.2g\n", K, R, N, t1, t2, worst);
        tot1 += mult[si] * t1; tot2 += mult[si] * t2;
    }
    printf("Student forward (10 pairs): %.1f us as 20 launches, %.1f us as 10\n", tot1, tot2);
    return 0;
}
