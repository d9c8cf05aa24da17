// Microbenchmark of the decoder front end (csrc/k_sepup.h) outside the engine: one translation unit, ~40 s to build, so that
// structural variants of sepup_pipe_kernel can be timed against the shipped instance in one short GPU session
// (tools/gpu_ub_sepup.sh).  Random inputs of the Student's two shapes (up2: 256 + 24 -> 128 at 64 x 64, up1: 256 + 40 -> 256 at
// 32 x 32); every variant's output is compared with the shipped instance's BIT FOR BIT (the variants change who issues what
// and when, never the arithmetic).  TOOL, not product: nothing in the library or the tests depends on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#include "k_sepup.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

template <typename F> static float time_ms(F&& launch, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct Shape { int W, C1, C2, N; };
static int g_dbg = 0;      // -DPF_ABLATE=1 builds: SepupArgs::dbg (1 no weight refresh, 2 no patch refresh, 4 no patch reads, 8 no MFMAs, 16 no stores)

template <int BN, int W, typename L>
static void run_shape(const char* name, const Shape& sh, int B, L&& variants) {
    const int H = sh.W, loH = H / 2, loW = sh.W / 2;
    const int C = sh.C1 + sh.C2, Cpad = (C + 31) / 32 * 32, NK = Cpad / 32, nskip = NK - sh.C1 / 32;
    const int skipLd = (sh.C2 + 3) / 4 * 4;
    unsigned seed = 12345;
    std::vector<float> lo((size_t)B * loH * loW * sh.C1), skip((size_t)B * H * sh.W * skipLd), dwlo(9 * sh.C1), dw2(9 * sh.C2), bias(sh.N);
    for (auto& v : lo) v = frand(seed) * 4.f;
    for (auto& v : skip) v = frand(seed) * 4.f;
    for (auto& v : dwlo) v = frand(seed);
    for (auto& v : dw2) v = frand(seed);
    for (auto& v : bias) v = frand(seed);
    std::vector<_Float16> wt((size_t)sh.N * NK * 64);           // [N][NK][hi 32 | lo 32]
    for (size_t i = 0; i < wt.size(); ++i)                       // a genuine split: hi plane O(0.1), lo plane 2^-12 of that (random lo
        wt[i] = (_Float16)(frand(seed) * 0.25f * (((i >> 5) & 1) ? 0.000244f : 1.0f));      // halves as large as hi would make the dropped lo x lo term visible)
    // VCOL filters: V[cls][j][kx][c] = sum_ky A_cls[ky][j] w[ky][kx][c]; classes first / last / even / odd (k_sepup.h pf_pos_class)
    std::vector<float> dwv((size_t)4 * 9 * sh.C1);
    {
        const double Ae[3][3] = {{0.75, 0.25, 0}, {0.25, 0.75, 0}, {0, 0.75, 0.25}}, Ao[3][3] = {{0.25, 0.75, 0}, {0, 0.75, 0.25}, {0, 0.25, 0.75}};
        for (int cls = 0; cls < 4; ++cls)
            for (int j = 0; j < 3; ++j)
                for (int kx = 0; kx < 3; ++kx)
                    for (int c = 0; c < sh.C1; ++c) {
                        double v = 0;
                        for (int ky = 0; ky < 3; ++ky) {
                            double aw = (cls == 0 || cls == 2) ? Ae[ky][j] : Ao[ky][j];
                            if ((cls == 0 && ky == 0) || (cls == 1 && ky == 2)) aw = 0;
                            v += aw * (double)dwlo[(ky * 3 + kx) * sh.C1 + c];
                        }
                        dwv[((size_t)cls * 9 + j * 3 + kx) * sh.C1 + c] = (float)v;
                    }
    }
    float *d_lo, *d_skip, *d_dwlo, *d_dw2, *d_bias, *d_out, *d_ref, *d_dwv; unsigned char *d_wt, *d_skipx;
    CK(hipMalloc(&d_dwv, dwv.size() * 4)); CK(hipMemcpy(d_dwv, dwv.data(), dwv.size() * 4, hipMemcpyHostToDevice));
    const size_t out_elems = (size_t)B * H * sh.W * sh.N;
    const int tpf = H * sh.W / 128;
    CK(hipMalloc(&d_lo, lo.size() * 4)); CK(hipMalloc(&d_skip, skip.size() * 4)); CK(hipMalloc(&d_dwlo, dwlo.size() * 4));
    CK(hipMalloc(&d_dw2, dw2.size() * 4)); CK(hipMalloc(&d_bias, bias.size() * 4)); CK(hipMalloc(&d_out, out_elems * 4)); CK(hipMalloc(&d_ref, out_elems * 4));
    CK(hipMalloc(&d_wt, wt.size() * 2)); CK(hipMalloc(&d_skipx, (size_t)B * tpf * nskip * 16384));
    CK(hipMemcpy(d_lo, lo.data(), lo.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_skip, skip.data(), skip.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dwlo, dwlo.data(), dwlo.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_dw2, dw2.data(), dw2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wt, wt.data(), wt.size() * 2, hipMemcpyHostToDevice));
    SepupArgs s{};
    s.lo = d_lo; s.skip = d_skip; s.out = d_ref; s.dw_lo = d_dwlo; s.dw_v = d_dwv; s.dw_w2 = d_dw2; s.wt = d_wt; s.bias = d_bias; s.skipx = d_skipx;
    s.B = B; s.H = H; s.C1 = sh.C1; s.C2 = sh.C2; s.loLd = sh.C1; s.skipLd = skipLd; s.outLd = sh.N;
    s.N = sh.N; s.Cpad = Cpad; s.act = PF_ACT_RELU; s.acc_scale = 1.0f; s.range_slot = nullptr; s.prof = nullptr; s.dbg = g_dbg;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int per_xcd = ((B + 7) / 8) * tpf;
    const int wgs = 8 * std::min(cus / 8, per_xcd);
    auto skipk = [&]() {
        if (sh.C2 <= 32) hipLaunchKernelGGL((sepup_skip_kernel<W, 32>), dim3(B * tpf), dim3(512), 0, 0, s);
        else hipLaunchKernelGGL((sepup_skip_kernel<W, 64>), dim3(B * tpf), dim3(512), 0, 0, s);
    };
    skipk();
    CK(hipDeviceSynchronize());
    printf("%s: B %d, %d workgroups, %d K steps per tile; skip kernel %.4f ms\n", name, B, wgs, NK, time_ms(skipk, 10));
    std::vector<float> ref(out_elems), got(out_elems);
    bool have_ref = false;
    variants([&](const char* vname, auto kernel) {
        s.out = have_ref ? d_out : d_ref;
        CK(hipMemset(s.out, 0xff, out_elems * 4));
        auto launch = [&]() { hipLaunchKernelGGL(kernel, dim3(wgs), dim3(1024), 0, 0, s); };
        launch();
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        CK(hipMemcpy(have_ref ? got.data() : ref.data(), s.out, out_elems * 4, hipMemcpyDeviceToHost));
        const char* verdict = "reference";
        char vb[256];
        if (have_ref) {
            if (memcmp(got.data(), ref.data(), out_elems * 4) == 0) verdict = "bit-identical";
            else {
                double worst = 0, scale = 0;
                for (size_t i = 0; i < out_elems; ++i) { worst = std::max(worst, (double)fabsf(got[i] - ref[i])); scale = std::max(scale, (double)fabsf(ref[i])); }
                double sum = 0; size_t big = 0, firstbig = 0;
                for (size_t i = 0; i < out_elems; ++i) { const double d = fabsf(got[i] - ref[i]); sum += d; if (d > 1e-5) { if (!big) firstbig = i; ++big; } }
                const size_t px = firstbig / sh.N;
                snprintf(vb, sizeof(vb), "differs: max |d| %.3g of range %.3g, mean %.2g, %zu > 1e-5 (first: face %zu y %zu x %zu n %zu)", worst, scale, sum / out_elems, big,
                         px / (H * sh.W), (px / sh.W) % H, px % sh.W, firstbig % sh.N);
                verdict = vb;
            }
        }
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) best = std::min(best, time_ms(launch, 10));
        printf("  %-44s %.4f ms  %s\n", vname, best, verdict);
        have_ref = true;
    });
    for (void* q : {(void*)d_lo, (void*)d_skip, (void*)d_dwlo, (void*)d_dw2, (void*)d_bias, (void*)d_out, (void*)d_ref, (void*)d_wt, (void*)d_skipx}) (void)hipFree(q);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    g_dbg = argc > 2 ? atoi(argv[2]) : 0;
    if (g_dbg) printf("ABLATION dbg = %d (results are wrong by construction)\n", g_dbg);
    run_shape<128, 64>("up2 (280 -> 128 at 64 x 64)", Shape{64, 256, 24, 128}, B, [&](auto run) {
        run("shipped <128,64,D=3,W by consumers,defer>", sepup_pipe_kernel<128, 64, 3, false, true>);
        run("patch requests by consumers", sepup_pipe_kernel<128, 64, 3, false, true, true, false>);
        run("D=4 (bias in registers)", sepup_pipe_kernel<128, 64, 4, false, true, false, true>);
        run("D=4 + patch requests by consumers", sepup_pipe_kernel<128, 64, 4, false, true, true, true>);
        run("D=3, bias in registers only", sepup_pipe_kernel<128, 64, 3, false, true, false, true>);
        run("weights by producers", sepup_pipe_kernel<128, 64, 3, true, true>);
        run("no deferred stores", sepup_pipe_kernel<128, 64, 3, false, false>);
        run("VCOL", sepup_pipe_kernel<128, 64, 3, false, true, false, false, true>);
        run("VCOL + D=4", sepup_pipe_kernel<128, 64, 4, false, true, false, true, true>);
    });
    run_shape<256, 32>("up1 (296 -> 256 at 32 x 32)", Shape{32, 256, 40, 256}, B, [&](auto run) {
        run("shipped <256,32,D=2,W by producers>", sepup_pipe_kernel<256, 32, 2, true, false>);
        run("weights by consumers", sepup_pipe_kernel<256, 32, 2, false, false>);
        run("patch requests by consumers", sepup_pipe_kernel<256, 32, 2, true, false, true, false>);
        run("all requests by consumers", sepup_pipe_kernel<256, 32, 2, false, false, true, false>);
        run("VCOL", sepup_pipe_kernel<256, 32, 2, true, false, false, false, true>);
    });
    return 0;
}
