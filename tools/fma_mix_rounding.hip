// Does v_fma_mixlo_f16(x, r, 0) round once or twice?  (round 5; profiles/r05_run57_fma_mix_rounding.txt)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/fma_mix_rounding.hip -o r_mix          (the compiler selects v_fma_mixlo_f16)
//   hipcc ... -Xclang -target-feature -Xclang -fma-mix-insts tools/fma_mix_rounding.hip -o r_ref    (v_mul_f32 + v_cvt_f16_f32)
// On MI355X the mixed instruction rounds the EXACT product to f16 once; f16(f32(x * r)) rounds twice: 286 of 2^22 inputs differ.  A
// kernel in which the compiler computes "the same" (half)(x * r) once each way -- the stored high half by v_mul + v_cvt_pk, the one
// the low half is taken against by the mixed instruction -- gets a split that is off by an f16 ulp for those inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
__global__ void k(const float* xa, const float* ra, _Float16* hi, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hi[i] = (_Float16)(xa[i] * ra[i]);
}
int main(int argc, char** argv) {
    const int n = 1 << 22;
    std::vector<float> hx(n), hr(n);
    srand(7);
    for (int i = 0; i < n; ++i) { hx[i] = ldexpf(1.f + rand() / (float)RAND_MAX, -3 + rand() % 8) * ((rand() & 1) ? 1.f : -1.f); hr[i] = 0.25f + 0.75f * (rand() / (float)RAND_MAX); }
    float *dx, *dr; _Float16* dh;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dr, n * 4); (void)hipMalloc(&dh, n * 2);
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dr, hr.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dr, dh, n);
    std::vector<unsigned short> hh(n);
    (void)hipMemcpy(hh.data(), dh, n * 2, hipMemcpyDeviceToHost);
    // reference on the host: the f32 product rounded to f16 (double rounding), and the exact product rounded once
    int ne_double = 0, ne_single = 0;
    for (int i = 0; i < n; ++i) {
        const float p32 = hx[i] * hr[i];
        const _Float16 d = (_Float16)p32;
        const _Float16 s = (_Float16)((double)hx[i] * (double)hr[i]);
        unsigned short ud, us; __builtin_memcpy(&ud, &d, 2); __builtin_memcpy(&us, &s, 2);
        if (hh[i] != ud) ++ne_double;
        if (hh[i] != us) ++ne_single;
    }
    printf("%s: f16(x * r) on the GPU differs from f16(f32(x * r)) [two roundings] in %d of %d, from f16(exact x * r) [one rounding] in %d\n", argv[1], ne_double, n, ne_single);
    return 0;
}
