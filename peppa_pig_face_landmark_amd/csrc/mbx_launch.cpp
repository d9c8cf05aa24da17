// Translation unit of the input-stationary inverted-residual block kernel (k_mbx.h).  It MUST be built with -fno-slp-vectorize
// (peppa_pig_face_landmark_amd/build.py now passes it to every source of the library -- the other VALU-heavy kernels gain 5-13 % from
// it too, profiles/r05_run19_noslp_whole_library.txt -- but here it is a requirement): hipcc's SLP vectoriser packs the
// depthwise taps of a thread's channel pair into v_pk_fma_f32 and shuffles registers to feed them -- 200 packed fma + 210 v_mov per
// thread and tile where 400 scalar v_fmac do (a packed f32 fma costs two scalar ones on gfx950: MI355X_MICROARCH.md calls it an
// anti-lever; profiles/r05_run17_mbx_ab_packed_taps.txt).  Breaking the pairs with asm statements inside the fma stream is NOT an
// option: with ~400 of them per phase the results differed from run to run on MI355X (the hazard recogniser does not see through asm
// statements; profiles/r05_run5_mbx_determinism.txt, r05_run6).
#include "k_mbx.h"

int pf_mbx_launch(const MbxArgs& a, int nw, int KS, int Cout, int K, int dil, int mode, int grid, hipStream_t stream) {
    const bool proj = mode == 0 || mode == 2;
#define PF_MBX_CASE(NW_, KS_, NTO_, K_, DIL_, MODE_)                                                                         \
    if (nw == NW_ && KS == KS_ && K == K_ && dil == DIL_ && mode == MODE_ && (!proj || Cout == 16 * NTO_)) {                 \
        hipLaunchKernelGGL((mbx_kernel<NW_, KS_, NTO_, K_, DIL_, MODE_>), dim3(grid), dim3(NW_ * 64), 0, stream, a);         \
        return (int)hipGetLastError();                                                                                       \
    }
    // blocks 3.1 - 3.3 (80 -> 200 / 184 -> 80, no SE): one launch
    PF_MBX_CASE(16, 3, 5, 3, 1, 0) PF_MBX_CASE(8, 3, 5, 3, 1, 0)
    // squeeze passes (16 waves: no accumulators to hold), with (3) or without (1) the activated map stored for the layer-wise projection
    PF_MBX_CASE(16, 3, 7, 3, 1, 1) PF_MBX_CASE(16, 4, 7, 3, 1, 1) PF_MBX_CASE(16, 4, 10, 5, 1, 1) PF_MBX_CASE(16, 5, 10, 5, 2, 1)
    PF_MBX_CASE(16, 3, 7, 3, 1, 3) PF_MBX_CASE(16, 4, 7, 3, 1, 3) PF_MBX_CASE(16, 4, 10, 5, 1, 3) PF_MBX_CASE(16, 5, 10, 5, 2, 3)
    // recompute + gate + project: block 4.0 (80 -> 480 -> 112), 4.1 (112 -> 672 -> 112); 5.0 (112 -> 672 -> 160) and 5.1 / 5.2
    // (160 -> 960 -> 160) only fit 8 waves x 256 registers
    PF_MBX_CASE(16, 3, 7, 3, 1, 2) PF_MBX_CASE(16, 4, 7, 3, 1, 2) PF_MBX_CASE(8, 3, 7, 3, 1, 2) PF_MBX_CASE(8, 4, 7, 3, 1, 2)
    PF_MBX_CASE(8, 4, 10, 5, 1, 2) PF_MBX_CASE(8, 5, 10, 5, 2, 2)
#undef PF_MBX_CASE
    return -1;
}
