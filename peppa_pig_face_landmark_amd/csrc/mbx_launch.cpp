// Translation unit of the input-stationary inverted-residual block kernel (k_mbx.h), built with -fno-slp-vectorize
// (peppa_pig_face_landmark_amd/build.py; the CPU test build of the same sources does likewise): hipcc's SLP vectoriser packs the depthwise taps of a
// thread's channel pair into v_pk_fma_f32, which runs far below two scalar v_fma_f32 on gfx950 (MI355X_MICROARCH.md: "an
// anti-lever"; ~4x per flop measured in this kernel, profiles/r05_run3_mbx_phase_cycles_first_cut.txt).  Breaking the pairs with
// asm statements inside the fma stream is NOT an option: with ~400 of them per phase the results differed from run to run on
// MI355X (the hazard recogniser does not see through asm statements; profiles/r05_run5_mbx_determinism.txt, r05_run6).
#include "k_mbx.h"

int pf_mbx_launch(const MbxArgs& a, int KS, int Cout, int K, int dil, int mode, int grid, hipStream_t stream) {
#define PF_MBX_CASE(KS_, NTO_, K_, DIL_, MODE_)                                                                  \
    if (KS == KS_ && K == K_ && dil == DIL_ && mode == MODE_ && (MODE_ == 1 || Cout == 16 * NTO_)) {             \
        hipLaunchKernelGGL((mbx_kernel<KS_, NTO_, K_, DIL_, MODE_>), dim3(grid), dim3(512), 0, stream, a);       \
        return (int)hipGetLastError();                                                                           \
    }
    PF_MBX_CASE(3, 5, 3, 1, 0)                                    // blocks 3.1 - 3.3: 80 -> 200 / 184 -> 80
    PF_MBX_CASE(3, 7, 3, 1, 1) PF_MBX_CASE(3, 7, 3, 1, 2)         // block 4.0: 80 -> 480 -> 112
    PF_MBX_CASE(4, 7, 3, 1, 1) PF_MBX_CASE(4, 7, 3, 1, 2)         // block 4.1: 112 -> 672 -> 112
    PF_MBX_CASE(4, 10, 5, 1, 1) PF_MBX_CASE(4, 10, 5, 1, 2)       // block 5.0: 112 -> 672 -> 160, 5 x 5
    PF_MBX_CASE(5, 10, 5, 2, 1) PF_MBX_CASE(5, 10, 5, 2, 2)       // blocks 5.1 / 5.2: 160 -> 960 -> 160, 5 x 5 dilated
#undef PF_MBX_CASE
    return -1;
}
