// CDNA4 (gfx950) device intrinsics used by the peppa-hip kernels.
//
// Thin named wrappers over the gfx950 matrix-core and cross-lane builtins so every kernel
// states which instruction it relies on:
//   pf_mfma_16x16x32_f16  -> v_mfma_f32_16x16x32_f16  (8 f16 per lane per operand, f32 accumulate)
//   pf_mfma_16x16x4_f32   -> v_mfma_f32_16x16x4_f32   (exact f32, verification mode)
// Fragment layout (cdna_hip_programming.md section 3): operand lane l supplies row/col (l & 15) and
// k-group (l >> 4); accumulator lane l, register r holds D[4*(l>>4)+r][l&15].
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 pf_half;
typedef _Float16 pf_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf_half4 __attribute__((ext_vector_type(4)));
typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pf_f32x4 pf_mfma_16x16x32_f16(pf_half8 a, pf_half8 b, pf_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ pf_f32x4 pf_mfma_16x16x4_f32(float a, float b, pf_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float pf_shfl_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int pf_shfl_xor_i32(int v, int mask) { return __shfl_xor(v, mask, 64); }

// Orders LDS traffic between the lanes of ONE wave (producer lanes write, other lanes read) without a
// workgroup barrier: the wave issues its LDS instructions in order, so only the compiler has to be fenced.
__device__ __forceinline__ void pf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// global -> LDS copy of 16 bytes per lane without a VGPR round trip (global_load_lds_dwordx4).  The LDS
// destination is wave-uniform base + lane * 16: callers pass lane-linear pointers (lane 0's pointer is the base).
// Counts on vmcnt; a following __syncthreads() drains it.
__device__ __forceinline__ void pf_glds16(const void* gsrc, void* lds_lane_ptr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_lane_ptr, 16, 0, 0);
}

#define PF_BUILD_TAG "gfx950"
#define PF_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
