// CDNA4 (gfx950) device intrinsics used by the peppa-hip kernels.
//
// Thin named wrappers over the gfx950 matrix-core and cross-lane builtins so every kernel
// states which instruction it relies on:
//   pf_mfma_16x16x32_f16  -> v_mfma_f32_16x16x32_f16  (8 f16 per lane per operand, f32 accumulate)
//   pf_mfma_16x16x16_f16  -> v_mfma_f32_16x16x16_f16  (4 f16 per lane per operand: K = 16 blocks without zero padding)
//   pf_mfma_16x16x4_f32   -> v_mfma_f32_16x16x4_f32   (exact f32, verification mode)
// Fragment layout (cdna_hip_programming.md section 3): operand lane l supplies row/col (l & 15) and
// k-group (l >> 4); accumulator lane l, register r holds D[4*(l>>4)+r][l&15].
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 pf_half;
typedef _Float16 pf_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 pf_half2 __attribute__((ext_vector_type(2)));
typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pf_f32x4 pf_mfma_16x16x32_f16(pf_half8 a, pf_half8 b, pf_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x16_f16: 4 f16 per lane per operand (k = 4 * (lane >> 4) .. + 3), same accumulator layout
__device__ __forceinline__ pf_f32x4 pf_mfma_16x16x16_f16(pf_half4 a, pf_half4 b, pf_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ pf_f32x4 pf_mfma_16x16x4_f32(float a, float b, pf_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float pf_shfl_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int pf_shfl_xor_i32(int v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int pf_shfl_i32(int v, int src_lane) { return __shfl(v, src_lane, 64); }

// Cross-lane moves inside a 16-lane row by DPP (one VALU instruction, no LDS): lane ^ 1, lane ^ 2 (quad permutes), then the
// mirror of an 8-lane half and of the whole row -- after the four steps of a reduction every lane of the row has seen all 16.
template <int STEP> __device__ __forceinline__ int pf_row_xchg_i32(int v) {
    constexpr int ctrl = STEP == 0 ? 0xB1 : (STEP == 1 ? 0x4E : (STEP == 2 ? 0x141 : 0x140));   // quad_perm[1,0,3,2] / [2,3,0,1] / row_half_mirror / row_mirror
    return __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false);
}
template <int STEP> __device__ __forceinline__ float pf_row_xchg_f32(float v) {
    return __int_as_float(pf_row_xchg_i32<STEP>(__float_as_int(v)));
}

__device__ __forceinline__ int pf_readlane_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }   // lane: compile-time constant

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

// The same copy issued from inline assembly.  The compiler orders every ds_read behind ALL pending LDS-DMA it knows about
// (s_waitcnt vmcnt(0) before the read), which serialises a ring of more than two stages; requests issued here are invisible
// to that bookkeeping, so the caller owns the ordering: pf_wait_vm_barrier<N>() before anybody reads the bytes.  m0 is saved
// and restored because the compiler may hold a value there.
__device__ __forceinline__ void pf_glds16_raw(const void* gsrc, void* lds_lane_ptr) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_lane_ptr);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(base) : "memory");
}

// ... with a compile-time byte offset in the instruction (unrolled loops: one base pointer for every step instead of one per step).
// The instruction's offset is added to BOTH addresses, the global one and the LDS one (LLVM's llvm.amdgcn.global.load.lds: "imm
// offset, applied to both global and LDS address"): the LDS base handed to M0 is moved back by it.
template <int OFF> __device__ __forceinline__ void pf_glds16_raw_off(const void* gsrc, void* lds_lane_ptr) {
    static_assert(OFF >= -4096 && OFF < 4096, "13-bit signed instruction offset");
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_lane_ptr) - (unsigned)OFF;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(base), "n"(OFF) : "memory");
}

// ... and with the address as wave-uniform base + 32-bit per-lane byte offset (one VGPR instead of two per pointer)
template <int OFF> __device__ __forceinline__ void pf_glds16_raw_soff(const void* sbase, unsigned voff, void* lds_lane_ptr) {
    static_assert(OFF >= -4096 && OFF < 4096, "13-bit signed instruction offset");
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_lane_ptr) - (unsigned)OFF;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(base), "n"(OFF) : "memory");
}

#ifndef PF_STRICT_WAITS
#define PF_STRICT_WAITS 0
#endif
// Workgroup barrier that leaves the wave's N youngest VMEM operations (LDS-DMA requests, global loads) in flight:
// "s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier".  __syncthreads() drains vmcnt to 0, which ends every software-pipeline
// stage with a full global-memory latency; LDS-DMA stays in flight across s_barrier (MI355X_MICROARCH.md).
template <int N> __device__ __forceinline__ void pf_wait_vm_barrier() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    // no fence builtins here: a workgroup-scope release fence is lowered to s_waitcnt vmcnt(0) whenever LDS-DMA is
    // pending, which is exactly the drain this barrier exists to avoid; the memory clobber keeps the compiler from moving
    // LDS / global accesses across it, lgkmcnt(0) retires this wave's own LDS writes before the rendezvous
    // PF_STRICT_WAITS (test flavour libpeppa_hip_strict.so, build.py): every partial wait drains completely, so the rings lose their
    // look-ahead but cannot read a stage early -- tests/test_gpu_race_net.py requires production == strict, bit for bit
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(PF_STRICT_WAITS ? 0 : N) : "memory");
}

// Nothing is scheduled across this point (machine scheduler only): the steps of a completely unrolled K loop are one basic block,
// and without fences the scheduler drags address arithmetic and epilogue set-up of later steps to the front until registers spill.
__device__ __forceinline__ void pf_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// the value must be in its register here (ends the compiler's freedom to postpone the arithmetic that produces it)
__device__ __forceinline__ void pf_pin(unsigned& v) { asm volatile("" : "+v"(v)); }

// v, which the CALLER knows to be the same in every lane of the wave, as a scalar (branches on it become scalar branches)
__device__ __forceinline__ int pf_uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// a copy of v the compiler must assume is a different value (loop-invariant index arithmetic derived from it stays inside the loop)
__device__ __forceinline__ int pf_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// The low half of the f32s split: lo = f16(v - f32(hi)).
// Round 5 tried to have it selected as ONE v_fma_mixlo / mixhi_f16 (f16 source, f32 addend, f16 result: 2.5 instead of 4 VALU
// instructions per element) by writing it fma(f32(hi), m, v) with m = -1 out of a scalar register the compiler cannot see through.
// The values are identical (a micro-test over 2^20 inputs and the Student / Teacher outputs bit for bit:
// profiles/r05_run39_split_mix_vs_sub_bitwise.txt, r05_run42_fma_mix_vs_sub.txt) and the pipeline gained 0.5 % -- but the same round
// found the detector at 6e-4 of its range against the oracle instead of 5e-5 whenever the compiler selects mixed-precision fma
// instructions ON ITS OWN in the detector kernels (it does once the SLP vectoriser is off: hi = mixlo(x, r, 0), lo = mixlo(x, r, -hi)
// behind the SiLU), and at 5e-5 again with the instruction family switched off (profiles/r05_run44 ... r05_run47).  The cause
// (profiles/r05_run57_fma_mix_rounding.txt, tools/fma_mix_rounding.hip): v_fma_mixlo_f16(x, r, 0) rounds the EXACT product to f16
// ONCE, (half)(float)(x * r) rounds twice, and they differ for 1 input in 15 000 -- and the compiler computes the one source-level
// value hi = (half)(x * r) BOTH ways: v_mul_f32 + v_cvt_pk_f16_f32 for the high half that is stored (the f32 product exists anyway,
// for the range guard), the mixed instruction for the one the low half is taken against.  For those inputs hi + lo is off by an
// f16 ulp (2^-11).  The library is therefore built with -fma-mix-insts (build.py), and the split is the plain subtraction
// everywhere.
__device__ __forceinline__ pf_half pf_split_lo(float v, pf_half hi) { return (pf_half)(v - (float)hi); }
// s_setprio: issue priority of this wave among the waves of its SIMD (0 = default .. 3).  A wave about to run a stretch of VALU
// work next to waves that feed the matrix pipe (whose instructions keep that pipe busy for 8-16 cycles each) gets its issue slots
// first at a higher priority.
template <int P> __device__ __forceinline__ void pf_setprio() { __builtin_amdgcn_s_setprio(P); }

// shader clock (s_memtime), for the per-wave time accounting of the ablation build
__device__ __forceinline__ unsigned long long pf_clock() { return __builtin_amdgcn_s_memtime(); }

#define PF_BUILD_TAG "gfx950"
#define PF_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
