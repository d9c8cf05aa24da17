// Shared device-side helpers: activation-type traits, 16-byte vectors, activation functions.
//
// Activations are NHWC.  The engine is instantiated for two element types:
//   pf_half (f16 storage, f32 accumulate, v_mfma_f32_16x16x32_f16)  -- the production path
//   float   (f32 storage, v_mfma_f32_16x16x4_f32)                   -- exact verification path
// Every kernel moves activations as 16-byte vectors: 8 x f16 or 4 x f32.
#pragma once
#include <pf_intrinsics.h>  // resolved through -I (csrc/ for the product build)

// Timing ablations (wave-uniform bit masks tested inside the GEMM kernels; results are WRONG when set) exist only in the
// ablation build -- `python -m peppa_pig_face_landmark_amd.build --ablate` compiles the same sources with -DPF_ABLATE=1 into
// libpeppa_hip_ablate.so for tools/ab_env.py.  In the production library pf_dbg() is the constant 0: every ablation branch
// folds away, and no environment variable can make a kernel skip work or switch the range guard off.
#ifndef PF_ABLATE
#define PF_ABLATE 0
#endif
template <typename Args> __device__ __forceinline__ int pf_dbg(const Args& a) { return PF_ABLATE ? a.dbg : 0; }

// ---- range guard of the split-precision (f32s) kernels, always on ---------------------------------------------------------------
// Every kernel that writes f32 values as f16 hi + lo keeps a running maximum of |v| over EVERYTHING it splits, as raw bits
// (NaN > inf > every finite value in that order, so a NaN is reported as well), and commits it with one atomicMax per wave
// into its op's slot; range_verdict_kernel (k_layers.h) judges and clears the slots at the end of every forward.  Cost: an AND
// and an unsigned max per split element, next to the four conversions of the split itself.
__device__ __forceinline__ unsigned pf_amax(unsigned m, float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    return b > m ? b : m;
}
// One op owns PF_RANGE_SUBSLOTS consecutive words; a wave commits to the word its (workgroup, wave) index selects, and only if
// its maximum beats what is already there (a plain load first).  atomicMax of tens of thousands of waves on ONE address
// serialises in the L2 at ~80 ns apiece (measured: +0.5 ms on an 8192-workgroup launch, +20 us on a 960-workgroup one even
// with 16 words); spread over 256 words the first round of a launch puts ~16 on each and every later wave only loads.
#define PF_RANGE_SUBSLOTS 256
// words of one op are PF_RANGE_STRIDE words apart (32 = one word per 128-byte line: the 4096 commits of a 256-workgroup launch of
// 16-wave workgroups all arrive within a microsecond, and atomics on the same LINE queue behind each other)
#ifndef PF_RANGE_STRIDE
#define PF_RANGE_STRIDE 32
#endif
#define PF_RANGE_OP_WORDS (PF_RANGE_SUBSLOTS * PF_RANGE_STRIDE)
__device__ __forceinline__ unsigned* pf_amax_word(unsigned* slot) {
    return slot + ((((blockIdx.x + 5 * blockIdx.y) << 4) + (threadIdx.x >> 6)) & (PF_RANGE_SUBSLOTS - 1)) * PF_RANGE_STRIDE;
}
// at kernel entry: what the wave's word holds now (the load's latency hides behind the kernel body; a stale value only means an
// atomic that was not needed)
// -- but only where it pays: the words of an op sit in eight cache lines that every wave of the launch reads at the same moment,
// and vector loads return in order, so the wave's first input data waits behind this load.  A launch of fewer than 4096
// workgroups is about one generation of resident workgroups -- every wave reads "nothing committed yet" and commits anyway -- so
// there the load is skipped (ShuffleNet units at 48 x 80: 0.122 -> 0.089 ms per 32 frames; LOAD = false is for kernels of many
// SHORT workgroups that measured faster committing unconditionally: profiles/r05_run26_guard_load_variants.txt)
template <bool LOAD = true> __device__ __forceinline__ unsigned pf_amax_seen(unsigned* slot) {
    if constexpr (!LOAD) return 0u;
    return (slot && gridDim.x * gridDim.y >= 4096u) ? __atomic_load_n(pf_amax_word(slot), __ATOMIC_RELAXED) : 0u;
}
__device__ __forceinline__ void pf_amax_commit(unsigned* slot, unsigned m, unsigned seen) {
    if (!slot) return;
    // wave maximum: four DPP exchanges inside each 16-lane row, then the four row results through SGPRs
    { const unsigned o = (unsigned)pf_row_xchg_i32<0>((int)m); m = o > m ? o : m; }
    { const unsigned o = (unsigned)pf_row_xchg_i32<1>((int)m); m = o > m ? o : m; }
    { const unsigned o = (unsigned)pf_row_xchg_i32<2>((int)m); m = o > m ? o : m; }
    { const unsigned o = (unsigned)pf_row_xchg_i32<3>((int)m); m = o > m ? o : m; }
    const unsigned r0 = (unsigned)pf_readlane_i32((int)m, 0), r1 = (unsigned)pf_readlane_i32((int)m, 16);
    const unsigned r2 = (unsigned)pf_readlane_i32((int)m, 32), r3 = (unsigned)pf_readlane_i32((int)m, 48);
    const unsigned a01 = r0 > r1 ? r0 : r1, a23 = r2 > r3 ? r2 : r3;
    m = a01 > a23 ? a01 : a23;
    if ((threadIdx.x & 63) == 0 && m > seen) atomicMax(pf_amax_word(slot), m);
}

// x / d for the per-lane tile arithmetic of the stem / detector kernels (pixel -> (row, column) of a tile whose width is a launch
// argument): a 32-bit integer division is ~25 VALU instructions on this chip (no divider: reciprocal + corrections) and these kernels
// are bound by VALU issue (4 cycles per instruction).  One multiply and a shift with m = 2^20 / d + 1, computed once per thread; exact
// for 0 <= x < min(4096, 2^20 / d) (checked exhaustively for d <= 1024).  Worth 2-4 % on the stem and ShuffleNet-unit kernels
// (profiles/r05_run52: stem conv 0.097 -> 0.094 ms, units at 24 x 40 0.124 -> 0.120): the divisions were fewer than they looked.
__device__ __forceinline__ unsigned pf_div_magic(int d) { return (1u << 20) / (unsigned)d + 1u; }
__device__ __forceinline__ int pf_div_small(int x, unsigned m) { return (int)(((unsigned)x * m) >> 20); }

enum PfAct : int { PF_ACT_NONE = 0, PF_ACT_RELU = 1, PF_ACT_HSWISH = 2, PF_ACT_SILU = 3, PF_ACT_SIGMOID = 4,
                   PF_ACT_HSIGMOID = 5 };

template <typename T> struct PfVec;
template <> struct PfVec<pf_half> {
    static constexpr int N = 8;
    typedef pf_half8 type;
};
template <> struct PfVec<float> {
    static constexpr int N = 4;
    typedef pf_f32x4 type;
};

template <int ACT> __device__ __forceinline__ float pf_act_c(float v) {
    if constexpr (ACT == PF_ACT_RELU) return __builtin_fmaxf(v, 0.f);   // ONE v_max_f32 (the select form compiles to a canonicalising v_max(v, v) plus the max; NaN -> 0 either way)
    else if constexpr (ACT == PF_ACT_HSWISH || ACT == PF_ACT_HSIGMOID) {
        // x * relu6(x + 3) / 6 with the division as a multiplication (onnxruntime's HardSigmoid alpha = 1/6 form)
        float r = v + 3.f;
        r = __builtin_fminf(__builtin_fmaxf(r, 0.f), 6.f) * (1.f / 6.f);   // one v_med3 / max+min instead of two compare+select pairs
        return ACT == PF_ACT_HSWISH ? v * r : r;
    } else if constexpr (ACT == PF_ACT_SILU) return v / (1.f + expf(-v));
    else if constexpr (ACT == PF_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    else return v;
}

// one value, activation known only at run time (cold paths; folds when `act` is a literal)
__device__ __forceinline__ float pf_act(float v, int act) {
    switch (act) {
        case PF_ACT_RELU: return pf_act_c<PF_ACT_RELU>(v);
        case PF_ACT_HSWISH: return pf_act_c<PF_ACT_HSWISH>(v);
        case PF_ACT_SILU: return pf_act_c<PF_ACT_SILU>(v);
        case PF_ACT_SIGMOID: return pf_act_c<PF_ACT_SIGMOID>(v);
        case PF_ACT_HSIGMOID: return pf_act_c<PF_ACT_HSIGMOID>(v);
        default: return v;
    }
}

// N values behind ONE scalar branch on the (wave-uniform) kernel argument.  The empty asm statements keep
// the compiler from flattening the switch into "evaluate every activation, then select", which used to put
// an expf and two IEEE divisions behind every output element of every epilogue.
template <int N, typename V> __device__ __forceinline__ void pf_act_n(V& v, int act) {
    switch (act) {
        case PF_ACT_NONE: break;
#define PF_ACT_CASE(A)                                            \
    case A:                                                       \
        asm volatile("");                                         \
        _Pragma("unroll") for (int i = 0; i < N; ++i) v[i] = pf_act_c<A>(v[i]); \
        break;
        PF_ACT_CASE(PF_ACT_RELU)
        PF_ACT_CASE(PF_ACT_HSWISH)
        PF_ACT_CASE(PF_ACT_SILU)
        PF_ACT_CASE(PF_ACT_SIGMOID)
        PF_ACT_CASE(PF_ACT_HSIGMOID)
#undef PF_ACT_CASE
        default: break;
    }
}

// two-way variant for the fused encoder-block kernels: the Student's inverted-residual blocks use ReLU
// (stages 1-2) or hard-swish (stages 3-5) only (validated by the host); far lighter on registers than the
// full switch when applied to 16+ values at once
template <int N, typename V> __device__ __forceinline__ void pf_act_rh(V& v, int act) {
    if (act == PF_ACT_HSWISH) {
        asm volatile("");                      // keep the (wave-uniform) branch a branch
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = pf_act_c<PF_ACT_HSWISH>(v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = pf_act_c<PF_ACT_RELU>(v[i]);
    }
}

// none | relu | hard-swish | SiLU (everything the conv-like layers of the two networks use) behind one
// wave-uniform branch; lighter on registers than pf_act_n when applied to 16+ values at once
template <int N, typename V> __device__ __forceinline__ void mb_act(V& v, int act) {
    if (act == PF_ACT_HSWISH) {
        asm volatile("");
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = pf_act_c<PF_ACT_HSWISH>(v[i]);
    } else if (act == PF_ACT_SILU) {
        asm volatile("");
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = pf_act_c<PF_ACT_SILU>(v[i]);
    } else if (act == PF_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = pf_act_c<PF_ACT_RELU>(v[i]);
    }
}

// 16-byte global load/store of an activation vector
template <typename T> __device__ __forceinline__ typename PfVec<T>::type pf_ldv(const T* p) {
    return *reinterpret_cast<const typename PfVec<T>::type*>(p);
}
template <typename T> __device__ __forceinline__ void pf_stv(T* p, typename PfVec<T>::type v) {
    *reinterpret_cast<typename PfVec<T>::type*>(p) = v;
}
template <typename T> __device__ __forceinline__ typename PfVec<T>::type pf_zero_vec() {
    typename PfVec<T>::type z;
#pragma unroll
    for (int i = 0; i < PfVec<T>::N; ++i) z[i] = (T)0;
    return z;
}

__host__ __device__ __forceinline__ int pf_div_up(int a, int b) { return (a + b - 1) / b; }
