// Shared device-side helpers: activation-type traits, 16-byte vectors, activation functions.
//
// Activations are NHWC.  The engine is instantiated for two element types:
//   pf_half (f16 storage, f32 accumulate, v_mfma_f32_16x16x32_f16)  -- the production path
//   float   (f32 storage, v_mfma_f32_16x16x4_f32)                   -- exact verification path
// Every kernel moves activations as 16-byte vectors: 8 x f16 or 4 x f32.
#pragma once
#include <pf_intrinsics.h>  // resolved through -I (csrc/ for the product build)

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

__device__ __forceinline__ float pf_act(float v, int act) {
    switch (act) {
        case PF_ACT_RELU: return v > 0.f ? v : 0.f;
        case PF_ACT_HSWISH: { float r = v + 3.f; r = r < 0.f ? 0.f : (r > 6.f ? 6.f : r); return v * r / 6.f; }
        case PF_ACT_SILU: return v / (1.f + expf(-v));
        case PF_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case PF_ACT_HSIGMOID: { float r = v + 3.f; r = r < 0.f ? 0.f : (r > 6.f ? 6.f : r); return r / 6.f; }
        default: return v;
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
