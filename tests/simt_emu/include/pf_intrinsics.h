// CPU SIMT emulator flavour of csrc/pf_intrinsics.h -- TEST INFRASTRUCTURE ONLY.
// Found ahead of the product header through the include path of the emulator build; implements
// the CDNA4 matrix-core semantics the kernels rely on:
//   v_mfma_f32_16x16x32_f16 : A lane l = A[i=l&15][k=8*(l>>4)..+7], B lane l = B[k=8*(l>>4)..+7][j=l&15]
//   v_mfma_f32_16x16x4_f32  : A lane l = A[i=l&15][k=l>>4],        B lane l = B[k=l>>4][j=l&15]
//   C/D (both)              : lane l, reg r -> row 4*(l>>4)+r, col l&15
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 pf_half;
typedef _Float16 pf_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 pf_half2 __attribute__((ext_vector_type(2)));
typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));

inline pf_f32x4 pf_mfma_16x16x32_f16(pf_half8 a, pf_half8 b, pf_f32x4 c) {
    struct Pack { pf_half8 a, b; };
    static_assert(sizeof(Pack) == 32, "pack");
    pf_emu::WaveState& w = pf_emu::wave_state();
    const int buf = w.gen & 1;
    const int l = pf_emu::lane_id();
    Pack p{a, b};
    std::memcpy(w.stage[buf][l], &p, sizeof(p));
    pf_emu::wave_barrier();
    const int col = l & 15;
    pf_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            Pack pa, pb;
            std::memcpy(&pa, w.stage[buf][row + 16 * g], sizeof(Pack));
            std::memcpy(&pb, w.stage[buf][col + 16 * g], sizeof(Pack));
            for (int e = 0; e < 8; ++e) acc += (float)pa.a[e] * (float)pb.b[e];
        }
        d[r] = acc;
    }
    return d;
}

//   v_mfma_f32_16x16x16_f16 : A lane l = A[i=l&15][k=4*(l>>4)..+3], B lane l = B[k=4*(l>>4)..+3][j=l&15]
inline pf_f32x4 pf_mfma_16x16x16_f16(pf_half4 a, pf_half4 b, pf_f32x4 c) {
    struct Pack { pf_half4 a, b; };
    static_assert(sizeof(Pack) == 16, "pack");
    pf_emu::WaveState& w = pf_emu::wave_state();
    const int buf = w.gen & 1;
    const int l = pf_emu::lane_id();
    Pack p{a, b};
    std::memcpy(w.stage[buf][l], &p, sizeof(p));
    pf_emu::wave_barrier();
    const int col = l & 15;
    pf_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            Pack pa, pb;
            std::memcpy(&pa, w.stage[buf][row + 16 * g], sizeof(Pack));
            std::memcpy(&pb, w.stage[buf][col + 16 * g], sizeof(Pack));
            for (int e = 0; e < 4; ++e) acc += (float)pa.a[e] * (float)pb.b[e];
        }
        d[r] = acc;
    }
    return d;
}

inline pf_f32x4 pf_mfma_16x16x4_f32(float a, float b, pf_f32x4 c) {
    struct Pack { float a, b; };
    pf_emu::WaveState& w = pf_emu::wave_state();
    const int buf = w.gen & 1;
    const int l = pf_emu::lane_id();
    Pack p{a, b};
    std::memcpy(w.stage[buf][l], &p, sizeof(p));
    pf_emu::wave_barrier();
    const int col = l & 15;
    pf_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            Pack pa, pb;
            std::memcpy(&pa, w.stage[buf][row + 16 * g], sizeof(Pack));
            std::memcpy(&pb, w.stage[buf][col + 16 * g], sizeof(Pack));
            acc = std::fmaf(pa.a, pb.b, acc);
        }
        d[r] = acc;
    }
    return d;
}

inline float pf_shfl_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }
inline int pf_shfl_xor_i32(int v, int mask) { return __shfl_xor(v, mask, 64); }
inline int pf_shfl_i32(int v, int src_lane) { return __shfl(v, src_lane, 64); }

template <int STEP> inline int pf_row_xchg_i32(int v) {      // same lane pairing as the DPP controls of the product header
    const int l = pf_emu::lane_id();
    const int src = STEP == 0 ? (l ^ 1) : (STEP == 1 ? (l ^ 2) : (STEP == 2 ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15)))));
    return __shfl(v, src, 64);
}
template <int STEP> inline float pf_row_xchg_f32(float v) {
    const int l = pf_emu::lane_id();
    const int src = STEP == 0 ? (l ^ 1) : (STEP == 1 ? (l ^ 2) : (STEP == 2 ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15)))));
    return __shfl(v, src, 64);
}

inline int pf_readlane_i32(int v, int lane) { return __shfl(v, lane, 64); }

inline void pf_wave_sync() { pf_emu::wave_barrier(); }

inline void pf_glds16(const void* gsrc, void* lds_lane_ptr) { std::memcpy(lds_lane_ptr, gsrc, 16); }
inline void pf_glds16_raw(const void* gsrc, void* lds_lane_ptr) { std::memcpy(lds_lane_ptr, gsrc, 16); }
template <int OFF> inline void pf_glds16_raw_soff(const void* sbase, unsigned voff, void* lds_lane_ptr) { std::memcpy(lds_lane_ptr, static_cast<const unsigned char*>(sbase) + voff + OFF, 16); }
template <int OFF> inline void pf_glds16_raw_off(const void* gsrc, void* lds_lane_ptr) { std::memcpy(lds_lane_ptr, static_cast<const unsigned char*>(gsrc) + OFF, 16); }

template <int N> inline void pf_wait_vm_barrier() { __syncthreads(); }   // the emulator's copies are synchronous

inline int pf_opaque(int v) { return v; }
template <int P> inline void pf_setprio() {}
inline pf_half pf_split_lo(float v, pf_half hi) { return (pf_half)(v - (float)hi); }
inline int pf_uniform_i32(int v) { return v; }
inline void pf_sched_fence() {}
inline void pf_pin(unsigned&) {}
inline unsigned long long pf_clock() { return 0; }

#define PF_BUILD_TAG "simt-emu"
#define PF_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
