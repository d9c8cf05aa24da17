// =====================================================================================
// CPU SIMT emulator -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// A minimal stand-in for <hip/hip_runtime.h> that lets the engine's HIP sources be compiled
// by the *host* clang and executed on CPU so that kernel indexing / tiling / fragment-layout
// logic can be checked against the oracle without a GPU (this container has none).  Every
// GPU thread is a fiber; __syncthreads() and the wave-level collectives (shuffles, MFMA) are
// rendez-vous points between fibers.  MFMA fragment layouts follow the CDNA4 mapping documented
// in /opt/skills/guides/cdna_hip_programming.md section 3.
//
// The product library (libpeppa_hip.so) is always built by hipcc against the real HIP
// runtime; this header is only ever found through `-I tests/simt_emu/include`.
// =====================================================================================
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#define PF_SIMT_EMULATION 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct pf_emu_stream* hipStream_t;
typedef struct pf_emu_event { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };

namespace pf_emu {

constexpr int kWave = 64;
constexpr size_t kStackBytes = 128 * 1024;

struct Block;
struct Fiber {
    void* sp = nullptr;          // saved stack pointer
    char* stack = nullptr;
    Block* block = nullptr;
    uint3_ tid{0, 0, 0};
    int linear = 0;
    bool done = false;
};
struct WaveState {
    alignas(16) unsigned char stage[2][kWave][64];  // 64 B per lane per collective, double buffered
    int arrived = 0;
    unsigned gen = 0;
    int live = 0;
};
struct Block {
    uint3_ bid{0, 0, 0};
    dim3 bdim, gdim;
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    int live = 0;
    int arrived = 0;
    unsigned gen = 0;
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
};

extern thread_local Block* tl_block;
inline Fiber* cur() { return tl_block->cur; }

extern "C" void pf_emu_switch(void** save_sp, void* load_sp);
void yield();
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);

inline void block_barrier() {
    Block* b = tl_block;
    unsigned g = b->gen;
    b->arrived++;
    while (b->gen == g) {
        if (b->arrived >= b->live) { b->arrived = 0; b->gen++; break; }
        yield();
    }
}
// rendez-vous of the live lanes of the calling lane's wave; returns the staging buffer index to use
inline WaveState& wave_state() { return tl_block->waves[cur()->linear / kWave]; }
inline void wave_barrier() {
    WaveState& w = wave_state();
    unsigned g = w.gen;
    w.arrived++;
    while (w.gen == g) {
        if (w.arrived >= w.live) { w.arrived = 0; w.gen++; break; }
        yield();
    }
}
inline int lane_id() { return cur()->linear % kWave; }

template <typename T> inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "stage too small");
    WaveState& w = wave_state();
    int buf = w.gen & 1;
    std::memcpy(w.stage[buf][lane_id()], &v, sizeof(T));
    wave_barrier();
    T r;
    std::memcpy(&r, w.stage[buf][src_lane & (kWave - 1)], sizeof(T));
    return r;
}
}  // namespace pf_emu

#define threadIdx (pf_emu::cur()->tid)
#define blockIdx (pf_emu::tl_block->bid)
#define blockDim (pf_emu::tl_block->bdim)
#define gridDim (pf_emu::tl_block->gdim)
#define warpSize 64

using std::max;
using std::min;
// numpy-equivalent (non-contracted) float ops; the emulator is built with -ffp-contract=off
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }

inline void __syncthreads() { pf_emu::block_barrier(); }

template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = pf_emu::lane_id();
    int base = l & ~(width - 1);
    return pf_emu::wave_exchange(v, base + ((l ^ mask) & (width - 1)));
}
template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int l = pf_emu::lane_id();
    int base = l & ~(width - 1);
    int src = (l & (width - 1)) + (int)delta;
    return pf_emu::wave_exchange(v, src < width ? base + src : l);
}
template <typename T> inline T __shfl(T v, int src, int width = 64) {
    int l = pf_emu::lane_id();
    int base = l & ~(width - 1);
    return pf_emu::wave_exchange(v, base + (src & (width - 1)));
}
inline unsigned long long __ballot(int pred) {
    unsigned long long mine = pred ? (1ull << pf_emu::lane_id()) : 0ull, r = 0;
    // gather via 64 exchanges would be slow; use the staging area directly
    pf_emu::WaveState& w = pf_emu::wave_state();
    int buf = w.gen & 1;
    std::memcpy(w.stage[buf][pf_emu::lane_id()], &mine, 8);
    // lanes that already exited contribute 0
    pf_emu::wave_barrier();
    int base = (pf_emu::cur()->linear / 64) * 64;
    for (int i = 0; i < 64; ++i) {
        if (base + i < (int)pf_emu::tl_block->fibers.size() && !pf_emu::tl_block->fibers[base + i].done) {
            unsigned long long x; std::memcpy(&x, w.stage[buf][i], 8); r |= x;
        }
    }
    return r;
}

inline float atomicAdd(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f; std::memcpy(&f, &old, 4);
        const float nf = f + v;
        unsigned nu; std::memcpy(&nu, &nf, 4);
        if (__atomic_compare_exchange_n(u, &old, nu, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
struct uint4 { unsigned x, y, z, w; };
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- host runtime API subset (synchronous) -------------------------------------------
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
#define hipHostMallocPortable 0x1u
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
// hipGraph: not emulated -- capture fails loudly so tests never silently skip the replay path
typedef struct pf_emu_graph* hipGraph_t;
typedef struct pf_emu_graph_exec* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, unsigned long long) { return hipErrorInvalidValue; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
double pf_emu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new pf_emu_event{0}; return hipSuccess; }
#define hipEventDisableTiming 0x2u
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new pf_emu_event{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = pf_emu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }   // the emulator's streams are synchronous
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*smem*/, hipStream_t, Args&&... args) {
    std::tuple<std::decay_t<KArgs>...> packed(std::forward<Args>(args)...);
    std::function<void()> body = [&]() { std::apply(kernel, packed); };
    pf_emu::run_grid(grid, block, body);
}
