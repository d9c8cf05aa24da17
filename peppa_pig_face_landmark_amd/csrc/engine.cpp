// peppa-hip engine: program executor + C ABI (include/peppa_hip.h).
//
// One pf_handle = one HIP device + one stream + up to PF_NET_SLOTS loaded network programs.
// A program (pf_program.h) is executed as a straight-line sequence of fused-layer kernel launches
// over a static activation arena; nothing is allocated on the hot path.
#include "../../include/peppa_hip.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "k_conv_gemm.h"
#include "k_layers.h"
#include "k_mbconv.h"
#include "k_chain.h"
#include "k_det.h"
#include "k_front.h"
#include "k_front2.h"
#include "k_hrb.h"
#include "k_hero.h"
#include "k_sepup.h"
#include "k_pwhead.h"
#include "k_mbx_args.h"
#include "k_jpeg.h"
#include "k_prepost.h"
#include "k_track.h"
#include "pf_program.h"


namespace {

// Compute units of a handle's device: 256 on MI355X (8 XCDs x 32 CUs), queried at pf_create and kept WITH THE HANDLE (pf_handle::num_cus;
// a process-wide variable would follow whichever handle was created last).  Persistent kernels size their grids with it and the tile
// pickers count rounds of workgroups over the chip with it; correctness never depends on it (tiles are strided over gridDim).  The
// blockIdx & 7 == XCD affinity of the sepup kernels is an MI355X speed assumption only.
constexpr int kDefaultCUs = 256;
// workgroups per CU x CUs of the tile-walking kernels (k_det.h det_stem_kernel, k_front.h): the CPU emulator flavour caps the grid
// at 5 workgroups so that its small test images still make every workgroup walk several tiles
#ifdef PF_SIMT_EMULATION
inline int persistent_grid_n(int tiles, int, int) { return std::min(tiles, 5); }
#else
inline int persistent_grid_n(int tiles, int per_cu, int num_cus) { return std::min(tiles, per_cu * num_cus); }
#endif
#define persistent_grid(tiles, per_cu) persistent_grid_n((tiles), (per_cu), h->num_cus)      /* `h` is in scope at every launch site */

struct Program {
    bool loaded = false;
    PfHeader hdr{};
    std::vector<PfBufRec> bufs;
    std::vector<PfTensorRec> tens;
    std::vector<PfOpRec> ops;
    char* d_const = nullptr;
    char* d_arena = nullptr;
    unsigned* d_range = nullptr;     // f32s range guard: one slot per op (k_layers.h range_verdict_kernel)
    size_t arena_bytes = 0;
    int max_batch = 0;
    int esize = 2;

    char* buf_ptr(int b) const { return d_arena + (size_t)bufs[b].offset_units * 256 * (size_t)max_batch; }
    size_t buf_item_bytes(int b) const {
        const int e = bufs[b].etype;
        const size_t es = e == PF_ELEM_ACT ? (size_t)esize : (e == PF_ELEM_U8 ? 1 : 4);
        return (size_t)bufs[b].elems_per_item * es;
    }
    char* tensor_ptr(int t) const { return buf_ptr(tens[t].buf) + (size_t)tens[t].coff * esize; }
    const void* cptr(int off) const { return off < 0 ? nullptr : (const void*)(d_const + off); }
};

struct ProfEntry { double ms = 0; int count = 0; };

struct GraphKey { const void* p[6]; int i[6]; float f[3]; unsigned long long epoch; };      // i[5]: 0 whole pf_run_frames call, 1 front half, 2 tail half (pipeline.inl)
struct GraphEntry { GraphKey key; hipGraphExec_t exec; };

}  // namespace

struct pf_handle {
    int device = 0;
    int num_cus = kDefaultCUs;      // compute units of `device`
    hipStream_t stream = nullptr;
    Program prog[PF_NET_SLOTS];
    std::string err;
    // staging for host-side inputs / outputs
    char* d_stage = nullptr;
    size_t stage_bytes = 0;
    // pipeline scratch (k_prepost)
    PipelineScratch pipe;
    // hipGraph replay of pf_run_frames* (PF_OPT_HIP_GRAPH)
    bool use_graphs = false, capturing = false;
    std::vector<GraphEntry> graphs;
    // bumped whenever a device allocation a captured graph may reference is (re)made -- scratch growth, program
    // (re)load -- so that stale graphs are destroyed instead of replayed over freed memory
    unsigned long long alloc_epoch = 0;
    // kernel-variant switches for A/B timing on the GPU (environment, read once at pf_create): PEPPA_SEPUP=patch
    // selects the previous LDS-class-filter decoder front end instead of the register-blocked one
    unsigned long long* d_dbg = nullptr;   // PEPPA_DBG & 64: cycle accounting of sepup_pipe_kernel
    int dbg = 0;             // PEPPA_DBG: timing ablations of the GEMM kernels (ConvGemmArgs::dbg), never set in production
    // tracking state of the handle's video stream (pf_track_frame, k_track.h)
    TrackState track;
    JpegState jpeg;          // pf_decode_jpeg (jpeg.inl)
    // f32s range guard (pf_common.h pf_amax, k_layers.h range_verdict_kernel): on for every forward unless switched off with
    // PF_OPT_RANGE_CHECK = 0; the slots live with each program
    int range_every = 1;
    int jpeg_entropy = 0;               // PF_OPT_JPEG_ENTROPY: 0 automatic, 1 host, 2 device (jpeg.inl)
    int jpeg_rounds = 0;                // PF_OPT_JPEG_SYNC_ROUNDS: 0 = all PF_JPEG_SYNC_ROUNDS
    unsigned long long n_calls = 0;
    int* h_status = nullptr;            // page-locked, device-visible: {code, op, value bits, program slot}
    // RCCL communicator for pf_broadcast_weights (comm.inl); created lazily, one per handle
    void* comm = nullptr;
    unsigned char comm_id[128] = {0};
    int comm_rank = -1, comm_world = 0;
    // profiling
    bool profiling = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<std::string> prof_order;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static std::string g_create_error;

static inline int host_dbg(const pf_handle* h) { return PF_ABLATE ? h->dbg : 0; }   // see pf_common.h: constant 0 in the production library

namespace { void comm_release(pf_handle* h); }   // comm.inl

#define PF_FAIL(h, ...)                                   \
    do {                                                  \
        char _b[512];                                     \
        snprintf(_b, sizeof(_b), __VA_ARGS__);            \
        (h)->err = _b;                                    \
        return 1;                                         \
    } while (0)

// every kernel launch is checked where it is made: a bad launch configuration (too much LDS, too many
// registers for the block size) must not surface one call later.  `h` is in scope at every launch site.
#undef PF_LAUNCH
#define PF_LAUNCH(kernel, grid, block, stream, ...)                                                          \
    do {                                                                                                     \
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                     \
        const hipError_t _le = hipGetLastError();                                                            \
        if (_le != hipSuccess) PF_FAIL(h, "launch of %s failed: %s (%s:%d)", #kernel, hipGetErrorString(_le), __FILE__, __LINE__); \
    } while (0)

#define PF_HIP(h, call)                                                                        \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) PF_FAIL(h, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// profiling helper: wraps one launch in an event pair when enabled
struct ProfScope {
    pf_handle* h;
    std::string tag;
    ProfScope(pf_handle* h_, const char* tag_) : h(h_) {
        if (h->profiling) { tag = tag_; (void)hipEventRecord(h->ev0, h->stream); }
    }
    ~ProfScope() {
        if (!h->profiling) return;
        (void)hipEventRecord(h->ev1, h->stream);
        (void)hipEventSynchronize(h->ev1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
        auto it = h->prof.find(tag);
        if (it == h->prof.end()) { h->prof_order.push_back(tag); it = h->prof.emplace(tag, ProfEntry()).first; }
        it->second.ms += ms;
        it->second.count += 1;
    }
};

// ---------------------------------------------------------------------------------------------
// Output tile of a workgroup-level detector kernel (k_det.h): the largest-benefit TH x TW whose input region
// ((TH-1)*S+3) x ((TW-1)*S+3) fits the kernel's MAXR LDS rows.  Cost model: launches of these kernels are chains of a few
// phases (~3 us of fixed latency = ~768 rows' worth of work), so fewer rounds of workgroups over the chip come first, then
// the smaller tile; the halo rows every extra tile re-computes count with the chip's width.
static int g_det_tile_th = 0, g_det_tile_tw = 0;     // ablation build only: PEPPA_DET_TILE=th,tw forces the tile wherever it fits (tile sweeps)
static void det_pick_tile(int num_cus, int outH, int outW, int S, int max_rows, int B, int wg_per_cu, int* TH, int* TW) {
    double best = 1e30;
    *TH = 1; *TW = 1;
    if (PF_ABLATE != 0 && g_det_tile_th > 0 && ((g_det_tile_th - 1) * S + 3) * ((g_det_tile_tw - 1) * S + 3) <= max_rows) {
        *TH = std::min(g_det_tile_th, outH); *TW = std::min(g_det_tile_tw, outW);
        return;
    }
    for (int div = 1; div <= 16; ++div) {
        const int tw = (outW + div - 1) / div;
        if (div > 1 && tw == (outW + div - 2) / (div - 1)) continue;
        const int rw = (tw - 1) * S + 3;
        for (int th = 1; th <= outH; ++th) {
            const int rows = ((th - 1) * S + 3) * rw;
            if (rows > max_rows) break;
            const long long wgs = (long long)B * ((outH + th - 1) / th) * ((outW + tw - 1) / tw);
            const long long rounds = (wgs + (long long)num_cus * wg_per_cu - 1) / ((long long)num_cus * wg_per_cu);
            const double cost = (double)rounds * (768.0 + rows) + 0.5 * (double)wgs * rows / num_cus;
            if (cost < best) { best = cost; *TH = th; *TW = tw; }
        }
    }
}

// pf_div_small (pf_common.h) is exact for 0 <= x < min(4096, 2^20 / d): launch sites whose kernels divide a staging index by a tile-derived
// row length check the largest index they will produce against that domain instead of trusting the hard-coded tile
static inline bool pf_div_small_domain_ok(int max_x_exclusive, int d) {
    return d > 0 && max_x_exclusive <= 4096 && (long long)max_x_exclusive <= (1ll << 20) / d;
}

template <typename T, bool SPLIT>
static int launch_conv(pf_handle* h, const Program& p, const PfOpRec& op, int B, unsigned* range_slot) {
    const int32_t* f = op.f;
    const PfTensorRec& ti = p.tens[f[0]];
    const PfTensorRec& to = p.tens[f[1]];
    ConvGemmArgs a{};
    a.in = p.tensor_ptr(f[0]);
    a.wt = p.cptr(f[2]);
    a.bias = (const float*)p.cptr(f[3]);
    a.out = p.tensor_ptr(f[1]);
    a.res = f[4] >= 0 ? p.tensor_ptr(f[4]) : nullptr;
    a.resLd = f[4] >= 0 ? p.tens[f[4]].ld : 0;
    a.gate = f[5] >= 0 ? (const float*)p.buf_ptr(f[5]) : nullptr;
    a.fbias = f[6] >= 0 ? (const float*)p.buf_ptr(f[6]) : nullptr;
    a.amax_val = f[17] >= 0 ? (float*)p.buf_ptr(f[17]) : nullptr;
    a.amax_idx = f[18] >= 0 ? (int*)p.buf_ptr(f[18]) : nullptr;
    a.B = B; a.inH = ti.H; a.inW = ti.W; a.inC = ti.C; a.inLd = ti.ld;
    a.outH = to.H; a.outW = to.W; a.N = f[14]; a.Npad = f[13]; a.outLd = to.ld; a.outCs = f[16];
    a.outCpad = f[16] == 1 ? to.C : f[14];
    a.KH = f[7]; a.KW = f[8]; a.stride = f[9]; a.pad = f[10]; a.dil = f[11]; a.Cpad = f[12];
    a.act = f[15]; a.amaxN = f[19]; a.store_out = f[20];
    memcpy(&a.acc_scale, &f[22], 4);
    a.dbg = h->dbg;
    a.range_slot = range_slot;
    // tile configurations: index -> (BM pixels, BN channels).  The channel tile is chosen so that
    // q tiles of NT*16 channels cover Npad with the least padding (NT <= 8), ties -> fewer tiles.
    static const int bm[PF_CONV_NCFG] = {128, 128, 256, 256, 128, 128, 128, 256, 128};
    static const int bn[PF_CONV_NCFG] = {128, 64, 32, 16, 80, 96, 112, 48, 160};
    static const int cfg_of_nt[9] = {-1, 3, 2, 7, 1, 4, 5, 6, 0};
    int cfg = f[21];
    if (cfg < 0) {
        const int t16 = a.Npad / 16;
        int best_q = 0, best_nt = 0, best_cost = 1 << 30;
        for (int q = (t16 + 7) / 8; q <= (t16 + 7) / 8 + 3; ++q) {
            const int nt = (t16 + q - 1) / q;
            if (nt < 1 || nt > 8) continue;
            if (q * nt < best_cost) { best_cost = q * nt; best_q = q; best_nt = nt; }
        }
        (void)best_q;
        cfg = cfg_of_nt[best_nt];
        // 160 output channels (stage-5 projections, K = 672 / 960): one 128 x 160 tile reads the wide input once
        // instead of twice (two 80-channel tiles); split-precision pointwise only
        if (SPLIT && f[23] != 0 && a.Npad == 160 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.amax_val) cfg = 8;
    }
    const int M = B * a.outH * a.outW;
    if (a.amax_val && ((a.outH * a.outW) % bm[cfg]) != 0) PF_FAIL(h, "argmax conv: H*W=%d not a multiple of BM=%d", a.outH * a.outW, bm[cfg]);
    dim3 grid(pf_div_up(M, bm[cfg]), pf_div_up(a.Npad, bn[cfg]));
    char tagbuf[96];
    tagbuf[0] = 0;
    if (h->profiling)
        snprintf(tagbuf, sizeof(tagbuf), "conv%dx%d%s_c%d_n%d_%dx%d", a.KH, a.KW, a.amax_val ? "_argmax" : "", a.inC, a.N,
                 a.outH, a.outW);
    ProfScope ps(h, tagbuf);
    const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;
    const bool use_split = f[23] != 0;   // per-conv choice made by the packer (weights are laid out accordingly)
#define PF_CONV_CASE(idx, BM_, BN_, WM_, WN_)                                                              \
    case idx:                                                                                           \
        if (SPLIT && use_split) {   /* 8 waves per workgroup: twice the M-waves of the direct kernel */ \
            if (pointwise) PF_LAUNCH((conv_gemm_split_kernel<BM_, BN_, 2 * WM_, WN_, 1>), grid, dim3(512), h->stream, a); \
            else PF_LAUNCH((conv_gemm_split_kernel<BM_, BN_, 2 * WM_, WN_, 3>), grid, dim3(512), h->stream, a);          \
        } else {                                                                                        \
            if (pointwise) PF_LAUNCH((conv_gemm_kernel<T, BM_, BN_, WM_, WN_, 1>), grid, dim3(256), h->stream, a); \
            else PF_LAUNCH((conv_gemm_kernel<T, BM_, BN_, WM_, WN_, 3>), grid, dim3(256), h->stream, a);          \
        }                                                                                               \
        break;
    // 3x3 / stride 1 / pad 1 with 128 outputs on 16-, 32- or 64-pixel-wide maps: input patch resident in LDS
    // (the Student's hero conv; HRNet's 18 / 36 / 72-channel 3x3 stacks of the Teacher take the narrow variants)
    if (SPLIT && use_split && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && !a.gate && !a.amax_val &&
        (a.outW == 16 || a.outW == 32 || a.outW == 64) && ((a.outH * a.outW) % 128) == 0 && a.inH == a.outH && a.inW == a.outW &&
        (a.Npad == 128 || (a.Npad == 64 && a.outW == 64) || a.Npad == 32 || a.Npad == 48 || a.Npad == 80)) {
        if constexpr (SPLIT) {
            grid = dim3(pf_div_up(M, 128), 1);
            const bool big = ((a.outH * a.outW) % 256) == 0 && !(host_dbg(h) & 1024);     // narrow variants: 256-pixel tiles
            if (big && a.Npad <= 64) grid = dim3(pf_div_up(M, 256), 1);
            // k_hero.h loads all 128 channels of every pixel as 16-byte vectors, unmasked: a 3x3 conv whose inC < Cpad == 128 would feed
            // neighbouring bytes to the MFMAs and the range guard -- such a layer takes the masked halo kernel below
            const bool hero = a.Npad == 128 && a.Cpad == 128 && a.inC == 128 && (a.inLd & 3) == 0 && a.outW == 64;
            if (hero && (host_dbg(h) & 16384)) PF_LAUNCH((conv3x3_hero_kernel<4, false>), grid, dim3(512), h->stream, a);   // A/B aid (ablation build)
            else if (hero && f[23] == 2) PF_LAUNCH((conv3x3_hero_kernel<4, true, true>), grid, dim3(512), h->stream, a);   // ONE f16 product (opt-in per conv)
            else if (hero && !(host_dbg(h) & 2048)) PF_LAUNCH((conv3x3_hero_kernel<4>), grid, dim3(512), h->stream, a);   // k_hero.h
            else if (a.Npad == 128) PF_LAUNCH((conv3x3_halo_split_kernel<128, 4, 2>), grid, dim3(512), h->stream, a);
            else if (a.Npad == 64 && big) PF_LAUNCH((conv3x3_halo_split_kernel<64, 4, 2, 256>), grid, dim3(512), h->stream, a);   // HRNet layer1's 64 -> 64
            else if (a.Npad == 64) PF_LAUNCH((conv3x3_halo_split_kernel<64, 4, 2>), grid, dim3(512), h->stream, a);
            else if (a.Npad == 80) PF_LAUNCH((conv3x3_halo_split_kernel<80, 8, 1>), grid, dim3(512), h->stream, a);
            else if (a.Npad == 48 && big) PF_LAUNCH((conv3x3_halo_split_kernel<48, 8, 1, 256>), grid, dim3(512), h->stream, a);
            else if (a.Npad == 48) PF_LAUNCH((conv3x3_halo_split_kernel<48, 8, 1>), grid, dim3(512), h->stream, a);
            else if (big) PF_LAUNCH((conv3x3_halo_split_kernel<32, 8, 1, 256>), grid, dim3(512), h->stream, a);
            else PF_LAUNCH((conv3x3_halo_split_kernel<32, 8, 1>), grid, dim3(512), h->stream, a);
            return 0;
        }
    }
    // heat-map score head: bias-only arg-max epilogue (nothing stored, tiles never straddle a face)
    if (SPLIT && use_split && pointwise && cfg == 0 && a.amax_val && !a.store_out && !a.res && !a.fbias && !a.gate && a.act == PF_ACT_NONE &&
        (M % 128) == 0) {
        if constexpr (SPLIT) {
            // 128 input channels (the Student's and the Teacher's head): the weight-stationary stream of k_pwhead.h
            if (a.Cpad == 128 && a.inC == 128 && a.Npad <= 112 && ((a.outH * a.outW) % 128) == 0 && !(host_dbg(h) & 524288))
            {
                // work item = a run of tiles of one face; a face is split only while there are fewer faces than CUs
                const int tpf = (a.outH * a.outW) / 128;
                int segs = 1;
#ifdef PF_SIMT_EMULATION
                const int fill = 8;                 // CPU test build: a few faces must still exercise items of SEVERAL tiles and several items per face
#else
                const int fill = h->num_cus;
#endif
                while (segs < tpf && (tpf % (2 * segs)) == 0 && B * segs < fill) segs *= 2;
                if (B * segs > fill) {              // more items than CUs: pick the split whose LAST round of the persistent grid is full (384 faces on
                    double best = 1e30;             // 256 CUs: two rounds of faces, the second half empty, or three rounds of half faces)
                    for (int sg = 1; sg <= 8 && sg <= tpf && (tpf % sg) == 0; sg *= 2) {
                        const double face_times = (double)pf_div_up(B * sg, fill) / sg + 0.02 * (sg - 1);
                        if (face_times < best - 1e-9) { best = face_times; segs = sg; }
                    }
                }
                a.head_segs = segs;
                if (f[23] == 2) PF_LAUNCH((pw_head_kernel<4, true>), dim3(persistent_grid(B * segs, 1)), dim3(512), h->stream, a);   // ONE f16 product (opt-in per conv)
                else PF_LAUNCH((pw_head_kernel<4>), dim3(persistent_grid(B * segs, 1)), dim3(512), h->stream, a);
            }
            else if (a.Cpad == 128) PF_LAUNCH((conv_gemm_split_kernel<128, 128, 4, 2, 1, 0, -1, 1, 0, 4>), grid, dim3(512), h->stream, a);   // K loop unrolled, two steps ahead
            else PF_LAUNCH((conv_gemm_split_kernel<128, 128, 4, 2, 1, 0, -1>), grid, dim3(512), h->stream, a);
            return 0;
        }
    }
    // plain pointwise convs whose K depth has an unrolled instance (two K steps of look-ahead, k_conv_gemm.h): the Student's
    // stage-3 to stage-5 projections at 256 x 256 and a few neighbours; every other depth takes the rolled loop of the same kernel
    if constexpr (SPLIT) {
        if (use_split && pointwise && !a.amax_val) {
            const int nk = a.Cpad / 32;
#define PF_PW_NK(CFG, BM_, BN_, WM_, WN_, NK_)                                                                        \
            if (cfg == CFG && nk == NK_) {                                                                            \
                PF_LAUNCH((conv_gemm_split_kernel<BM_, BN_, WM_, WN_, 1, 0, 0, 1, 0, NK_>), grid, dim3(512), h->stream, a); \
                return 0;                                                                                             \
            }
            // measured (profiles/r04_run25 vs run23): 960 -> 160 0.204 -> 0.175 ms per 256 faces, 480 -> 112 -10 %; the shallow ones
            // (K <= 224, and the expand + depthwise launches) did not move and keep the rolled loop
            PF_PW_NK(8, 128, 160, 4, 2, 30) PF_PW_NK(8, 128, 160, 4, 2, 21)
            PF_PW_NK(6, 128, 112, 8, 1, 21) PF_PW_NK(6, 128, 112, 8, 1, 15)
#undef PF_PW_NK
        }
    }
    if (cfg == 8) {
        if constexpr (SPLIT) {
            PF_LAUNCH((conv_gemm_split_kernel<128, 160, 4, 2, 1>), grid, dim3(512), h->stream, a);
            return 0;
        } else {
            PF_FAIL(h, "conv tile configuration 8 is split-precision only");
        }
    }
    switch (cfg) {
        PF_CONV_CASE(0, 128, 128, 2, 2)
        PF_CONV_CASE(1, 128, 64, 2, 2)
        PF_CONV_CASE(2, 256, 32, 4, 1)
        PF_CONV_CASE(3, 256, 16, 4, 1)
        PF_CONV_CASE(4, 128, 80, 4, 1)
        PF_CONV_CASE(5, 128, 96, 4, 1)
        PF_CONV_CASE(6, 128, 112, 4, 1)
        default:
            PF_CONV_CASE(7, 256, 48, 4, 1)
    }
#undef PF_CONV_CASE
    return 0;
}

template <typename T, bool SPLIT>
static int run_program_t(pf_handle* h, int slot, const void* d_input, int input_kind, int B) {
    Program& p = h->prog[slot];
    constexpr int VE = PfVec<T>::N;
    const bool guard = SPLIT && h->range_every > 0 && p.d_range != nullptr;     // every call, graph-captured ones included
    auto slot_of = [&](size_t oi) -> unsigned* { return guard ? p.d_range + oi * PF_RANGE_OP_WORDS : nullptr; };
    for (size_t oi = 0; oi < p.ops.size(); ++oi) {
        const PfOpRec& op = p.ops[oi];
        const int32_t* f = op.f;
        switch (op.code) {
            case PF_OP_STEM: {
                const PfTensorRec& to = p.tens[f[1]];
                StemArgs a{};
                a.in = f[0] < 0 ? d_input : (const void*)p.tensor_ptr(f[0]);
                a.in_f32_nchw = (f[0] < 0 && input_kind == PF_INPUT_F32_NCHW) ? 1 : 0;
                a.wt = (const float*)p.cptr(a.in_f32_nchw ? f[5] : f[2]);
                a.bias = (const float*)p.cptr(f[3]);
                a.out = p.tensor_ptr(f[1]);
                a.B = B; a.inH = p.hdr.in_h; a.inW = p.hdr.in_w;
                a.outH = to.H; a.outW = to.W; a.outLd = to.ld; a.act = f[4]; a.CO = to.C;
                if (to.C % 16) PF_FAIL(h, "stem conv needs a multiple of 16 output channels, got %d", to.C);
                if (a.act != PF_ACT_NONE && a.act != PF_ACT_RELU && a.act != PF_ACT_HSWISH && a.act != PF_ACT_SILU) PF_FAIL(h, "stem conv: unsupported activation %d", a.act);
                ProfScope ps(h, "stem_conv");
                if constexpr (SPLIT) {
                    // f32s programs whose packer provided MFMA weights: the staged-image matrix-core kernel (k_front.h)
                    if (f[0] < 0 && f[6] >= 0 && (to.C == 16 || to.C == 64) && (p.hdr.in_w & 3) == 0 && ((size_t)d_input & 3) == 0 && to.H == p.hdr.in_h / 2) {
                        StemMfmaArgs s{};
                        s.in = d_input; s.out = (float*)p.tensor_ptr(f[1]); s.outLd = to.ld;
                        s.w_u8 = (const pf_half*)p.cptr(f[6]); s.w_f32 = (const pf_half*)p.cptr(f[7]); s.bias = (const float*)p.cptr(f[3]);
                        memcpy(&s.s_u8, &f[8], 4); memcpy(&s.s_f32, &f[9], 4);
                        s.B = B; s.H = p.hdr.in_h; s.W = p.hdr.in_w; s.OH = to.H; s.OW = to.W; s.act = a.act;
                        s.TH = 8; s.TW = 32; s.tilesX = pf_div_up(to.W, s.TW);
                        s.range_slot = slot_of(oi);
                        // the float-input staging loop divides i < IRH * IRW * 3 by IRW * 3 with pf_div_small (IRH = 2 TH + 1, IRW = 2 TW + 1)
                        if (!pf_div_small_domain_ok((2 * s.TH + 1) * (2 * s.TW + 1) * 3, (2 * s.TW + 1) * 3)) PF_FAIL(h, "stem: tile %dx%d outside pf_div_small's exact range", s.TH, s.TW);
                        const dim3 sg(s.tilesX * pf_div_up(to.H, s.TH), B);
                        // tile 8 x 32 output pixels: image region 17 rows x 65 pixels (200 halves per LDS row)
                        if (to.C == 16) {
                            if (a.in_f32_nchw) PF_LAUNCH((stem_mfma_kernel<1, 256, 17, 200, true>), sg, dim3(256), h->stream, s);
                            else PF_LAUNCH((stem_mfma_kernel<1, 256, 17, 200, false>), sg, dim3(256), h->stream, s);
                        } else {
                            if (a.in_f32_nchw) PF_LAUNCH((stem_mfma_kernel<4, 256, 17, 200, true>), sg, dim3(256), h->stream, s);
                            else PF_LAUNCH((stem_mfma_kernel<4, 256, 17, 200, false>), sg, dim3(256), h->stream, s);
                        }
                        break;
                    }
                }
                PF_LAUNCH((stem_conv_kernel<T>), dim3(pf_div_up(B * to.H * to.W, 256), to.C / 16), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_CONV:
                if (launch_conv<T, SPLIT>(h, p, op, B, slot_of(oi))) return 1;
                break;
            case PF_OP_SEPUP: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused upsample+depthwise+pointwise op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& tl = p.tens[f[0]];
                    const PfTensorRec& tk = p.tens[f[1]];
                    const PfTensorRec& to = p.tens[f[2]];
                    ConvGemmArgs a{};
                    a.up_lo = (const float*)p.tensor_ptr(f[0]); a.up_skip = (const float*)p.tensor_ptr(f[1]);
                    a.out = p.tensor_ptr(f[2]);
                    a.dw_w = (const float*)p.cptr(f[3]); a.dw_b = (const float*)p.cptr(f[4]); a.dw_w2 = (const float*)p.cptr(f[12]);
                    a.wt = p.cptr(f[5]); a.bias = (const float*)p.cptr(f[6]);
                    a.Cpad = f[7]; a.Npad = f[8]; a.N = f[9]; a.act = f[10]; memcpy(&a.acc_scale, &f[11], 4);
                    a.loH = tl.H; a.loW = tl.W; a.C1 = tl.C; a.loLd = tl.ld; a.skipLd = tk.ld;
                    a.B = B; a.inH = to.H; a.inW = to.W; a.inC = tl.C + tk.C; a.inLd = 0;
                    a.outH = to.H; a.outW = to.W; a.outLd = to.ld; a.outCs = 1; a.outCpad = to.C;
                    a.KH = a.KW = 1; a.stride = 1; a.pad = 0; a.dil = 1; a.store_out = 1;
                    a.dbg = h->dbg;
                    a.range_slot = slot_of(oi);
                    if (to.H != 2 * tl.H || to.W != 2 * tl.W || tk.H != to.H || tk.W != to.W || (tl.C % 32) != 0 || to.H < 6 || to.W < 6)
                        PF_FAIL(h, "sepup: inconsistent tensor shapes");
                    dim3 grid(pf_div_up(B * to.H * to.W, 128), pf_div_up(a.Npad, 128));
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "sepup_c%d_n%d_%dx%d", a.inC, a.N, to.H, to.W);
                    ProfScope ps(h, tagbuf);
                    const bool patch_ok = (to.W == 16 || to.W == 32 || to.W == 64) && ((to.H * to.W) % 128) == 0 && (tl.C % 32) == 0 && a.Cpad <= 640;
                    // producer / consumer pipelined kernel (k_sepup.h): persistent workgroups, one per CU
                    const bool pipe_ok = patch_ok && (to.H * to.W) / 128 >= 2 && (tk.C % 8) == 0 && tk.C <= 64 && a.N == a.Npad && (a.Npad == 128 || a.Npad == 256) &&
                                         f[13] > 0 && f[14] > 0 && f[15] > 0 && !(host_dbg(h) & 2048);
                    if (pipe_ok) {
                        SepupArgs s{};
                        s.lo = a.up_lo; s.skip = a.up_skip; s.out = (float*)a.out; s.dw_lo = (const float*)p.cptr(f[14]); s.dw_w2 = a.dw_w2;
                        s.dw_v = (const float*)p.cptr(f[15]);
                        s.gap_part = f[16] > 0 ? (float*)p.buf_ptr(f[16] - 1) : nullptr;
                        if (s.gap_part && !(a.Npad == 256)) PF_FAIL(h, "sepup: per-tile channel sums need the 256-output instance");
                        s.wt = (const unsigned char*)a.wt; s.bias = a.bias; s.skipx = (unsigned char*)p.buf_ptr(f[13]);
                        s.B = B; s.H = to.H; s.C1 = tl.C; s.C2 = tk.C; s.loLd = tl.ld; s.skipLd = tk.ld; s.outLd = to.ld;
                        s.N = a.N; s.Cpad = a.Cpad; s.act = a.act; s.acc_scale = a.acc_scale; s.dbg = h->dbg; s.range_slot = a.range_slot;
                        if (host_dbg(h) & 64) {      // per-role cycle accounting of the pipelined kernel (printed at pf_destroy)
                            if (!h->d_dbg) { PF_HIP(h, hipMalloc((void**)&h->d_dbg, 64 * 16 * sizeof(unsigned long long))); PF_HIP(h, hipMemset(h->d_dbg, 0, 64 * 16 * sizeof(unsigned long long))); }
                            s.prof = h->d_dbg + (a.Npad == 128 ? 0 : 16);
                        }
                        const int tpf = to.H * to.W / 128, nskip = a.Cpad / 32 - tl.C / 32;
                        const int per_xcd = ((B + 7) / 8) * tpf;                  // tiles of the busiest XCD
                        const int wgs = 8 * std::min(h->num_cus / 8, per_xcd);
                        const dim3 sg(B * tpf);
// (VCOL: at W = 16 a tile is eight image rows and their eight filter sets do not fit the request stage)
#define PF_SEPUP_CASE(WW)                                                                                          \
    if (to.W == WW) {                                                                                              \
        if (nskip > 0 && tk.C <= 32) PF_LAUNCH((sepup_skip_kernel<WW, 32>), sg, dim3(512), h->stream, s);          \
        else if (nskip > 0) PF_LAUNCH((sepup_skip_kernel<WW, 64>), sg, dim3(512), h->stream, s);                   \
        if (a.Npad == 128) PF_LAUNCH((sepup_pipe_kernel<128, WW, 3, false, true, false, false, (WW >= 32)>), dim3(wgs), dim3(1024), h->stream, s);     \
        else PF_LAUNCH((sepup_pipe_kernel<256, WW, 2, true, false, false, false, (WW >= 32)>), dim3(wgs), dim3(1024), h->stream, s);                   \
    }
                        PF_SEPUP_CASE(64) PF_SEPUP_CASE(32) PF_SEPUP_CASE(16)
#undef PF_SEPUP_CASE
                    } else if (f[16] > 0) {
                        PF_FAIL(h, "sepup: the program asks for per-tile channel sums, which only the pipelined kernel produces");
                    } else
                    if (patch_ok && a.Npad == 256) {
                        grid.y = 1;
                        PF_LAUNCH((sepup_patch_kernel<256, 4, 2>), grid, dim3(512), h->stream, a);
                    } else if (patch_ok && a.Npad <= 128) {
                        PF_LAUNCH((sepup_patch_kernel<128, 4, 2>), grid, dim3(512), h->stream, a);
                    } else if (a.Npad == 256) {   // both 128-channel halves from one pass of the (expensive) fused producer
                        grid.y = 1;
                        PF_LAUNCH((conv_gemm_split_kernel<128, 256, 4, 2, 1, 1>), grid, dim3(512), h->stream, a);
                    } else {
                        PF_LAUNCH((conv_gemm_split_kernel<128, 128, 4, 2, 1, 1>), grid, dim3(512), h->stream, a);
                    }
                }
                break;
            }
            case PF_OP_MBCONV: {
                if constexpr (!std::is_same<T, float>::value) {
                    PF_FAIL(h, "fused inverted-residual op needs f32 tensors (f32 / f32s program)");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    MbconvArgs a{};
                    a.in = (const float*)p.tensor_ptr(f[0]); a.out = (float*)p.tensor_ptr(f[1]);
                    a.res = f[2] >= 0 ? (const float*)p.tensor_ptr(f[2]) : nullptr;
                    a.resLd = f[2] >= 0 ? p.tens[f[2]].ld : 0;
                    a.w_exp = (const pf_half*)p.cptr(f[3]); a.b_exp = (const float*)p.cptr(f[4]);
                    a.w_dw = (const float*)p.cptr(f[5]); a.b_dw = (const float*)p.cptr(f[6]);
                    a.w_pwl = (const pf_half*)p.cptr(f[7]); a.b_pwl = (const float*)p.cptr(f[8]);
                    const int K = f[9], S = f[10], dil = f[12], KS = f[15];
                    a.pad = f[11]; a.act = f[13]; a.MidPad = f[14]; a.CoutPad = f[16]; a.Cout = f[17]; a.Mid16 = f[18];
                    memcpy(&a.scale_exp, &f[19], 4); memcpy(&a.scale_pwl, &f[20], 4);
                    a.B = B; a.inH = ti.H; a.inW = ti.W; a.Cin = ti.C; a.inLd = ti.ld;
                    a.outH = to.H; a.outW = to.W; a.outLd = to.ld;
                    a.act_dw = a.act; a.act_out = PF_ACT_NONE; a.outCs = 1;
                    a.range_slot = slot_of(oi);
                    if (f[21] == 3) {   // ShuffleNetV2 unit: separate activations, channel-strided store, pass-through copy
                        a.act_dw = f[22]; a.act_out = f[23]; a.outCs = f[24];
                        if (f[25] >= 0) {
                            a.pass_src = (const float*)p.tensor_ptr(f[25]); a.pass_dst = (float*)p.tensor_ptr(f[26]);
                            a.passLd = p.tens[f[25]].ld; a.passC = p.tens[f[25]].C;
                            if (a.passC % 16) PF_FAIL(h, "shuffle unit: pass-through channels must be a multiple of 16");
                        }
                    }
                    if ((f[21] == 0 && ((a.MidPad % 32) || a.Cin > 32 * KS)) || (a.Cin % 8) || K != 3 || dil != 1) PF_FAIL(h, "mbconv: unsupported block shape");
                    if (f[21] != 3 && a.act != PF_ACT_RELU && a.act != PF_ACT_HSWISH) PF_FAIL(h, "mbconv: activation must be relu or hard-swish");
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "%s_k%ds%d_c%d_m%d_n%d_%dx%d", f[21] == 3 ? "shuffle" : "mbconv", K, S, a.Cin, a.Mid16, a.Cout, to.H, to.W);
                    ProfScope ps(h, tagbuf);
                    // (stride, Cin/32, Cout/16) -> patch shape and mid-channel split; low-resolution blocks use MSPLIT = 4
#define PF_MBCONV_CASE(SS, KSS, PHH, PWW, NTT, MS)                                                                \
    if (S == SS && KS == KSS && a.CoutPad <= 16 * NTT) {                                                           \
        const int patches = pf_div_up(to.H, PHH) * pf_div_up(to.W, PWW);                                           \
        PF_LAUNCH((mbconv_wave_kernel<SS, KSS, PHH, PWW, NTT, MS>),                                                \
                  dim3(MS > 1 ? patches : pf_div_up(patches, 4), B), dim3(256), h->stream, a);                     \
    } else
                    if (f[21] == 2) {   // depthwise-separable block (no expand conv): dw 3x3 + act -> pointwise [+ x]
                        a.w_pwl32 = (const float*)p.cptr(f[7]);
                        if (S != 1 || a.Cin != 16 || a.Mid16 != 16 || a.MidPad != 16 || a.CoutPad > 32) PF_FAIL(h, "dsconv: unsupported block shape");
                        // (exact f32 in f32s programs too: with no expand conv the block is bandwidth-bound, the split flavour measured 0.17 vs 0.16 ms)
                        if (a.act == PF_ACT_RELU) PF_LAUNCH((mbconv_wave_f32_kernel<1, 16, 4, 8, true, false, PF_ACT_RELU>), dim3(pf_div_up(pf_div_up(to.H, 4) * pf_div_up(to.W, 8), 4), B), dim3(256), h->stream, a);
                        else PF_LAUNCH((mbconv_wave_f32_kernel<1, 16, 4, 8, true>), dim3(pf_div_up(pf_div_up(to.H, 4) * pf_div_up(to.W, 8), 4), B), dim3(256), h->stream, a);
                    } else if (f[21] == 1) {   // exact-f32 variant (high-resolution blocks), weights packed as f32
                        a.w_exp32 = (const float*)p.cptr(f[3]); a.w_pwl32 = (const float*)p.cptr(f[7]);
                        const int CP = f[15];
                        if (a.MidPad != a.Mid16 || a.CoutPad > 32) PF_FAIL(h, "mbconv(f32): unsupported block shape");
                        if (S == 2 && CP == 16) {
                            const dim3 g(pf_div_up(pf_div_up(to.H, 4) * pf_div_up(to.W, 4), 4), B);
                            if (SPLIT && a.act == PF_ACT_RELU) PF_LAUNCH((mbconv_wave_f32_kernel<2, 16, 4, 4, false, true, PF_ACT_RELU>), g, dim3(256), h->stream, a);
                            else if (SPLIT) PF_LAUNCH((mbconv_wave_f32_kernel<2, 16, 4, 4, false, true>), g, dim3(256), h->stream, a);
                            else PF_LAUNCH((mbconv_wave_f32_kernel<2, 16, 4, 4>), g, dim3(256), h->stream, a);
                        } else if (S == 1 && CP == 32) {
                            const dim3 g(pf_div_up(pf_div_up(to.H, 4) * pf_div_up(to.W, 8), 4), B);
                            if (SPLIT && a.act == PF_ACT_RELU) PF_LAUNCH((mbconv_wave_f32_kernel<1, 32, 4, 8, false, true, PF_ACT_RELU>), g, dim3(256), h->stream, a);
                            else if (SPLIT) PF_LAUNCH((mbconv_wave_f32_kernel<1, 32, 4, 8, false, true>), g, dim3(256), h->stream, a);
                            else PF_LAUNCH((mbconv_wave_f32_kernel<1, 32, 4, 8>), g, dim3(256), h->stream, a);
                        }
                        else PF_FAIL(h, "mbconv(f32): no kernel for stride %d, %d input channels", S, a.Cin);
                    } else
                    if (f[21] == 3) {
                        PF_MBCONV_CASE(1, 1, 4, 8, 2, 1)
                        PF_MBCONV_CASE(1, 2, 4, 8, 4, 4)
                        PF_MBCONV_CASE(1, 4, 4, 4, 8, 4)
                        PF_FAIL(h, "shuffle unit: no kernel for %d channels", a.Cin);
                    } else
                    PF_MBCONV_CASE(2, 2, 4, 4, 5, 4)
                    PF_MBCONV_CASE(1, 3, 4, 8, 5, 4)
                    PF_FAIL(h, "mbconv: no kernel for stride %d, %d input channels, %d output channels", S, a.Cin, a.Cout);
#undef PF_MBCONV_CASE
                }
                break;
            }
            case PF_OP_DETUNIT: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused ShuffleV2Block op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    DetUnitArgs a{};
                    a.in = (const float*)p.tensor_ptr(f[0]); a.out = (float*)p.tensor_ptr(f[1]);
                    a.w1 = (const pf_half*)p.cptr(f[2]); a.b1 = (const float*)p.cptr(f[3]);
                    a.wd = (const float*)p.cptr(f[4]); a.bd = (const float*)p.cptr(f[5]);
                    a.w2 = (const pf_half*)p.cptr(f[6]); a.b2 = (const float*)p.cptr(f[7]);
                    a.wd1 = (const float*)p.cptr(f[8]); a.bd1 = (const float*)p.cptr(f[9]);
                    a.w3 = (const pf_half*)p.cptr(f[10]); a.b3 = (const float*)p.cptr(f[11]);
                    memcpy(&a.s1, &f[12], 4); memcpy(&a.s2, &f[13], 4); memcpy(&a.s3, &f[14], 4);
                    const int C = f[15], K1 = f[16], S = f[17];
                    a.Cin = f[18];
                    a.B = B; a.inH = ti.H; a.inW = ti.W; a.inLd = ti.ld; a.outH = to.H; a.outW = to.W; a.outLd = to.ld;
                    a.range_slot = slot_of(oi);
                    if (host_dbg(h) & 4096) {      // per-phase cycle accounting of det_unit_kernel (ablation build; printed at pf_destroy)
                        if (!h->d_dbg) { PF_HIP(h, hipMalloc((void**)&h->d_dbg, 64 * 16 * sizeof(unsigned long long))); PF_HIP(h, hipMemset(h->d_dbg, 0, 64 * 16 * sizeof(unsigned long long))); }
                        a.prof = h->d_dbg + 64 + 8 * ((C == 32 ? 0 : (C == 64 ? 1 : 2)) + 3 * (S - 1));
                    }
                    if (to.C != 2 * C || ti.C != a.Cin || (S != 1 && S != 2) || to.H != (ti.H - 1) / S + 1 || to.W != (ti.W - 1) / S + 1 ||
                        (S == 1 && a.Cin != 2 * C) || (S == 2 && !a.w3))
                        PF_FAIL(h, "detunit: inconsistent shapes");
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "unit_s%d_c%d_%dx%d", S, C, to.H, to.W);
                    ProfScope ps(h, tagbuf);
#define PF_DETUNIT_CASE(CC, KK, SS, MAXR, NTHR, PERCU)                                                             \
    if (C == CC && K1 == KK && S == SS) {                                                                          \
        det_pick_tile(h->num_cus, to.H, to.W, SS, MAXR, B, PERCU, &a.TH, &a.TW);                                               \
        a.tilesX = pf_div_up(to.W, a.TW); a.tpf = a.tilesX * pf_div_up(to.H, a.TH);                                 \
        PF_LAUNCH((det_unit_kernel<CC, KK, SS, MAXR, NTHR, PERCU * NTHR / 256>), dim3(a.tpf * B), dim3(NTHR), h->stream, a); \
    } else
                    PF_DETUNIT_CASE(32, 32, 1, 256, 512, 2)
                    PF_DETUNIT_CASE(64, 64, 1, 128, 512, 2)
                    PF_DETUNIT_CASE(128, 128, 1, 144, 512, 1)
                    PF_DETUNIT_CASE(32, 32, 2, 480, 512, 1)
                    PF_DETUNIT_CASE(64, 64, 2, 256, 512, 1)
                    PF_DETUNIT_CASE(128, 128, 2, 128, 512, 1)
                    PF_FAIL(h, "detunit: no kernel for %d branch channels, K %d, stride %d", C, K1, S);
#undef PF_DETUNIT_CASE
                }
                break;
            }
            case PF_OP_DETSTEM: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused StemBlock op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& to = p.tens[f[0]];
                    DetStemArgs a{};
                    a.in = d_input; a.in_f32_nchw = input_kind == PF_INPUT_F32_NCHW ? 1 : 0;
                    a.out = (float*)p.tensor_ptr(f[0]); a.outLd = to.ld;
                    a.w1_u8 = (const pf_half*)p.cptr(f[1]); a.w1_f32 = (const pf_half*)p.cptr(f[2]); a.b1 = (const float*)p.cptr(f[3]);
                    a.w2a = (const pf_half*)p.cptr(f[4]); a.b2a = (const float*)p.cptr(f[5]);
                    a.w2b = (const pf_half*)p.cptr(f[6]); a.b2b = (const float*)p.cptr(f[7]);
                    a.w3 = (const pf_half*)p.cptr(f[8]); a.b3 = (const float*)p.cptr(f[9]);
                    memcpy(&a.s1_u8, &f[10], 4); memcpy(&a.s1_f32, &f[11], 4); memcpy(&a.s2a, &f[12], 4); memcpy(&a.s2b, &f[13], 4); memcpy(&a.s3, &f[14], 4);
                    a.B = B; a.H = p.hdr.in_h; a.W = p.hdr.in_w; a.SH = (a.H + 1) / 2; a.SW = (a.W + 1) / 2; a.OH = to.H; a.OW = to.W;
                    if (to.C != 16 || a.OH != (a.SH + 1) / 2 || a.OW != (a.SW + 1) / 2) PF_FAIL(h, "detstem: inconsistent shapes");
                    a.TH = 4; a.TW = 16; a.tilesX = pf_div_up(a.OW, a.TW);
                    a.range_slot = slot_of(oi);
                    // the float-input staging loop divides i < IRH * IRW * 3 by IRW * 3 with pf_div_small (IRH = 4 TH + 3, IRW = 4 TW + 3)
                    if (!pf_div_small_domain_ok((4 * a.TH + 3) * (4 * a.TW + 3) * 3, (4 * a.TW + 3) * 3)) PF_FAIL(h, "detstem: tile %dx%d outside pf_div_small's exact range", a.TH, a.TW);
                    ProfScope ps(h, "stem_block");
                    if ((a.W & 3) || ((size_t)d_input & 3)) PF_FAIL(h, "detstem: the image width must be a multiple of 4 and the input 4-byte aligned");
                    // tile 4 x 16: stem_1 region 9 x 33 = 297 (304 rows), image region 19 rows x 67 pixels (208 halves per LDS row)
                    const dim3 sg(persistent_grid(a.tilesX * pf_div_up(a.OH, a.TH) * B, 3));     // persistent: three workgroups per CU walk the tiles
                    if (a.in_f32_nchw) PF_LAUNCH((det_stem_kernel<64, 304, 19, 208, true, 256>), sg, dim3(256), h->stream, a);
                    else PF_LAUNCH((det_stem_kernel<64, 304, 19, 208, false, 256>), sg, dim3(256), h->stream, a);
                }
                break;
            }
            case PF_OP_FRONT2: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused stem + first block op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& to = p.tens[f[0]];
                    Front2Args a{};
                    a.in = d_input; a.out = (float*)p.tensor_ptr(f[0]); a.outLd = to.ld;
                    a.w_u8 = (const pf_half*)p.cptr(f[1]); a.w_f32 = (const pf_half*)p.cptr(f[2]); a.b_stem = (const float*)p.cptr(f[3]);
                    memcpy(&a.s_u8, &f[4], 4); memcpy(&a.s_f32, &f[5], 4);
                    a.act_stem = f[6];
                    a.w_dw = (const float*)p.cptr(f[7]); a.b_dw = (const float*)p.cptr(f[8]); a.w_pw = (const float*)p.cptr(f[9]); a.b_pw = (const float*)p.cptr(f[10]);
                    a.B = B; a.H = p.hdr.in_h; a.W = p.hdr.in_w; a.OH = to.H; a.OW = to.W; a.tilesX = pf_div_up(to.W, 32);
                    a.range_slot = slot_of(oi);
                    if (to.C != 16 || (to.ld & 3) || to.H != (a.H + 1) / 2 || to.W != (a.W + 1) / 2 || (a.W & 3) || ((size_t)d_input & 3))
                        PF_FAIL(h, "front2: unsupported shapes (%dx%d input, %dx%dx%d output)", a.H, a.W, to.H, to.W, to.C);
                    ProfScope ps(h, "stem_block0");
                    const dim3 grid(a.tilesX * pf_div_up(to.H, 8), B);
                    if (input_kind == PF_INPUT_F32_NCHW) PF_LAUNCH((lm_front2_kernel<true>), grid, dim3(256), h->stream, a);
                    else PF_LAUNCH((lm_front2_kernel<false>), grid, dim3(256), h->stream, a);
                }
                break;
            }
            case PF_OP_HRB: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused Bottleneck op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    HrbArgs a{};
                    a.x = (const float*)p.tensor_ptr(f[0]); a.out = (float*)p.tensor_ptr(f[1]);
                    a.w1 = (const pf_half*)p.cptr(f[2]); a.b1 = (const float*)p.cptr(f[3]);
                    a.w2 = (const pf_half*)p.cptr(f[4]); a.b2 = (const float*)p.cptr(f[5]);
                    a.w3 = (const pf_half*)p.cptr(f[6]); a.b3 = (const float*)p.cptr(f[7]);
                    a.wd = (const pf_half*)p.cptr(f[8]); a.bd = (const float*)p.cptr(f[9]);
                    memcpy(&a.s1, &f[10], 4); memcpy(&a.s2, &f[11], 4); memcpy(&a.s3, &f[12], 4); memcpy(&a.sd, &f[13], 4);
                    const int CIN = f[14];
                    a.B = B; a.H = ti.H; a.W = ti.W; a.xLd = ti.ld; a.outLd = to.ld;
                    if (ti.C != CIN || to.C != 256 || to.H != ti.H || to.W != ti.W || (CIN == 64) != (a.wd != nullptr))
                        PF_FAIL(h, "hrb: inconsistent shapes");
                    {   // tile: <= 128 pixels (MAXP), region <= 256 pixels (one load item per thread and chunk); least halo'd pixels in total
                        long best = -1;
                        for (int tw = 128; tw >= 8; tw /= 2) {
                            const int th = 128 / tw;
                            const int twc = std::min(tw, (int)ti.W), thc = std::min(th, (int)ti.H);
                            const int region = (thc + 2) * (twc + 2);
                            if (region > 256 || thc < 1) continue;
                            const long cost = (long)pf_div_up(ti.H, thc) * pf_div_up(ti.W, twc) * (region + 64);
                            if (best < 0 || cost < best) { best = cost; a.TR = thc; a.TW = twc; }
                        }
                        if (best < 0) PF_FAIL(h, "hrb: no tile shape for a %d x %d map", ti.H, ti.W);
                    }
                    a.tiles_x = pf_div_up(ti.W, a.TW);
                    a.tpf = a.tiles_x * pf_div_up(ti.H, a.TR);
                    a.range_slot = slot_of(oi);
                    if (host_dbg(h) & 4096) {
                        if (!h->d_dbg) { PF_HIP(h, hipMalloc((void**)&h->d_dbg, 64 * 16 * sizeof(unsigned long long))); PF_HIP(h, hipMemset(h->d_dbg, 0, 64 * 16 * sizeof(unsigned long long))); }
                        a.prof = h->d_dbg + 144 + (CIN == 64 ? 0 : 4);
                    }
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "bottleneck_c%d_%dx%d", CIN, ti.H, ti.W);
                    ProfScope ps(h, tagbuf);
                    if (CIN == 64) PF_LAUNCH((hr_bottleneck_kernel<64, true, 272, 128, 1>), dim3(a.tpf * B), dim3(1024), h->stream, a);
                    else if (CIN == 256) PF_LAUNCH((hr_bottleneck_kernel<256, false, 272, 128, 1>), dim3(a.tpf * B), dim3(1024), h->stream, a);
                    else PF_FAIL(h, "hrb: no kernel for %d input channels", CIN);
                }
                break;
            }
            case PF_OP_DETC3: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused C3 op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ta = p.tens[f[0]];
                    DetC3Args a{};
                    a.srcA = (const float*)p.tensor_ptr(f[0]); a.ldA = ta.ld; a.CA = ta.C;
                    if (f[1] >= 0) { a.srcB = (const float*)p.tensor_ptr(f[1]); a.ldB = p.tens[f[1]].ld; }
                    if (f[2] >= 0) { a.out = (float*)p.tensor_ptr(f[2]); a.outLd = p.tens[f[2]].ld; }
                    if (f[3] >= 0) { a.out2 = (float*)p.tensor_ptr(f[3]); a.out2Ld = p.tens[f[3]].ld; }
                    if (f[4] >= 0) a.rows = (float*)p.buf_ptr(f[4]);
                    a.wA = (const pf_half*)p.cptr(f[5]); a.bA = (const float*)p.cptr(f[6]);
                    a.wB = (const pf_half*)p.cptr(f[7]); a.bB = (const float*)p.cptr(f[8]);
                    a.wC = (const pf_half*)p.cptr(f[9]); a.bC = (const float*)p.cptr(f[10]);
                    a.wD = (const pf_half*)p.cptr(f[11]); a.bD = (const float*)p.cptr(f[12]);
                    a.wE = (const pf_half*)p.cptr(f[13]); a.bE = (const float*)p.cptr(f[14]);
                    a.anchors = (const float*)p.cptr(f[15]);
                    memcpy(&a.sA, &f[16], 4); memcpy(&a.sB, &f[17], 4); memcpy(&a.sC, &f[18], 4); memcpy(&a.sD, &f[19], 4);
                    memcpy(&a.sE, &f[20], 4); memcpy(&a.det_stride, &f[21], 4);
                    const int CIN = f[22], tail = f[23];
                    a.upA = f[24]; a.row0 = f[25]; a.nrows_total = f[26];
                    a.B = B; a.H = ta.H << a.upA; a.W = ta.W << a.upA;
                    a.range_slot = slot_of(oi);
                    const int cb = f[1] >= 0 ? p.tens[f[1]].C : 0;
                    if (ta.C + cb != CIN || (ta.C % 8) || (f[1] >= 0 && (p.tens[f[1]].H != a.H || p.tens[f[1]].W != a.W)) ||
                        (tail == 1 && !a.out2) || (tail == 2 && (!a.rows || !a.anchors)))
                        PF_FAIL(h, "detc3: inconsistent shapes");
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "c3_c%d_t%d_%dx%d", CIN, tail, a.H, a.W);
                    ProfScope ps(h, tagbuf);
#define PF_DETC3_CASE(CC, TT, MAXR, NTHR)                                                                          \
    if (CIN == CC && tail == TT) {                                                                                 \
        det_pick_tile(h->num_cus, a.H, a.W, 1, MAXR, B, 1, &a.TH, &a.TW);                                                      \
        a.tilesX = pf_div_up(a.W, a.TW); a.tpf = a.tilesX * pf_div_up(a.H, a.TH);                                   \
        PF_LAUNCH((det_c3_kernel<CC, TT, MAXR, NTHR>), dim3(a.tpf * B), dim3(NTHR), h->stream, a); \
    } else
                    PF_DETC3_CASE(192, 1, 128, 512)
                    PF_DETC3_CASE(128, 2, 176, 512)
                    PF_FAIL(h, "detc3: no kernel for %d input channels, tail %d", CIN, tail);
#undef PF_DETC3_CASE
                }
                break;
            }
            case PF_OP_CHAIN: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "BasicBlock chain op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    ChainArgs a{};
                    a.in = (const float*)p.tensor_ptr(f[0]); a.out = (float*)p.tensor_ptr(f[1]);
                    a.B = B; a.inLd = ti.ld; a.outLd = to.ld;
                    a.n_convs = f[2];
                    const int C = f[3];
                    if (a.n_convs < 2 || a.n_convs > PF_CHAIN_MAX_CONVS || (a.n_convs & 1)) PF_FAIL(h, "chain: %d convs", a.n_convs);
                    if (ti.C != C || to.C != C || ti.H != to.H || ti.W != to.W || ti.H != ti.W) PF_FAIL(h, "chain: tensor shapes");
                    for (int c = 0; c < a.n_convs; ++c) {
                        a.wt[c] = p.cptr(f[4 + 3 * c]); a.bias[c] = (const float*)p.cptr(f[5 + 3 * c]);
                        memcpy(&a.acc_scale[c], &f[6 + 3 * c], 4);
                    }
                    a.range_slot = slot_of(oi);
                    a.dbg = h->dbg;
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "chain%d_c%d_%dx%d", a.n_convs, C, ti.H, ti.W);
                    ProfScope ps(h, tagbuf);
                    // 16 / 12 waves per workgroup and a 3 / 4-stage weight ring: measured against 8 waves and against two stages
                    // (profiles/r02_run14_teacher_*): 1.30 vs 1.38 / 1.39 ms and 0.68 vs 0.75 / 0.83 ms per 64 faces
                    if (C == 72 && ti.H == 16) PF_LAUNCH((basic_chain_kernel<72, 16, 8, 2, 3, 3>), dim3(B), dim3(1024), h->stream, a);
                    else if (C == 144 && ti.H == 8) PF_LAUNCH((basic_chain_kernel<144, 8, 4, 3, 3, 4>), dim3(B), dim3(768), h->stream, a);
                    else PF_FAIL(h, "chain: no kernel for %d channels at %dx%d", C, ti.H, ti.W);
                }
                break;
            }
            case PF_OP_BLOCK: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "BasicBlock op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    BlockArgs a{};
                    a.in = (const float*)p.tensor_ptr(f[0]); a.out = (float*)p.tensor_ptr(f[1]);
                    a.B = B; a.H = ti.H; a.inLd = ti.ld; a.outLd = to.ld; a.Cs = ti.C;
                    const int C = f[2];
                    if (to.C != ti.C || ti.H != to.H || ti.W != to.W || ti.H != ti.W || ti.C < C || ti.C >= C + 4) PF_FAIL(h, "block: tensor shapes");
                    for (int c = 0; c < 2; ++c) {
                        a.wt[c] = p.cptr(f[3 + 3 * c]); a.bias[c] = (const float*)p.cptr(f[4 + 3 * c]);
                        memcpy(&a.acc_scale[c], &f[5 + 3 * c], 4);
                    }
                    a.range_slot = slot_of(oi);
                    a.dbg = h->dbg;
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "block_c%d_%dx%d", C, ti.H, ti.W);
                    ProfScope ps(h, tagbuf);
                    if (C == 18 && ti.W == 64) PF_LAUNCH((basic_block_kernel<18, 64, 4, 7, 1>), dim3(B * (ti.H / 4)), dim3(512), h->stream, a);
                    else if (C == 36 && ti.W == 32) PF_LAUNCH((basic_block_kernel<36, 32, 8, 1, 2>), dim3(B * (ti.H / 8)), dim3(512), h->stream, a);
                    else if (C == 18 && ti.W == 16) PF_LAUNCH((basic_block_kernel<18, 16, 8, 7, 1>), dim3(B * (ti.H / 8)), dim3(512), h->stream, a);
                    else PF_FAIL(h, "block: no kernel for %d channels at %dx%d", C, ti.H, ti.W);
                }
                break;
            }
            case PF_OP_EXPDW: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused expand+depthwise op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    ConvGemmArgs a{};
                    a.in = p.tensor_ptr(f[0]); a.out = p.tensor_ptr(f[1]);
                    a.gap_out = f[2] >= 0 ? (float*)p.buf_ptr(f[2]) : nullptr;
                    a.wt = p.cptr(f[3]); a.bias = (const float*)p.cptr(f[4]);
                    a.dw_w2 = (const float*)p.cptr(f[5]); a.dw_b = (const float*)p.cptr(f[6]);
                    const int K = f[7], pad = f[8], dil = f[9];
                    a.act = f[10]; a.Cpad = f[11]; a.Npad = f[12]; a.N = f[13]; memcpy(&a.acc_scale, &f[14], 4);
                    a.B = B; a.inH = ti.H; a.inW = ti.W; a.inC = ti.C; a.inLd = ti.ld;
                    a.outH = to.H; a.outW = to.W; a.outLd = to.ld; a.outCs = 1; a.outCpad = to.C;
                    a.KH = a.KW = 1; a.stride = 1; a.pad = 0; a.dil = 1; a.store_out = 1;
                    a.dbg = h->dbg;
                    a.range_slot = slot_of(oi);
                    const int ohw = to.H * to.W;
                    const int dstride = f[15] > 0 ? f[15] : 1;
                    if (dstride == 2) {   // 64 x 64 -> 32 x 32, depthwise stride 2: whole image per workgroup, quadrant by quadrant
                        if (ti.H != 64 || ti.W != 64 || to.H != 32 || to.W != 32 || a.Cpad != 32 || (a.inC % 8) || K != 5 || dil != 1 || pad != 2 ||
                            to.C != a.N || (a.act != PF_ACT_RELU && a.act != PF_ACT_HSWISH))
                            PF_FAIL(h, "expdw(stride 2): unsupported shape");
                        char tagbuf2[96];
                        tagbuf2[0] = 0;
                        if (h->profiling) snprintf(tagbuf2, sizeof(tagbuf2), "expdw%dx%ds2_c%d_n%d_%dx%d", K, K, a.inC, a.N, to.H, to.W);
                        ProfScope ps2(h, tagbuf2);
                        if (a.act == PF_ACT_RELU) PF_LAUNCH((expdw_image_s2_kernel<5, PF_ACT_RELU>), dim3(B, pf_div_up(a.N, 16)), dim3(512), h->stream, a);
                        else PF_LAUNCH((expdw_image_s2_kernel<5>), dim3(B, pf_div_up(a.N, 16)), dim3(512), h->stream, a);
                        break;
                    }
                    if (to.H == 32 && to.W == 32 && ti.H == 32 && ti.W == 32) {   // whole 32 x 32 image per workgroup, GEMM straight from global
                        if (a.Cpad > 64 || (a.inC % 8) || pad != dil * (K - 1) / 2 || to.C != a.N || (a.act != PF_ACT_RELU && a.act != PF_ACT_HSWISH))
                            PF_FAIL(h, "expdw(32x32): unsupported shape");
                        char tagbuf2[96];
                        tagbuf2[0] = 0;
                        if (h->profiling) snprintf(tagbuf2, sizeof(tagbuf2), "expdw%dx%dd%d_c%d_n%d_%dx%d", K, K, dil, a.inC, a.N, to.H, to.W);
                        ProfScope ps2(h, tagbuf2);
                        dim3 g2(B, pf_div_up(a.N, 16));
                        if (K == 5 && dil == 1 && a.act == PF_ACT_RELU) PF_LAUNCH((expdw_image_kernel<5, 1, PF_ACT_RELU>), g2, dim3(512), h->stream, a);
                        else if (K == 5 && dil == 1) PF_LAUNCH((expdw_image_kernel<5, 1>), g2, dim3(512), h->stream, a);
                        else if (K == 3 && dil == 1) PF_LAUNCH((expdw_image_kernel<3, 1>), g2, dim3(512), h->stream, a);
                        else PF_FAIL(h, "expdw(32x32): no kernel for k%d dil %d", K, dil);
                        break;
                    }
                    if (ti.H != to.H || ti.W != to.W || to.W > 16 || (256 % ohw) != 0 || 256 / ohw > 4 || pad != dil * (K - 1) / 2 || to.C != a.N)
                        PF_FAIL(h, "expdw: unsupported shape (%dx%d, k%d pad %d dil %d)", to.H, to.W, K, pad, dil);
                    if (a.act != PF_ACT_RELU && a.act != PF_ACT_HSWISH) PF_FAIL(h, "expdw: activation must be relu or hard-swish");
                    dim3 grid(pf_div_up(B * ohw, 256), pf_div_up(a.N, 64));
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "expdw%dx%dd%d_c%d_n%d_%dx%d", K, K, dil, a.inC, a.N, to.H, to.W);
                    ProfScope ps(h, tagbuf);
                    const bool w16 = to.W == 16 && to.H == 16;      // image shape known at compile time: leaner depthwise epilogue
                    if (K == 3 && dil == 1 && w16) PF_LAUNCH((conv_gemm_split_kernel<256, 64, 8, 1, 1, 0, 3, 1, 16>), grid, dim3(512), h->stream, a);
                    else if (K == 5 && dil == 1 && w16) PF_LAUNCH((conv_gemm_split_kernel<256, 64, 8, 1, 1, 0, 5, 1, 16>), grid, dim3(512), h->stream, a);
                    else if (K == 5 && dil == 2 && w16) PF_LAUNCH((conv_gemm_split_kernel<256, 64, 8, 1, 1, 0, 5, 2, 16>), grid, dim3(512), h->stream, a);
                    else if (K == 3 && dil == 1) PF_LAUNCH((conv_gemm_split_kernel<256, 64, 8, 1, 1, 0, 3, 1>), grid, dim3(512), h->stream, a);
                    else if (K == 5 && dil == 1) PF_LAUNCH((conv_gemm_split_kernel<256, 64, 8, 1, 1, 0, 5, 1>), grid, dim3(512), h->stream, a);
                    else if (K == 5 && dil == 2) PF_LAUNCH((conv_gemm_split_kernel<256, 64, 8, 1, 1, 0, 5, 2>), grid, dim3(512), h->stream, a);
                    else PF_FAIL(h, "expdw: no kernel for k%d dil %d", K, dil);
                }
                break;
            }
            case PF_OP_MBX: {
                if constexpr (!SPLIT) {
                    PF_FAIL(h, "fused inverted-residual op needs a split-precision (f32s) program");
                } else {
                    const PfTensorRec& ti = p.tens[f[0]];
                    MbxArgs a{};
                    a.in = (const float*)p.tensor_ptr(f[0]);
                    a.out = f[1] >= 0 ? (float*)p.tensor_ptr(f[1]) : nullptr;
                    a.res = f[2] >= 0 ? (const float*)p.tensor_ptr(f[2]) : nullptr;
                    a.gap_out = f[3] >= 0 ? (float*)p.buf_ptr(f[3]) : nullptr;
                    a.gate = f[4] >= 0 ? (const float*)p.buf_ptr(f[4]) : nullptr;
                    a.w1 = (const unsigned char*)p.cptr(f[5]); a.ctile = (const float*)p.cptr(f[6]);
                    a.w2 = (const unsigned char*)p.cptr(f[7]); a.b2 = (const float*)p.cptr(f[8]);
                    const int K = f[9], pad = f[10], dil = f[11], KS = f[13], Cout = f[15], mode = f[19], nw = f[20];
                    a.act = f[12]; a.T = f[14]; a.CEXP = f[16];
                    memcpy(&a.scale1, &f[17], 4); memcpy(&a.scale2, &f[18], 4);
                    a.B = B; a.inC = ti.C; a.inLd = ti.ld;
                    a.outLd = f[1] >= 0 ? p.tens[f[1]].ld : 0; a.resLd = f[2] >= 0 ? p.tens[f[2]].ld : 0;
                    a.range_slot = slot_of(oi);
                    a.dbg = h->dbg;
                    const bool proj = mode == 0 || mode == 2, sq = mode == 1 || mode == 3;
                    if (host_dbg(h) & 64) {      // per-wave cycle accounting (ablation build; printed at pf_destroy)
                        if (!h->d_dbg) { PF_HIP(h, hipMalloc((void**)&h->d_dbg, 64 * 16 * sizeof(unsigned long long))); PF_HIP(h, hipMemset(h->d_dbg, 0, 64 * 16 * sizeof(unsigned long long))); }
                        const int shape = KS == 3 ? 0 : (KS == 4 ? (K == 3 ? 1 : 2) : 3);
                        a.prof = h->d_dbg + 160 + 8 * (shape * 4 + mode);
                    }
                    if (ti.H != 16 || ti.W != 16 || (ti.C & 3) || ti.C > 32 * KS || (ti.ld & 3) || pad != dil * (K - 1) / 2 || mode < 0 || mode > 3 ||
                        (a.act != PF_ACT_RELU && a.act != PF_ACT_HSWISH) || a.T < 1 || a.CEXP > 32 * a.T || (nw != 8 && nw != 16) ||
                        (proj && (!a.out || !a.w2 || !a.b2 || (a.outLd & 3) || p.tens[f[1]].C != Cout || (a.res && (a.resLd & 3)))) ||
                        (sq && !a.gap_out) || (mode == 2 && (!a.gate || (a.CEXP & 31))) ||      // mode 2 DMAs whole 32-float gate tiles of the face
                         (mode == 3 && (!a.out || (a.outLd & 1) || p.tens[f[1]].C < a.CEXP)))
                        PF_FAIL(h, "mbx: unsupported shape (%dx%dx%d, k%d pad %d dil %d, mode %d, %d waves)", ti.H, ti.W, ti.C, K, pad, dil, mode, nw);
                    char tagbuf[96];
                    tagbuf[0] = 0;
                    if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "mbx%s%dx%dd%d_c%d_m%d_n%d_16x16", mode == 0 ? "" : (mode == 1 ? "A" : (mode == 2 ? "B" : "S")), K, K, dil, ti.C, a.CEXP, proj ? Cout : 0);
                    ProfScope ps(h, tagbuf);
                    // One workgroup per CU, work units strided over the grid.  A unit is a face -- or, in the squeeze modes (whose channel
                    // tiles are independent), one of `nsplit` tile ranges of a face, chosen so that the last round of the persistent grid is
                    // full: 384 faces on 256 CUs are two rounds of faces (the second half empty) but three rounds of half faces.
                    a.nsplit = 1;
                    if (sq) {
                        const int cus = std::max(1, persistent_grid(1 << 20, 1));
                        double best = 1e30;
                        for (int ns = 1; ns <= 4 && ns <= a.T; ++ns) {
                            const double face_times = (double)pf_div_up(B * ns, cus) / ns + 0.02 * (ns - 1);      // (+ the input fetched ns times)
                            if (face_times < best - 1e-9) { best = face_times; a.nsplit = ns; }
                        }
                    }
                    const dim3 grid(persistent_grid(B * a.nsplit, 1));
                    const int lrc = pf_mbx_launch(a, nw, KS, Cout, K, dil, mode, (int)grid.x, h->stream);      // mbx_launch.cpp (own translation unit)
                    if (lrc > 0) PF_FAIL(h, "launch of mbx_kernel failed: %s", hipGetErrorString((hipError_t)lrc));
                    const bool launched = lrc == 0;
                    if (!launched) PF_FAIL(h, "mbx: no kernel for %d waves, KS %d Cout %d k%d dil %d mode %d", nw, KS, Cout, K, dil, mode);
                }
                break;
            }
            case PF_OP_DW: {
                const PfTensorRec& ti = p.tens[f[0]];
                const PfTensorRec& to = p.tens[f[1]];
                DwArgs a{};
                a.in = p.tensor_ptr(f[0]); a.wt = p.cptr(f[2]); a.bias = (const float*)p.cptr(f[3]);
                a.out = p.tensor_ptr(f[1]);
                a.B = B; a.inH = ti.H; a.inW = ti.W; a.C = ti.C; a.inLd = ti.ld;
                a.outH = to.H; a.outW = to.W; a.outLd = to.ld;
                a.K = f[4]; a.stride = f[5]; a.pad = f[6]; a.dil = f[7]; a.act = f[8];
                char tagbuf[64];
                tagbuf[0] = 0;
                if (h->profiling) snprintf(tagbuf, sizeof(tagbuf), "dw%dx%ds%dd%d_c%d_%dx%d", a.K, a.K, a.stride, a.dil, a.C, a.outH, a.outW);
                ProfScope ps(h, tagbuf);
                auto tgrid = [&](int tx) {
                    const long long n = (long long)B * to.H * ((to.W + tx - 1) / tx) * (ti.C / VE);
                    return dim3((unsigned)((n + 255) / 256));
                };
                if (a.K == 3 && a.stride == 1 && a.dil == 1) PF_LAUNCH((dw_conv_tiled_kernel<T, 3, 1, 1, 4>), tgrid(4), dim3(256), h->stream, a);
                else if (a.K == 3 && a.stride == 2 && a.dil == 1) PF_LAUNCH((dw_conv_tiled_kernel<T, 3, 2, 1, 4>), tgrid(4), dim3(256), h->stream, a);
                else if (a.K == 5 && a.stride == 1 && a.dil == 1) PF_LAUNCH((dw_conv_tiled_kernel<T, 5, 1, 1, 4>), tgrid(4), dim3(256), h->stream, a);
                else if (a.K == 5 && a.stride == 2 && a.dil == 1) PF_LAUNCH((dw_conv_tiled_kernel<T, 5, 2, 1, 4>), tgrid(4), dim3(256), h->stream, a);
                else if (a.K == 5 && a.stride == 1 && a.dil == 2) PF_LAUNCH((dw_conv_tiled_kernel<T, 5, 1, 2, 8>), tgrid(8), dim3(256), h->stream, a);
                else {
                    const long long total = (long long)B * to.H * to.W * (ti.C / VE);
                    PF_LAUNCH((dw_conv_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), h->stream, a);
                }
                break;
            }
            case PF_OP_UPCAT: {
                const PfTensorRec& tl = p.tens[f[0]];
                const PfTensorRec& tk = p.tens[f[1]];
                const PfTensorRec& to = p.tens[f[2]];
                UpcatArgs a{};
                a.lo = p.tensor_ptr(f[0]); a.skip = p.tensor_ptr(f[1]); a.out = p.tensor_ptr(f[2]);
                a.B = B; a.loH = tl.H; a.loW = tl.W; a.C1 = tl.C; a.loLd = tl.ld;
                a.C2 = tk.C; a.skipLd = tk.ld; a.outLd = to.ld;
                const long long total = (long long)B * to.H * to.W * ((tl.C + tk.C) / VE);
                ProfScope ps(h, "upsample_concat");
                PF_LAUNCH((upsample_concat_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_GAP: {
                const PfTensorRec& ti = p.tens[f[0]];
                GapArgs a{};
                a.in = p.tensor_ptr(f[0]); a.out = (float*)p.buf_ptr(f[1]);
                a.B = B; a.HW = ti.H * ti.W; a.C = ti.C; a.ld = ti.ld;
                ProfScope ps(h, "gap");
                PF_LAUNCH((gap_kernel<T>), dim3(pf_div_up(ti.C / VE, 8), B), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_FC: {
                FcArgs a{};
                a.x = (const float*)p.buf_ptr(f[0]); a.y = (float*)p.buf_ptr(f[1]);
                a.wt = (const float*)p.cptr(f[2]); a.bias = (const float*)p.cptr(f[3]);
                a.B = B; a.K = f[4]; a.N = f[5]; a.act = f[6];
                a.scale2 = (const float*)p.cptr(f[7]); a.shift2 = (const float*)p.cptr(f[8]); a.act2 = f[9];
                ProfScope ps(h, "fc");
                if (a.K <= PF_FC_MAXK) PF_LAUNCH(fc_kernel<true>, dim3(pf_div_up(a.N, PF_FC_BN), pf_div_up(B, PF_FC_BB)), dim3(256), h->stream, a);
                else PF_LAUNCH(fc_kernel<false>, dim3(pf_div_up(a.N, PF_FC_BN), pf_div_up(B, PF_FC_BB)), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_FC2: {
                Fc2Args a{};
                a.x = (const float*)p.buf_ptr(f[0]); a.y = (float*)p.buf_ptr(f[1]);
                a.w1 = (const float*)p.cptr(f[2]); a.b1 = (const float*)p.cptr(f[3]);
                a.K = f[4]; a.R = f[5]; a.act1 = f[6];
                a.scale2 = (const float*)p.cptr(f[7]); a.shift2 = (const float*)p.cptr(f[8]); a.act1b = f[9];
                a.w2 = (const float*)p.cptr(f[10]); a.b2 = (const float*)p.cptr(f[11]); a.N = f[12]; a.act2 = f[13];
                a.B = B;
                a.nparts = f[14] > 0 ? f[14] : 1; memcpy(&a.xscale, &f[15], 4);
                if (a.nparts == 1) a.xscale = 1.f;
                if (a.K < 1 || a.K > 960 || a.R < 4 || a.R > 960 || (a.R & 3) || a.N < 4 || a.N > 960 || (a.N & 3) || !a.w1 || !a.w2 || (a.scale2 && !a.shift2))
                    PF_FAIL(h, "fc2: unsupported shape %d -> %d -> %d", a.K, a.R, a.N);
                ProfScope ps(h, "fc");
                PF_LAUNCH(fc2_kernel, dim3(pf_div_up(B, PF_FC2_FB)), dim3(1024), h->stream, a);
                break;
            }
            case PF_OP_SCSE: {
                const PfTensorRec& ti = p.tens[f[0]];
                const PfTensorRec& to = p.tens[f[1]];
                ScseArgs a{};
                a.in = p.tensor_ptr(f[0]); a.out = p.tensor_ptr(f[1]);
                a.cse = (const float*)p.buf_ptr(f[2]); a.sse_w = (const float*)p.cptr(f[3]);
                memcpy(&a.sse_b, &f[4], 4);
                a.B = B; a.HW = ti.H * ti.W; a.C = ti.C; a.ld = ti.ld; a.outLd = to.ld;
                const int lpp = ti.C / VE;
                if (lpp < 1 || lpp > 64 || (lpp & (lpp - 1))) PF_FAIL(h, "scse: C/VE=%d must be a power of two <= 64", lpp);
                const long long total = (long long)B * a.HW;
                ProfScope ps(h, "scse");
                PF_LAUNCH((scse_kernel<T>), dim3((unsigned)((total + 256 / lpp - 1) / (256 / lpp))), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_HMDEC: {
                const PfTensorRec& tf = p.tens[f[2]];
                HmDecodeArgs a{};
                a.amax_val = (const float*)p.buf_ptr(f[0]); a.amax_idx = (const int*)p.buf_ptr(f[1]);
                a.feat = p.tensor_ptr(f[2]); a.off_wt = (const float*)p.cptr(f[3]); a.off_bias = (const float*)p.cptr(f[4]);
                a.P = f[5]; a.nslots = f[6];
                a.loc = (float*)p.buf_ptr(f[7]); a.score = (float*)p.buf_ptr(f[8]);
                a.crop = h->pipe.d_crop_for_decode; a.kps = h->pipe.d_kps_for_decode;
                a.B = B; a.H = tf.H; a.W = tf.W; a.C = tf.C; a.featLd = tf.ld;
                ProfScope ps(h, "hm_decode");
                PF_LAUNCH((hm_decode_kernel<T>), dim3(pf_div_up(B * a.P, 4)), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_ADDUP: {
                const PfTensorRec& ta = p.tens[f[0]];
                const PfTensorRec& tb = p.tens[f[1]];
                const PfTensorRec& to = p.tens[f[2]];
                AddUpArgs a{};
                a.a = p.tensor_ptr(f[0]); a.b = p.tensor_ptr(f[1]); a.out = p.tensor_ptr(f[2]);
                a.B = B; a.H = ta.H; a.W = ta.W; a.C = ta.C; a.aLd = ta.ld; a.bLd = tb.ld; a.outLd = to.ld;
                a.shift = f[3]; a.act = f[4];
                if ((tb.H << a.shift) != ta.H || (tb.W << a.shift) != ta.W || tb.C != ta.C || to.C != ta.C)
                    PF_FAIL(h, "addup: inconsistent shapes");
                const long long total = (long long)B * ta.H * ta.W * (ta.C / VE);
                ProfScope ps(h, "add_upsample");
                PF_LAUNCH((add_upsample_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_FUSEUP: {
                if constexpr (sizeof(T) != 4) {
                    PF_FAIL(h, "fused HRNet fuse sum needs f32 tensors");
                } else {
                    const PfTensorRec& ty = p.tens[f[0]];
                    const PfTensorRec& to = p.tens[f[1]];
                    FuseUpArgs a{};
                    a.y = (const float*)p.tensor_ptr(f[0]); a.out = (float*)p.tensor_ptr(f[1]);
                    a.B = B; a.H = ty.H; a.W = ty.W; a.Cs = ty.C; a.C = f[4 + 12]; a.yLd = ty.ld; a.outLd = to.ld; a.act = f[2]; a.nsrc = f[3];
                    if (a.nsrc < 1 || a.nsrc > 3 || to.H != ty.H || to.W != ty.W || to.C != ty.C || (ty.C & 3) || a.C > a.Cs) PF_FAIL(h, "fuseup: inconsistent shapes");
                    int need = 0;
                    for (int s = 0; s < a.nsrc; ++s) {
                        const PfTensorRec& ts = p.tens[f[4 + 4 * s]];
                        a.src[s] = (const float*)p.tensor_ptr(f[4 + 4 * s]); a.wt[s] = (const float*)p.cptr(f[5 + 4 * s]); a.bias[s] = (const float*)p.cptr(f[6 + 4 * s]);
                        a.shift[s] = f[7 + 4 * s]; a.srcLd[s] = ts.ld; a.srcC[s] = ts.C;
                        if (a.shift[s] < 1 || a.shift[s] > 3 || (ts.H << a.shift[s]) != ty.H || (ts.W << a.shift[s]) != ty.W || (ts.C & 3))
                            PF_FAIL(h, "fuseup: source %d does not match the output", s);
                        const int r = std::max(1, 16 >> a.shift[s]);
                        need += ts.C * a.Cs + r * r * (ts.C + a.Cs);
                    }
                    ProfScope ps(h, "fuse_up");
                    const int ntiles = B * pf_div_up(ty.H, 16) * pf_div_up(ty.W, 16);
                    // (512-thread workgroups with y requested at the head of a tile and an 80 KB middle tier: 1.59 ms for the Teacher's 18
                    // launches against 1.18 ms in this form, profiles/r04_run29 / r04_run30)
                    if (need <= 12288) PF_LAUNCH((fuse_up_kernel<12288>), dim3(persistent_grid(ntiles, 3)), dim3(256), h->stream, a);        // 48 KB: three per CU
                    else if (need <= 20480) PF_LAUNCH((fuse_up_kernel<20480>), dim3(persistent_grid(ntiles, 2)), dim3(256), h->stream, a);   // 80 KB: two
                    else if (need <= 24576) PF_LAUNCH((fuse_up_kernel<24576>), dim3(persistent_grid(ntiles, 1)), dim3(256), h->stream, a);
                    else PF_FAIL(h, "fuseup: %d floats of LDS needed", need);
                }
                break;
            }
            case PF_OP_MAXPOOL: {
                const PfTensorRec& ti = p.tens[f[0]];
                const PfTensorRec& to = p.tens[f[1]];
                PoolArgs a{};
                a.in = p.tensor_ptr(f[0]); a.out = p.tensor_ptr(f[1]);
                a.B = B; a.inH = ti.H; a.inW = ti.W; a.C = ti.C; a.inLd = ti.ld;
                a.outH = to.H; a.outW = to.W; a.outLd = to.ld;
                const long long total = (long long)B * to.H * to.W * (ti.C / VE);
                ProfScope ps(h, "maxpool");
                PF_LAUNCH((maxpool2_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_COPY: {
                const PfTensorRec& ti = p.tens[f[0]];
                const PfTensorRec& to = p.tens[f[1]];
                CopyArgs a{};
                a.in = p.tensor_ptr(f[0]); a.out = p.tensor_ptr(f[1]);
                a.B = B; a.inH = ti.H; a.inW = ti.W; a.C = ti.C; a.inLd = ti.ld; a.outLd = to.ld;
                a.outCs = f[2]; a.up = f[3];
                const long long total = (long long)B * ti.H * a.up * ti.W * a.up * (ti.C / VE);
                ProfScope ps(h, "copy_channels");
                PF_LAUNCH((copy_channels_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), h->stream, a);
                break;
            }
            case PF_OP_DETDEC: {
                const PfTensorRec& ti = p.tens[f[0]];
                DetDecArgs a{};
                a.in = p.tensor_ptr(f[0]); a.rows = (float*)p.buf_ptr(f[1]);
                a.row0 = f[2]; memcpy(&a.stride, &f[3], 4); a.anchors = (const float*)p.cptr(f[4]);
                a.nrows_total = f[5];
                a.B = B; a.ny = ti.H; a.nx = ti.W; a.ld = ti.ld;
                const long long total = (long long)B * 3 * ti.H * ti.W;
                ProfScope ps(h, "detect_decode");
                PF_LAUNCH((detect_decode_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), h->stream, a);
                break;
            }
            default:
                PF_FAIL(h, "unknown op code %d at op %zu", op.code, oi);
        }
    }
    if (guard) {
        RangeVerdictArgs v{};
        v.slots = p.d_range; v.n_ops = (int)p.ops.size();
        v.lo = 0.0009765625f;            // 2^-10: below this a tensor's low halves sit in the f16 subnormal range
        v.hi = 6.0e4f;                   // f16 overflows at 65504
        v.status = h->h_status; v.prog_slot = slot;
        v.poison0 = (float*)p.buf_ptr(p.hdr.out_buf0); v.n0 = (long long)B * p.bufs[p.hdr.out_buf0].elems_per_item;
        if (p.hdr.out_buf1 >= 0 && p.hdr.out_buf1 != p.hdr.out_buf0) { v.poison1 = (float*)p.buf_ptr(p.hdr.out_buf1); v.n1 = (long long)B * p.bufs[p.hdr.out_buf1].elems_per_item; }
        if (h->pipe.d_kps_for_decode) { v.poison2 = h->pipe.d_kps_for_decode; v.n2 = (long long)B * 98 * 2; }
        PF_LAUNCH(range_verdict_kernel, dim3(v.n_ops), dim3(64), h->stream, v);
    }
    PF_HIP(h, hipGetLastError());
    return 0;
}

// Start of every forward-running entry point
static void begin_call(pf_handle* h) { h->n_calls++; }

// After a stream synchronisation: did a range-checked forward find a tensor the f32s kernels cannot represent?
static int check_numerics(pf_handle* h) {
    if (!h->h_status || h->h_status[0] == 0) return 0;
    const int code = h->h_status[0], op = h->h_status[1], slot = h->h_status[3];
    float v;
    memcpy(&v, &h->h_status[2], 4);
    h->h_status[0] = 0;
    if (code == 3)
        PF_FAIL(h, "pf_decode_jpeg_batch: the parallel entropy decoder did not synchronise (%d sub-sequence records still changing "
                   "after the last round): the frames of that batch are invalid; decode it again with pf_set_option(PF_OPT_JPEG_ENTROPY, 1)", op);
    PF_FAIL(h, "activation range check failed: input of op %d of program %d has max |x| = %g, %s the range [9.8e-4, 6e4] the "
               "split-precision (f32s) convolutions can represent (outputs were set to NaN); rebuild the program with dtype 'f32'",
            op, slot, (double)v, code == 1 ? "above" : "below");
}

static int run_program(pf_handle* h, int slot, const void* d_input, int input_kind, int B) {
    Program& p = h->prog[slot];
    if (!p.loaded) PF_FAIL(h, "no program loaded in slot %d", slot);
    if (B < 1 || B > p.max_batch) PF_FAIL(h, "batch %d outside [1, %d]", B, p.max_batch);
    if (p.hdr.dtype == PF_DTYPE_F16) return run_program_t<pf_half, false>(h, slot, d_input, input_kind, B);
    if (p.hdr.dtype == PF_DTYPE_F32_SPLIT) return run_program_t<float, true>(h, slot, d_input, input_kind, B);
    return run_program_t<float, false>(h, slot, d_input, input_kind, B);
}

static int ensure_stage(pf_handle* h, size_t bytes) {
    if (bytes <= h->stage_bytes) return 0;
    if (h->d_stage) (void)hipFree(h->d_stage);
    h->d_stage = nullptr;
    h->stage_bytes = 0;
    PF_HIP(h, hipMalloc((void**)&h->d_stage, bytes));
    h->stage_bytes = bytes;
    return 0;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* pf_version(void) { return "peppa-hip 0.1 " PF_BUILD_TAG; }

int pf_create(int device_id, pf_handle** out) {
    if (!out) return 1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_create_error = "no HIP device visible (the engine has no CPU fallback)"; return 1; }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return 1; }
    if (hipSetDevice(device_id) != hipSuccess) { g_create_error = "hipSetDevice failed"; return 1; }
    pf_handle* h = new pf_handle();
    h->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus >= 8) h->num_cus = cus;
    }
    if constexpr (PF_ABLATE != 0) {      // ablation build only (libpeppa_hip_ablate.so): ablated kernels compute garbage, so the guard is off
        if (const char* v = getenv("PEPPA_DBG")) { h->dbg = atoi(v); if (h->dbg & 0xffff) h->range_every = 0; }   // bits >= 16 only re-schedule (wave priorities): results stay right, the guard stays on
        if (const char* v = getenv("PEPPA_DET_TILE")) { if (sscanf(v, "%d,%d", &g_det_tile_th, &g_det_tile_tw) != 2) g_det_tile_th = g_det_tile_tw = 0; }
    }
    bool masked = false;
    if constexpr (PF_ABLATE != 0) {      // experiment: every handle of the process on its own share of the CUs (PEPPA_CU_PARTITION=parts,mode)
        static int s_lane = 0;
        int parts = 0, mode = 1;
        if (const char* v = getenv("PEPPA_CU_PARTITION")) (void)sscanf(v, "%d,%d", &parts, &mode);
        if (parts > 1) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int lane = s_lane++ % parts, per = h->num_cus / parts;
            for (int c = 0; c < h->num_cus; ++c)
                if ((mode == 1 && c % parts == lane) || (mode == 2 && c / per == lane) || (mode == 3 && (c / 8) % parts == lane)) mask[c >> 5] |= 1u << (c & 31);
            masked = hipExtStreamCreateWithCUMask(&h->stream, 8, mask) == hipSuccess;
            fprintf(stderr, "[peppa-hip] handle %d: CU partition %d/%d mode %d: %s\n", s_lane - 1, lane, parts, mode, masked ? "ok" : "FAILED");
        }
    }
    if ((!masked && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        g_create_error = "stream/event creation failed";
        delete h;
        return 1;
    }
    if (hipHostMalloc((void**)&h->h_status, 4 * sizeof(int), hipHostMallocPortable) != hipSuccess) {
        g_create_error = "hipHostMalloc(status) failed";
        delete h;
        return 1;
    }
    memset(h->h_status, 0, 4 * sizeof(int));
    *out = h;
    return 0;
}

void pf_destroy(pf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (auto& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    comm_release(h);
    for (auto& p : h->prog) {
        if (p.d_const) (void)hipFree(p.d_const);
        if (p.d_arena) (void)hipFree(p.d_arena);
        if (p.d_range) (void)hipFree(p.d_range);
    }
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->d_dbg) {
        unsigned long long u[48];
        if (hipMemcpy(u, h->d_dbg + 64, sizeof(u), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int k = 0; k < 6; ++k) {
                const unsigned long long* q = u + 8 * k;
                if (!q[4]) continue;
                const double n = (double)q[4];
                fprintf(stderr, "[det_unit C=%d S=%d] per workgroup (cycles): input+split %.0f | gemm1 %.0f | depthwise %.0f | gemm2+store %.0f  (%.0f workgroups)\n",
                        k % 3 == 0 ? 32 : (k % 3 == 1 ? 64 : 128), k / 3 + 1, q[0] / n, q[1] / n, q[2] / n, q[3] / n, n);
            }
        }
        unsigned long long hb[8];
        if (hipMemcpy(hb, h->d_dbg + 144, sizeof(hb), hipMemcpyDeviceToHost) == hipSuccess)
            for (int k = 0; k < 2; ++k)
                if (hb[4 * k + 3])
                    fprintf(stderr, "[det_hr_bottleneck CIN=%d] per workgroup (cycles): conv1 %.0f | conv2 %.0f | conv3+store %.0f  (%.0f workgroups)\n", k ? 256 : 64,
                            hb[4 * k] / (double)hb[4 * k + 3], hb[4 * k + 1] / (double)hb[4 * k + 3], hb[4 * k + 2] / (double)hb[4 * k + 3], (double)hb[4 * k + 3]);
        unsigned long long mb[128];
        if (hipMemcpy(mb, h->d_dbg + 160, sizeof(mb), hipMemcpyDeviceToHost) == hipSuccess)
            for (int k = 0; k < 16; ++k) {
                const unsigned long long* q = mb + 8 * k;
                if (!q[7]) continue;
                static const char* shp[4] = {"KS3 k3", "KS4 k3", "KS4 k5", "KS5 k5d2"};
                const double n = (double)q[7];
                fprintf(stderr, "[det_mbx %s mode %d] per wave and launch-face (cycles): prologue+expand0 %.0f | project %.0f | wait a %.0f | depthwise %.0f | expand %.0f | wait b %.0f | epilogue %.0f  (%.0f waves)\n",
                        shp[k / 4], k % 4, q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[4] / n, q[5] / n, q[6] / n, n);
            }
        unsigned long long v[32];
        if (hipMemcpy(v, h->d_dbg, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int k = 0; k < 2; ++k) {
                const unsigned long long* q = v + 16 * k;
                if (!q[2]) continue;
                const double steps = (double)q[3] / (double)q[2];          // K steps per wave
                fprintf(stderr, "[sepup_pipe BN=%d] per wave and K step (cycles): producer dma %.0f work %.0f wait %.0f | consumer dma %.0f mfma %.0f epilogue %.0f wait %.0f  (%.0f steps/wave)\n",
                        k ? 256 : 128, q[9] / (double)q[2] / steps, q[0] / (double)q[2] / steps, q[1] / (double)q[2] / steps, q[4] / (double)q[8] / steps, q[5] / (double)q[8] / steps,
                        q[6] / (double)q[8] / steps, q[7] / (double)q[8] / steps, steps);
            }
        }
        (void)hipFree(h->d_dbg);
    }
    if (h->h_status) (void)hipHostFree(h->h_status);
    h->pipe.release();
    h->track.release();
    h->jpeg.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* pf_last_error(pf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pf_sync(pf_handle* h) {
    if (!h) return 1;
    PF_HIP(h, hipStreamSynchronize(h->stream));
    return check_numerics(h);
}

int pf_load_program(pf_handle* h, int slot, const void* blob, size_t bytes, int max_batch) {
    if (!h) return 1;
    if (slot < 0 || slot >= PF_NET_SLOTS) PF_FAIL(h, "slot %d out of range", slot);
    if (!blob || bytes < sizeof(PfHeader)) PF_FAIL(h, "program blob too small");
    if (max_batch < 1) PF_FAIL(h, "max_batch must be >= 1");
    PF_HIP(h, hipSetDevice(h->device));
    PfHeader hd;
    memcpy(&hd, blob, sizeof(hd));
    if (hd.magic != PF_PROGRAM_MAGIC) PF_FAIL(h, "bad program magic 0x%08x", hd.magic);
    if (hd.version != PF_PROGRAM_VERSION) PF_FAIL(h, "program version %d, engine expects %d", hd.version, PF_PROGRAM_VERSION);
    if (hd.dtype != PF_DTYPE_F16 && hd.dtype != PF_DTYPE_F32 && hd.dtype != PF_DTYPE_F32_SPLIT) PF_FAIL(h, "bad dtype %d", hd.dtype);
    size_t off = sizeof(PfHeader);
    const size_t need = off + (size_t)hd.n_bufs * sizeof(PfBufRec) + (size_t)hd.n_tensors * sizeof(PfTensorRec) +
                        (size_t)hd.n_ops * sizeof(PfOpRec);
    if (bytes < need) PF_FAIL(h, "program blob truncated (tables)");
    Program& p = h->prog[slot];
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (p.d_const) { (void)hipFree(p.d_const); p.d_const = nullptr; }
    if (p.d_arena) { (void)hipFree(p.d_arena); p.d_arena = nullptr; }
    if (p.d_range) { (void)hipFree(p.d_range); p.d_range = nullptr; }
    h->alloc_epoch++;            // graphs captured over the old arena / constants must not be replayed
    p.loaded = false;
    p.hdr = hd;
    p.esize = hd.dtype == PF_DTYPE_F16 ? 2 : 4;
    const char* src = (const char*)blob;
    p.bufs.resize(hd.n_bufs);
    memcpy(p.bufs.data(), src + off, (size_t)hd.n_bufs * sizeof(PfBufRec)); off += (size_t)hd.n_bufs * sizeof(PfBufRec);
    p.tens.resize(hd.n_tensors);
    memcpy(p.tens.data(), src + off, (size_t)hd.n_tensors * sizeof(PfTensorRec)); off += (size_t)hd.n_tensors * sizeof(PfTensorRec);
    p.ops.resize(hd.n_ops);
    memcpy(p.ops.data(), src + off, (size_t)hd.n_ops * sizeof(PfOpRec)); off += (size_t)hd.n_ops * sizeof(PfOpRec);
    off = (off + 255) / 256 * 256;
    if (bytes < off + (size_t)hd.const_bytes) PF_FAIL(h, "program blob truncated (constants)");
    // validate indices once so the hot path can trust them
    for (const auto& t : p.tens)
        if (t.buf < 0 || t.buf >= hd.n_bufs) PF_FAIL(h, "tensor references buffer %d", t.buf);
    for (int b = 0; b < hd.n_bufs; ++b) {
        const size_t units = (p.buf_item_bytes(b) + 255) / 256;
        if ((size_t)p.bufs[b].offset_units + units > (size_t)hd.arena_units_per_item) PF_FAIL(h, "buffer %d outside the arena", b);
    }
    PF_HIP(h, hipMalloc((void**)&p.d_const, std::max<size_t>(hd.const_bytes, 256)));
    PF_HIP(h, hipMemcpy(p.d_const, src + off, hd.const_bytes, hipMemcpyHostToDevice));
    p.max_batch = max_batch;
    p.arena_bytes = (size_t)hd.arena_units_per_item * 256 * (size_t)max_batch;
    PF_HIP(h, hipMalloc((void**)&p.d_arena, p.arena_bytes + 256));          // + slack: masked pixel-operand units may read up to 124 bytes behind a tensor
    PF_HIP(h, hipMemset(p.d_arena, 0, p.arena_bytes + 256));
    if (hd.dtype == PF_DTYPE_F32_SPLIT) {
        const size_t rb = std::max<size_t>(hd.n_ops, 1) * PF_RANGE_OP_WORDS * sizeof(unsigned);
        PF_HIP(h, hipMalloc((void**)&p.d_range, rb + PF_RANGE_TAIL_WORDS * sizeof(unsigned)));
        PF_HIP(h, hipMemset(p.d_range, 0, rb + PF_RANGE_TAIL_WORDS * sizeof(unsigned)));
        PF_HIP(h, hipMemset((char*)p.d_range + rb, 0xFF, 8));       // the verdict key: all ones = no violation (range_verdict_kernel)
    }
    p.loaded = true;
    return 0;
}

static int net_forward_common(pf_handle* h, int slot, const void* input, int input_kind, int mem, int batch,
                              size_t in_item_bytes) {
    PF_HIP(h, hipSetDevice(h->device));
    const void* d_in = input;
    if (mem == PF_MEM_HOST) {
        if (ensure_stage(h, in_item_bytes * batch)) return 1;
        PF_HIP(h, hipMemcpyAsync(h->d_stage, input, in_item_bytes * batch, hipMemcpyHostToDevice, h->stream));
        d_in = h->d_stage;
    } else if ((size_t)input & 3) {
        // the staged-image kernels (k_det.h det_stem_kernel, k_front.h) fetch the uint8 image as 32-bit words: a device pointer that is
        // not 4-byte aligned (a view into somebody else's buffer) goes through the aligned staging buffer first
        if (ensure_stage(h, in_item_bytes * batch)) return 1;
        PF_HIP(h, hipMemcpyAsync(h->d_stage, input, in_item_bytes * batch, hipMemcpyDeviceToDevice, h->stream));
        d_in = h->d_stage;
    }
    return run_program(h, slot, d_in, input_kind, batch);
}

static int copy_out(pf_handle* h, const void* d_src, void* dst, size_t bytes, int out_mem) {
    if (!dst) return 0;
    PF_HIP(h, hipMemcpyAsync(dst, d_src, bytes, out_mem == PF_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int pf_landmark_forward(pf_handle* h, const void* input, int input_kind, int mem, int batch,
                        float* loc_fix, float* score, int out_mem) {
    if (!h) return 1;
    Program& p = h->prog[PF_NET_LANDMARK];
    if (!p.loaded) PF_FAIL(h, "landmark program not loaded");
    const size_t px = (size_t)p.hdr.in_h * p.hdr.in_w * 3;
    h->pipe.d_crop_for_decode = nullptr;
    h->pipe.d_kps_for_decode = nullptr;
    begin_call(h);
    if (net_forward_common(h, PF_NET_LANDMARK, input, input_kind, mem, batch, input_kind == PF_INPUT_U8_NHWC ? px : px * 4)) return 1;
    if (copy_out(h, p.buf_ptr(p.hdr.out_buf0), loc_fix, (size_t)batch * p.buf_item_bytes(p.hdr.out_buf0), out_mem)) return 1;
    if (copy_out(h, p.buf_ptr(p.hdr.out_buf1), score, (size_t)batch * p.buf_item_bytes(p.hdr.out_buf1), out_mem)) return 1;
    if (out_mem == PF_MEM_HOST) {
        PF_HIP(h, hipStreamSynchronize(h->stream));
        return check_numerics(h);
    }
    return 0;
}

int pf_detector_forward(pf_handle* h, const void* input, int input_kind, int mem, int batch, float* rows_out, int out_mem) {
    if (!h) return 1;
    Program& p = h->prog[PF_NET_DETECTOR];
    if (!p.loaded) PF_FAIL(h, "detector program not loaded");
    const size_t px = (size_t)p.hdr.in_h * p.hdr.in_w * 3;
    begin_call(h);
    if (net_forward_common(h, PF_NET_DETECTOR, input, input_kind, mem, batch, input_kind == PF_INPUT_U8_NHWC ? px : px * 4)) return 1;
    if (copy_out(h, p.buf_ptr(p.hdr.out_buf0), rows_out, (size_t)batch * p.buf_item_bytes(p.hdr.out_buf0), out_mem)) return 1;
    if (out_mem == PF_MEM_HOST) {
        PF_HIP(h, hipStreamSynchronize(h->stream));
        return check_numerics(h);
    }
    return 0;
}

int pf_read_tensor(pf_handle* h, int slot, int tensor_id, int batch, float* out_host, size_t out_elems) {
    if (!h) return 1;
    if (slot < 0 || slot >= PF_NET_SLOTS || !h->prog[slot].loaded) PF_FAIL(h, "slot %d not loaded", slot);
    Program& p = h->prog[slot];
    if (tensor_id < 0 || tensor_id >= (int)p.tens.size()) PF_FAIL(h, "tensor id %d out of range", tensor_id);
    const PfTensorRec& t = p.tens[tensor_id];
    const size_t n = (size_t)batch * t.H * t.W * t.C;
    if (out_elems < n) PF_FAIL(h, "output too small: %zu < %zu", out_elems, n);
    PF_HIP(h, hipStreamSynchronize(h->stream));
    const size_t item_elems = (size_t)t.H * t.W * t.ld;
    std::vector<char> tmp(item_elems * p.esize * batch);
    PF_HIP(h, hipMemcpy(tmp.data(), p.tensor_ptr(tensor_id) - (size_t)t.coff * p.esize, tmp.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b)
        for (size_t px = 0; px < (size_t)t.H * t.W; ++px)
            for (int c = 0; c < t.C; ++c) {
                const size_t si = (size_t)b * item_elems + px * t.ld + t.coff + c;
                float v;
                if (p.esize == 2) v = (float)((const pf_half*)tmp.data())[si];
                else v = ((const float*)tmp.data())[si];
                out_host[((size_t)b * t.H * t.W + px) * t.C + c] = v;
            }
    return 0;
}

int pf_profile_enable(pf_handle* h, int on) {
    if (!h) return 1;
    h->profiling = on != 0;
    h->prof.clear();
    h->prof_order.clear();
    return 0;
}

int pf_profile_fetch(pf_handle* h, char* names, size_t names_cap, float* ms, int* counts, int cap, int* n_out) {
    if (!h) return 1;
    std::string joined;
    int n = 0;
    for (const auto& tag : h->prof_order) {
        if (n >= cap) break;
        const ProfEntry& e = h->prof[tag];
        if (ms) ms[n] = (float)e.ms;
        if (counts) counts[n] = e.count;
        joined += tag;
        joined += '\n';
        ++n;
    }
    if (names && names_cap) {
        const size_t c = std::min(names_cap - 1, joined.size());
        memcpy(names, joined.data(), c);
        names[c] = 0;
    }
    if (n_out) *n_out = n;
    return 0;
}

}  // extern "C"

#include "pipeline.inl"
#include "comm.inl"
#include "track.inl"
#include "jpeg.inl"
#include "batch.inl"
