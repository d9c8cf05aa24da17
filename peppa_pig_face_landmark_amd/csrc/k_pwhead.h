// The heat-map score head of the landmark regressor (COTRAIN's `hm` conv, model.py:165-172 / 511-554: 1x1, 128 -> 98 score maps
// at 64 x 64, then the per-landmark arg-max) as a WEIGHT-STATIONARY stream (round 5).
//
// The generic pointwise kernel (conv_gemm_split_kernel, one 128 x 128 tile per workgroup) runs this layer at 2.2 TB/s of its
// 537 MB input: 8192 short-lived workgroups, each paying a first-load round trip with nothing to overlap it and re-staging the
// SAME 57 KB of weights through LDS -- as many bytes again as the activations.  Here one persistent workgroup per CU keeps its
// weights on chip for the whole launch -- the hi halves of a wave's fragments in REGISTERS (wave = 32 pixels x 64 channels of a
// 128 x 128 tile: 4 channel tiles x 4 K steps = 64 VGPRs), the lo halves and the bias in LDS (28.5 KB) -- and walks the pixel
// tiles with the next tile's 64 KB in flight in registers while this tile's MFMAs run; the rest of the LDS is two pixel-operand
// stages (all four K steps of a tile, hi and lo planes, same chunk rotation as everywhere).  One barrier per tile.  NOTHING in the loop waits on the vector-memory counter except the
// park of the tile requested a whole iteration earlier: vmcnt retires in order, so a bias load or a scratch reload behind the
// prefetch would wait for all of it (the first cut did both and ran at 10 us per tile instead of 3).
// Same products in the same order per accumulator as the generic kernel, same epilogue arithmetic: bit-identical landmarks.
// Measured per 256 faces (profiles/r05_run32 ... r05_run36_pw_head_ab.txt): generic 0.228 ms; this kernel with the per-tile arg-max
// exchange 0.335 (bias loads and scratch reloads behind the prefetch), 0.228 (those fixed: VALU-bound by the exchange), 0.148 with the
// arg-max carried per lane across a face (below).
#pragma once
#include "k_conv_gemm.h"

// ONEPROD (round 6, opt-in like conv3x3_hero_kernel's): one f16 product per 32 k -- no lo planes, no lo weight halves, a third of the MFMAs.
template <int NK, bool ONEPROD = false>
__global__ __launch_bounds__(512, 2) void pw_head_kernel(ConvGemmArgs a) {
    constexpr int BM = 128, BN = 128, WARPS_M = 4, WARPS_N = 2;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N, MT = WM / 16, NT = WN / 16;
    constexpr int PLANE = BM * 64;                  // one K step of one precision half: 128 rows x 64 B
    constexpr int STAGE = 2 * NK * PLANE;           // hi planes of the NK steps, then the lo planes
    constexpr int MAXN = 112;                       // weight rows kept (host: Npad <= MAXN)
    constexpr int WL_BYTES = NK * MAXN * 64;
    static_assert(2 * STAGE + WL_BYTES + BN * 4 <= 160 * 1024, "two pixel stages + the lo weights + the bias must fit the LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char s_wl[WL_BYTES];      // [K step][row][64 B, chunks rotated]
    __shared__ __attribute__((aligned(16))) float s_bias[BN];
    PF_EMU_POISON(smem);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int OHW = a.outH * a.outW;
    // host: OHW % BM == 0 (a tile never straddles a face), Cpad == 32 NK
    const float* __restrict__ in = static_cast<const float*>(a.in);
    unsigned amax = 0;                              // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);

    // ---- this wave's weight fragments, once: rows [wn WN + 16 j, + 16), K step s, hi | lo (rows past Npad repeat the last one:
    // their results are never written) ----------------------------------------------------------------------------------------
    pf_half8 whf[NT][NK];
    int wloff[NT];                                  // this lane's lo fragment of channel tile j inside a K step's block of s_wl
    {
        const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = min(wn * WN + j * 16 + frow, a.Npad - 1);
            const unsigned char* p = wt + (size_t)n * (NK * 128) + fchunk * 16;
#pragma unroll
            for (int s = 0; s < NK; ++s) whf[j][s] = *reinterpret_cast<const pf_half8*>(p + s * 128);
            wloff[j] = pf_lds_chunk_off(n, fchunk);
        }
        for (int i = t; i < NK * MAXN * 4; i += 512) {              // 16-byte chunks of the lo halves
            const int s = i / (MAXN * 4), r = (i - s * MAXN * 4) >> 2, c = i & 3;
            const int n = min(r, a.Npad - 1);
            *reinterpret_cast<pf_half8*>(s_wl + s * (MAXN * 64) + pf_lds_chunk_off(r, c)) =
                *reinterpret_cast<const pf_half8*>(wt + (size_t)n * (NK * 128) + s * 128 + 64 + c * 16);
        }
        if (t < BN) s_bias[t] = t < a.Npad ? a.bias[t] : 0.f;
    }

    // ---- pixel staging: thread = (row t >> 2, 8-channel unit t & 3) of every K step ---------------------------------------------
    const int xrow = t >> 2, xc = t & 3;
    const int xoff = pf_lds_chunk_off(xrow, xc);
    pf_f32x4 xr[NK][2];
    auto load_x = [&](int tile) {
        const float* p = in + ((size_t)tile * BM + xrow) * a.inLd + xc * 8;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            xr[s][0] = *reinterpret_cast<const pf_f32x4*>(p + s * 32);
            xr[s][1] = *reinterpret_cast<const pf_f32x4*>(p + s * 32 + 4);
        }
    };
    auto park_x = [&](int stage) {
        unsigned char* base = smem + stage * STAGE + xoff;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            pf_half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = xr[s][e >> 2][e & 3];
                const pf_half hv = (pf_half)v;
                hi[e] = hv;
                if constexpr (!ONEPROD) lo[e] = pf_split_lo(v, hv);
                amax = pf_amax(amax, v);
            }
            *reinterpret_cast<pf_half8*>(base + s * PLANE) = hi;
            if constexpr (!ONEPROD) *reinterpret_cast<pf_half8*>(base + (NK + s) * PLANE) = lo;
        }
    };
    pf_f32x4 acc[NT][MT];
    auto mma_tile = [&](int stage) {
        const unsigned char* base = smem + stage * STAGE;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            pf_half8 xhf[MT], xlf[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int off = pf_lds_chunk_off(wm * WM + i * 16 + frow, fchunk);
                xhf[i] = *reinterpret_cast<const pf_half8*>(base + s * PLANE + off);
                if constexpr (!ONEPROD) xlf[i] = *reinterpret_cast<const pf_half8*>(base + (NK + s) * PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {          // small terms first, the dominant hi * hi term last (as conv_gemm_split_kernel)
                if constexpr (!ONEPROD) {
                    const pf_half8 wlj = *reinterpret_cast<const pf_half8*>(s_wl + s * (MAXN * 64) + wloff[j]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(wlj, xhf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf[j][s], xlf[i], acc[j][i]);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf[j][s], xhf[i], acc[j][i]);
            }
        }
    };

    // ---- the arg-max, per LANE across the tiles of a work item, across lanes once per item -----------------------------------------
    // A work item is a run of `tps` consecutive tiles of ONE face (segs items per face; host: segs = 1 once there are as many faces
    // as CUs).  Each lane keeps the best score and pixel index of ITS pixels (column lane & 15 of the wave's two pixel tiles, in
    // ascending pixel order: strict > keeps the first maximum) in registers; the 16-lane exchange of conv_gemm_argmax_epilogue --
    // 650 of the 1150 VALU instructions per tile and wave when done per tile, and the kernel is VALU-bound (52 % busy at 4 cycles per
    // instruction against 17 % for the matrix pipe: profiles/r05_run35_pmc_pw_head_sq.json) -- runs once per item.  The partial of
    // (item segment, wave row) is written to ALL the slots of the [B][P][nslots] buffers that the per-tile partials of its pixels
    // used to fill: hm_decode_kernel's merge (greater score, then smaller index) does not care about duplicates.
    const int tpf = OHW / BM;                       // tiles per face
    const int segs = a.head_segs;
    const int tps = tpf / segs;
    const int nitems = a.B * segs;
    float best_v[NT][4] = {};
    int best_i[NT][4] = {};
    const int pcol = lane & 15, crow = (lane >> 4) * 4;
    auto update_best = [&](int local0, bool first) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(s_bias + wn * WN + j * 16 + crow);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float v = fmaf(acc[j][i][r], a.acc_scale, bv[r]);
                    if ((first && i == 0) || v > best_v[j][r]) { best_v[j][r] = v; best_i[j][r] = local0 + i * 16; }
                }
            }
        }
    };
    auto write_best = [&](int face, int seg) {
        const int nslots = tpf * WARPS_M;
        const int slot0 = seg * tps * WARPS_M;      // this item's tiles owned slots [slot0, slot0 + tps WARPS_M): tile-major, wave row minor
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = wn * WN + j * 16 + crow;
#define PF_AMAX_STEP(STEP)                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                              \
        const float ov = pf_row_xchg_f32<STEP>(best_v[j][r]);                                                                 \
        const int oi = pf_row_xchg_i32<STEP>(best_i[j][r]);                                                                   \
        if (ov > best_v[j][r] || (ov == best_v[j][r] && oi < best_i[j][r])) { best_v[j][r] = ov; best_i[j][r] = oi; }         \
    }
            PF_AMAX_STEP(0) PF_AMAX_STEP(1) PF_AMAX_STEP(2) PF_AMAX_STEP(3)
#undef PF_AMAX_STEP
            // every lane of the 16-lane row now holds the row's result: lane pcol writes the slots pcol, pcol + 16, ... of wave row wm
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < a.amaxN) {
                    const size_t o = ((size_t)face * a.amaxN + n + r) * nslots + slot0 + wm;
                    for (int k = pcol; k < tps; k += 16) {
                        a.amax_val[o + (size_t)k * WARPS_M] = best_v[j][r];
                        a.amax_idx[o + (size_t)k * WARPS_M] = best_i[j][r];
                    }
                }
        }
    };

    // ---- the walk: ONE request site and ONE park site per iteration, both unconditional (past the end the last tile is requested
    // again and a stage nobody reads is written): loads the compiler sees in two branches, or under a condition, are merged with
    // register copies right behind the request -- which wait for the data and undo the look-ahead ------------------------------------
    const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // host: grid <= nitems
    const int total = my_items * tps;               // tiles this workgroup visits, q = 0 .. total - 1
    if (total <= 0) return;                         // (a grid larger than the item list: nothing to do, and no barrier has been reached yet)
    auto tile_of = [&](int q, int& face, int& seg, int& kk) {
        const int item = blockIdx.x + (q / tps) * gridDim.x;
        kk = q - (q / tps) * tps;
        face = item / segs;
        seg = item - face * segs;
        return face * tpf + seg * tps + kk;
    };
    int f0, s0, k0;
    load_x(tile_of(0, f0, s0, k0));
    park_x(0);
    load_x(tile_of(min(1, total - 1), f0, s0, k0));
    __syncthreads();
    for (int q = 0, cur = 0; q < total; ++q, cur ^= 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        // the other stage was last read before the previous barrier: the next tile, requested a whole iteration ago, is split and
        // parked there and the tile after it requested; then this tile's MFMAs (the two waves of a SIMD drift apart by themselves)
        park_x(cur ^ 1);
        load_x(tile_of(min(q + 2, total - 1), f0, s0, k0));
        pf_sched_fence();                           // requested NOW: left alone the scheduler sinks the requests to the end of the MFMA chain
        mma_tile(cur);
        int face, seg, kk;
        tile_of(q, face, seg, kk);
        update_best((seg * tps + kk) * BM + wm * WM + pcol, kk == 0);
        if (kk == tps - 1) write_best(face, seg);
        __syncthreads();
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}
