// A whole inverted-residual block of the Student's encoder at 16 x 16 (timm MobileNetV3 stages 3-5 behind
// TRAIN/face_landmark/lib/core/base_trainer/model.py:252-264: expand 1x1 -> depthwise k x k -> [squeeze-excite] -> project 1x1
// [+ x]) with ONE persistent workgroup per face, the face's input stationary in registers and the expanded tensor in LDS (round 5).
//
// What it replaces (profiles/r04_run45_kernel_table.json, ms per 256 faces): per block one expand + depthwise launch
// (conv_gemm_split_kernel<..EPI_K>: 3 840 workgroups of one face x 64 expanded channels, every one a chain of ~8 memory round trips
// whose phases ADD UP -- skeleton 0.108 / operand fetches 0.090 / stores at the HBM roof 0.084 of 0.346 ms, r04_run20_expdw_ablations)
// that re-reads the block input once per 64 channels (15 x at 960) and writes the expanded f32 map, then the projection that reads it back.
//
// Here NW waves (16, or 8 where the registers ask for it) own a face: wave w owns MT = 16 / NW image rows (16-pixel MFMA tiles) and
// keeps their pixel fragments of ALL input channels in registers as split f16 hi / lo (KS k-steps x MT x 8 VGPRs), loaded once per
// face.  The expanded channels are walked 32 at a time ("tiles"):
//   expand(t)     E = act(W1[t] . x + b1) on the matrix cores (3 x v_mfma_f32_16x16x32_f16 per product, f32 accumulate), the 32 x (KS x 32)
//                 pre-split weight rows from an LDS stage (LDS-DMA), E -> LDS as f32 [256 px][ES];
//   depthwise(t)  k x k taps (dilation DIL, zero padding) on E out of LDS, thread = (channel pair, image row, 64 / NW pixels of the row),
//                 f32 fma in the order of the unfused kernel -- as SCALAR v_fma / v_fmac: the library is built with
//                 -fno-slp-vectorize (build.py; see mbx_launch.cpp) because hipcc's SLP vectoriser packs the channel pair into
//                 v_pk_fma_f32 -- which costs as much as the two scalar fma it replaces -- and pays for it with a register shuffle
//                 (v_mov) per operand: the stream doubles --, + bias, activation, then by MODE
//                   0 / 2  (x SE gate) -> split hi / lo -> the pixel-operand planes D[t & 1] in LDS,
//                   1 / 3  per-thread sums -> LDS -> per-face channel means (the SE squeeze); MODE 3 also stores the activated map (f32) for
//                          the layer-wise gated projection;
//   project(t)    MODE 0 / 2: out += W2[:, t] . D[t & 1] on the matrix cores, the accumulators (MT x NTO x 4 VGPRs) in registers for the face.
// MODE 0 / 2 run two barrier-separated phases per tile -- a: project(t - 1) | depthwise(t), b: expand(t + 1) -- MODE 1 / 3, which need no D
// planes and so have the LDS for two E tiles, ONE: expand(t + 1) | depthwise(t).  In a phase that pairs a matrix job with a VALU job the
// lower half of the waves runs the matrix job first and the upper half the VALU job first: waves w, w + 4, ... share a SIMD, so its
// matrix pipe and its VALU are busy at the same time without instruction-level interleaving.
// Everything that comes from memory inside the loop arrives by LDS-DMA issued at the START of a phase for a LATER phase (one ahead with
// two phases per tile, two tiles ahead with one), so every barrier is a plain "vmcnt(0) + s_barrier": no hand-counted partial waits.
//
// An SE block is: squeeze pass (MODE 1 or 3) -> the two small FC launches -> either MODE 2, which RECOMPUTES expand + depthwise (the input
// is in registers, the weights in L2) and projects the gated result -- the expanded tensor never reaches HBM --, or (after MODE 3) the
// layer-wise gated projection on the stored map.  The depthwise is VALU-bound (scalar f32 VALU issues one wave instruction per 4 cycles
// here; a dilated 5 x 5 is ~260 of them per thread and tile), so a second pass over it costs more than the map's round trip through HBM:
// storing measured faster for every SE block of the Student, 3 x 3 included (profiles/r05_run11_mbx_ab_v4b.txt; ir.py::mbx defaults to it).
//
// What the first cuts taught (profiles/r05_run3 ... r05_run9):
//   * 8 waves only (two per SIMD): a single wave issues one VALU instruction per ~4 cycles, so with its SIMD partner in a matrix job the
//     VALU ran half empty -- depthwise 7 k cycles per tile and wave against 3.3 k of instruction issue;
//   * v_pk_fma_f32 taps as the SLP vectoriser builds them: 200 packed fma + 210 v_mov per thread and tile instead of 400 scalar fma; written
//     by hand on the natural pairs (no shuffles) they measure the same as the scalar form: a packed f32 fma costs two scalar ones;
//   * breaking the vectoriser's pairs with asm statements in the fma stream gave RUN-TO-RUN DIFFERENT results on MI355X (the hazard
//     recogniser does not see through asm statements); the translation-unit flag is the fix;
//   * residual vectors added load by load between the stores of the epilogue: twenty dependent round trips (52 us per launch);
//   * carrying the tap results over a barrier (to finish them beside expand) spilled the accumulators inside the projection's MFMA chain.
// Arithmetic (product order per accumulator, fma order of the depthwise taps, bias / activation / gate / split) is the unfused path's.
//
// Host guarantees (engine.cpp PF_OP_MBX): 16 x 16 maps, inC % 4 == 0, inC <= 32 KS, COUT == 16 NTO, pad == DIL (K - 1) / 2,
// activation relu or hard-swish; weights packed by ir.py::mbx (W1 [32 T][KS][hi 32 | lo 32], per-tile constants [T][K K + 2][32] =
// taps | b1 | b_dw, W2 [COUT][T][hi 32 | lo 32]).
#pragma once
#include "pf_common.h"
#include "k_conv_gemm.h"
#include "k_det.h"        // PF_EMU_POISON
#include "k_mbx_args.h"

template <int NW, int KS, int NTO, int K, int DIL, int MODE>
__global__ __launch_bounds__(NW * 64, NW / 4) void mbx_kernel(MbxArgs a) {
    constexpr bool PROJECT = MODE == 0 || MODE == 2;        // the projection runs here (accumulators in registers)
    constexpr bool SQUEEZE = MODE == 1 || MODE == 3;        // per-face channel means of the depthwise output
    constexpr bool STORE_D = MODE == 3;                     // ... and the activated depthwise map itself to HBM
    constexpr int NTHR = NW * 64;
    constexpr int MT = 16 / NW;                             // image rows (16-pixel MFMA tiles) per wave
    constexpr int XP = 64 / NW;                             // pixels of a row per depthwise thread: 8 or 4
    constexpr int NP = 16 / XP;                             // depthwise threads along a row
    constexpr int NPART = 16 * NP;                          // partial sums per channel (MODE 1 / 3)
    constexpr int PAD = DIL * (K - 1) / 2;
    // floats per E pixel row: the row parts a 32-lane group of one ds_read_b64 covers (XP columns apart) must land on disjoint banks
    constexpr int ES = NW == 16 ? 40 : 36;
    constexpr int E_BYTES = 256 * ES * 4;
    constexpr int NE = PROJECT ? 1 : 2;                     // E tiles
    constexpr int D_BYTES = 32768;                          // hi plane 256 x 64 B + lo plane
    constexpr int W1_BYTES = KS * 4096;
    constexpr int NW1 = PROJECT ? 1 : 2;                    // W1 stages
    constexpr int COUT = NTO * 16;
    constexpr int W2_BYTES = PROJECT ? COUT * 128 : 0;
    constexpr int CT_FLOATS = (K * K + 2) * 32;
    constexpr int CT_SLOTS = ((CT_FLOATS / 4 + 63) / 64) * 64;   // 16-byte slots, whole waves
    constexpr int CT_BYTES = CT_SLOTS * 16 + 1024;          // + the gate's wave (32 floats used)
    constexpr int NCT = PROJECT ? 2 : 3;                    // constant stages
    constexpr int PS_BYTES = SQUEEZE ? 2 * NPART * 32 * 4 : 0;
    constexpr int NACC = PROJECT ? NTO : 1;
    constexpr int Z_BYTES = 1024;                           // zeros: what a depthwise thread reads for the columns outside the image
    constexpr int LDS_BYTES = NE * E_BYTES + (PROJECT ? 2 * D_BYTES : 0) + NW1 * W1_BYTES + W2_BYTES + NCT * CT_BYTES + PS_BYTES + Z_BYTES;
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    static_assert(PAD >= 1 && PAD <= XP && (K == 3 || K == 5), "depthwise window");
    static_assert(CT_SLOTS + 64 <= NTHR, "constants + gate: one request per thread");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    float* const es = reinterpret_cast<float*>(smem);
    unsigned char* const dbase = smem + NE * E_BYTES;
    unsigned char* const w1s = dbase + (PROJECT ? 2 * D_BYTES : 0);
    unsigned char* const w2s = w1s + NW1 * W1_BYTES;
    unsigned char* const cts = w2s + W2_BYTES;
    float* const psum = reinterpret_cast<float*>(cts + NCT * CT_BYTES);     // [2][NPART][32]
    unsigned char* const zeros = cts + NCT * CT_BYTES + PS_BYTES;
    static_assert((PAD - 1) * ES * 4 + 8 <= Z_BYTES, "zero block covers a side's PAD columns");
    PF_EMU_POISON(smem);
    if (threadIdx.x < Z_BYTES / 4) reinterpret_cast<float*>(zeros)[threadIdx.x] = 0.f;      // (the first barrier of the first face orders it)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned amax = 0;                                      // range guard (pf_common.h): everything this launch splits
    const unsigned amax_seen = pf_amax_seen(a.range_slot);

    // ---- LDS-DMA streams (16-byte slots, whole waves; the slot of lane l of a wave-instruction is base + 16 l) ------------------
    // Every thread-local index below is derived from an "opaque" copy of the thread index (pf_opaque: an empty asm the compiler must
    // assume changes it), once per call: left alone, the loop-invariant per-lane addresses of ALL phases (~40 of them, the DMA sources
    // as 64-bit pointers) are hoisted out of the tile loop, and next to the resident fragment / accumulator registers they spill
    // (first build: 616 bytes of scratch per lane).  Recomputing them costs a few dozen VALU instructions per phase.
    auto glds = [&](const void* sb, unsigned voff, void* dst) { pf_glds16_raw_soff<0>(sb, voff, dst); };
    // Tile indices below are RELATIVE to the work unit's first tile `tb` (round 6: in the squeeze modes a face is split into a.nsplit
    // tile ranges): LDS stages, parities and the loop structure see every unit as a run from tile 0, only global addresses add tb.
    int tb = 0;
    auto dma_w1 = [&](int tile) {                           // slot -> [k-step][plane][row][position], chunk rotation on the SOURCE
        const int tt = pf_opaque(t);
        const unsigned char* sb = a.w1 + (size_t)(tb + tile) * (32 * KS * 128);
        unsigned char* dst = w1s + (tile % NW1) * W1_BYTES;
#pragma unroll
        for (int r = 0; r < (KS * 256 + NTHR - 1) / NTHR; ++r) {
            const int sl = r * NTHR + tt;
            if (sl < KS * 256) {
                const int s = sl >> 8, plane = (sl >> 7) & 1, row = (sl >> 2) & 31;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                glds(sb, (unsigned)((row * KS + s) * 128 + plane * 64 + chunk * 16), dst + (size_t)sl * 16);
            }
        }
    };
    auto dma_w2 = [&](int tile) {                           // slot -> [plane][row][position]
        if constexpr (PROJECT) {
            const int tt = pf_opaque(t);
            const unsigned char* sb = a.w2 + (size_t)(tb + tile) * 128;
#pragma unroll
            for (int r = 0; r < (COUT * 8 + NTHR - 1) / NTHR; ++r) {
                const int sl = r * NTHR + tt;
                if (sl < COUT * 8) {
                    const int plane = sl >= COUT * 4 ? 1 : 0;
                    const int row = (sl - plane * COUT * 4) >> 2;
                    const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                    glds(sb, (unsigned)(row * a.T * 128 + plane * 64 + chunk * 16), w2s + (size_t)sl * 16);
                }
            }
        }
    };
    auto dma_ct = [&](int tile, int face) {                 // taps | b1 | b_dw of the tile (+ the face's gate values of its 32 channels)
        const int tt = pf_opaque(t);
        unsigned char* dst = cts + (tile % NCT) * CT_BYTES;
        if (tt < CT_SLOTS) {
            const int sl = tt < CT_FLOATS / 4 ? tt : 0;     // padding slots of the last wave re-read slot 0
            glds(a.ctile + (size_t)(tb + tile) * CT_FLOATS, (unsigned)(sl * 16), dst + (size_t)tt * 16);
        } else if (MODE == 2 && tt < CT_SLOTS + 64) {       // the next wave: the gate values, slots GATE_OFF / 16 ... (8 distinct ones)
            glds(a.gate + (size_t)face * a.CEXP + (tb + tile) * 32, (unsigned)((tt & 7) * 16), dst + (size_t)tt * 16);
        }
    };

    const bool prof = (pf_dbg(a) & 64) != 0;
    unsigned long long c_pro = 0, c_mma = 0, c_wa = 0, c_dw = 0, c_exp = 0, c_wb = 0, c_epi = 0;
    const int NS = SQUEEZE ? a.nsplit : 1;
    for (int unit = blockIdx.x; unit < a.B * NS; unit += gridDim.x) {
        const int face = unit / NS, upart = unit - face * NS;
        tb = a.T * upart / NS;
        const int T = a.T * (upart + 1) / NS - tb;          // this unit's tiles: [tb, tb + T)
        const unsigned long long q0 = prof ? pf_clock() : 0;
        dma_w1(0);
        dma_ct(0, face);
        if constexpr (!PROJECT) { if (T > 1) { dma_w1(1); dma_ct(1, face); } }
        pf_f32x4 oacc[MT][NACC];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NACC; ++j) oacc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- the face's input -> split pixel fragments in registers (lane = pixel pcol of the row, k-group kg): one batch of KS x 2
        // unconditional 16-byte loads per image row (channels beyond inC read the pixel's first channels and are zeroed) -------------------
        pf_half8 xh[MT][KS], xl[MT][KS];
        {
            const int pcol = lane & 15, kg = lane >> 4;
            const float* xin = a.in + ((size_t)face * 256 + wave * 16 * MT + pcol) * a.inLd;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                pf_f32x4 xv[KS][2];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int c0 = s * 32 + kg * 8;
                    const float* p = xin + (size_t)i * 16 * a.inLd;
                    xv[s][0] = *reinterpret_cast<const pf_f32x4*>(p + (c0 < a.inC ? c0 : 0));
                    xv[s][1] = *reinterpret_cast<const pf_f32x4*>(p + (c0 + 4 < a.inC ? c0 + 4 : 0));
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int c0 = s * 32 + kg * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = xv[s][e >> 2][e & 3];
                        if (!(c0 + (e & 4) < a.inC)) v = 0.f;
                        const pf_half hv = (pf_half)v;
                        xh[i][s][e] = hv;
                        xl[i][s][e] = pf_split_lo(v, hv);
                        amax = pf_amax(amax, v);
                    }
                }
            }
        }

        // ---- expand(tile): E[tile % NE] = act(W1[tile] . x + b1) ----------------------------------------------------------------------
        auto expand = [&](int tile) {
            const int tt = pf_opaque(t);
            const int pcol = tt & 15, kg = (tt >> 4) & 3, wv = tt >> 6;
            const float* ct = reinterpret_cast<const float*>(cts + (tile % NCT) * CT_BYTES);
            const unsigned char* wsrc = w1s + (tile % NW1) * W1_BYTES;
            float* edst = es + (tile % NE) * (E_BYTES / 4);
            pf_f32x4 acc[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            pf_half8 eq[2][2];                              // weight fragments of step (s, j): requested one step ahead, as in project()
            {
                const int off = pf_lds_chunk_off(pcol, kg);
                eq[0][0] = *reinterpret_cast<const pf_half8*>(wsrc + off);
                eq[0][1] = *reinterpret_cast<const pf_half8*>(wsrc + 2048 + off);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    constexpr int dummy = 0; (void)dummy;
                    const int step = s * 2 + j;
                    if (step + 1 < 2 * KS) {
                        const int s1 = (step + 1) >> 1, j1 = (step + 1) & 1;
                        const int off = s1 * 4096 + pf_lds_chunk_off(j1 * 16 + pcol, kg);
                        eq[(step + 1) & 1][0] = *reinterpret_cast<const pf_half8*>(wsrc + off);
                        eq[(step + 1) & 1][1] = *reinterpret_cast<const pf_half8*>(wsrc + 2048 + off);
                    }
                    const pf_half8 wh = eq[step & 1][0], wl = eq[step & 1][1];
                    if (pf_dbg(a) & 4) continue;
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wl, xh[i][s], acc[i][j]);      // small terms first
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wh, xl[i][s], acc[i][j]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wh, xh[i][s], acc[i][j]);
                    asm volatile("" ::: "memory");          // (register footprint: two weight fragments at a time)
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(ct + K * K * 32 + j * 16 + kg * 4);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    pf_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[i][j][r], a.scale1, bv[r]);
                    pf_act_rh<4>(v, a.act);
                    *reinterpret_cast<pf_f32x4*>(edst + (wv * 16 * MT + i * 16 + pcol) * ES + j * 16 + kg * 4) = v;
                }
            }
        };
        // ---- project(tile): out += W2[:, tile] . D[tile & 1] ---------------------------------------------------------------------------
        auto project = [&](int tile) {
            if constexpr (PROJECT) {
                const int tt = pf_opaque(t);
                const int pcol = tt & 15, kg = (tt >> 4) & 3, wv = tt >> 6;
                const unsigned char* dsrc = dbase + (tile & 1) * D_BYTES;
                pf_half8 dh[MT], dl[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int off = pf_lds_chunk_off(wv * 16 * MT + i * 16 + pcol, kg);
                    dh[i] = *reinterpret_cast<const pf_half8*>(dsrc + off);
                    dl[i] = *reinterpret_cast<const pf_half8*>(dsrc + 16384 + off);
                }
                // the weight fragments of channel tile j + 1 are requested in front of tile j's MFMAs and no further ahead (the compiler
                // fence): their LDS latency hides behind six MFMAs instead of standing in front of every tile's chain
                pf_half8 wq[2][2];
                {
                    const int off = pf_lds_chunk_off(pcol, kg);
                    wq[0][0] = *reinterpret_cast<const pf_half8*>(w2s + off);
                    wq[0][1] = *reinterpret_cast<const pf_half8*>(w2s + COUT * 64 + off);
                }
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    if (j + 1 < NTO) {
                        const int off = pf_lds_chunk_off((j + 1) * 16 + pcol, kg);
                        wq[(j + 1) & 1][0] = *reinterpret_cast<const pf_half8*>(w2s + off);
                        wq[(j + 1) & 1][1] = *reinterpret_cast<const pf_half8*>(w2s + COUT * 64 + off);
                    }
                    const pf_half8 wh = wq[j & 1][0], wl = wq[j & 1][1];
                    if (pf_dbg(a) & 4) continue;
#pragma unroll
                    for (int i = 0; i < MT; ++i) oacc[i][j] = pf_mfma_16x16x32_f16(wl, dh[i], oacc[i][j]);     // small terms first
#pragma unroll
                    for (int i = 0; i < MT; ++i) oacc[i][j] = pf_mfma_16x16x32_f16(wh, dl[i], oacc[i][j]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) oacc[i][j] = pf_mfma_16x16x32_f16(wh, dh[i], oacc[i][j]);
                    asm volatile("" ::: "memory");
                }
            }
        };
        // ---- depthwise(tile): E -> XP pixels x 2 channels per thread: taps, bias, activation -> D planes / sums / the stored map -------------
        auto depthwise = [&](int tile) {
            float of[2 * XP];                               // of[2 x + c]: pixel XP part + x of row yrow, channel c2 + c
            const int tt = pf_opaque(t);
            const int c2 = (tt & 15) * 2, part = (tt >> 4) & (NP - 1), yrow = tt / (16 * NP);
            const float* ct = reinterpret_cast<const float*>(cts + (tile % NCT) * CT_BYTES);
            const float* esrc = es + (tile % NE) * (E_BYTES / 4);
            const pf_f32x2 bd = *reinterpret_cast<const pf_f32x2*>(ct + (K * K + 1) * 32 + c2);
#pragma unroll
            for (int x = 0; x < XP; ++x) { of[2 * x] = bd[0]; of[2 * x + 1] = bd[1]; }
            // The PAD columns either side of this part of the row belong to its neighbours, or lie outside the image (PAD <= XP: a side is
            // wholly inside or wholly outside); an outside side is READ FROM A BLOCK OF ZEROS -- one address select per side and filter
            // row instead of a select per value (the depthwise is VALU-bound: scalar f32 VALU runs at 4 cycles per wave instruction here,
            // and the selects were a fifth of the instruction stream).  With 16 waves a wave is one image row, so a filter row that falls
            // above / below the image is skipped by a scalar branch -- with its reads and taps, 15 % of a dilated 5 x 5 -- where the
            // 8-wave layout (two rows per wave) has to zero the row's taps instead.
            const bool lok = part > 0, rok = part < NP - 1;
            const unsigned char* ebase = reinterpret_cast<const unsigned char*>(esrc + c2);
            const int mid_off = XP * part * ES * 4;
            constexpr bool ROW_UNIFORM = NW == 16;
            const int yrow_u = ROW_UNIFORM ? pf_uniform_i32(yrow) : yrow;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yy = yrow_u + ky * DIL - PAD;
                const bool yok = (unsigned)yy < 16u;
                if constexpr (ROW_UNIFORM) { if (!yok) continue; }
                const int yc = ROW_UNIFORM ? yy : (yy < 0 ? 0 : (yy > 15 ? 15 : yy));
                const unsigned char* erow = ebase + yc * 16 * ES * 4;
                const unsigned char* lp = lok ? erow + mid_off - PAD * ES * 4 : zeros;
                const unsigned char* rp = rok ? erow + mid_off + XP * ES * 4 : zeros;
                pf_f32x2 in[XP + 2 * PAD];                  // in[PAD + j] = column XP part + j, j = -PAD .. XP - 1 + PAD
#pragma unroll
                for (int j = 0; j < XP; ++j) in[PAD + j] = *reinterpret_cast<const pf_f32x2*>(erow + mid_off + j * ES * 4);
#pragma unroll
                for (int s = 0; s < PAD; ++s) {
                    in[s] = *reinterpret_cast<const pf_f32x2*>(lp + s * ES * 4);
                    in[PAD + XP + s] = *reinterpret_cast<const pf_f32x2*>(rp + s * ES * 4);
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    pf_f32x2 w = *reinterpret_cast<const pf_f32x2*>(ct + (ky * K + kx) * 32 + c2);
                    if constexpr (!ROW_UNIFORM) { if (!yok) w = pf_f32x2{0.f, 0.f}; }      // a filter row above / below the image: its taps contribute nothing
                    if (pf_dbg(a) & 2) continue;
#pragma unroll
                    for (int x = 0; x < XP; ++x) {
                        of[2 * x] = fmaf(w[0], in[x + kx * DIL][0], of[2 * x]);
                        of[2 * x + 1] = fmaf(w[1], in[x + kx * DIL][1], of[2 * x + 1]);
                    }
                }
                asm volatile("" ::: "memory");              // one filter row's LDS reads in flight at a time (register footprint)
            }
            pf_act_rh<2 * XP>(of, a.act);
            if constexpr (SQUEEZE) {
                pf_f32x2 rs = pf_f32x2{0.f, 0.f};
#pragma unroll
                for (int x = 0; x < XP; ++x) { rs[0] += of[2 * x]; rs[1] += of[2 * x + 1]; }
                *reinterpret_cast<pf_f32x2*>(psum + (tile & 1) * (NPART * 32) + (yrow * NP + part) * 32 + c2) = rs;
                if constexpr (STORE_D) {                    // a pixel's 32 channels of the tile are one 128-byte line: 16 lanes x 8 bytes
                    float* drow = a.out + ((size_t)face * 256 + yrow * 16 + XP * part) * a.outLd + (tb + tile) * 32 + c2;
                    if ((tb + tile) * 32 + c2 < a.CEXP && !(pf_dbg(a) & 16))
#pragma unroll
                        for (int x = 0; x < XP; ++x) *reinterpret_cast<pf_f32x2*>(drow + (size_t)x * a.outLd) = pf_f32x2{of[2 * x], of[2 * x + 1]};
                }
            } else {
                float g0 = 1.f, g1 = 1.f;
                if constexpr (MODE == 2) {
                    const pf_f32x2 g = *reinterpret_cast<const pf_f32x2*>(reinterpret_cast<const float*>(cts + (tile % NCT) * CT_BYTES + CT_SLOTS * 16) + c2);
                    g0 = g[0]; g1 = g[1];
                }
                // pixel-operand row of pixel P0 + x (P0 = 16 yrow + XP part): the chunk rotation of pf_lds_chunk_off depends on x only
                // through ((XP part + x) >> 2) & 1 (16 yrow >> 2 is a multiple of 4), so two base addresses + compile-time offsets do
                unsigned char* dp = dbase + (tile & 1) * D_BYTES + (yrow * 16 + XP * part) * 64 + (c2 & 7) * 2;
                const int rot0 = (c2 >> 3) + 2 * ((XP * part) >> 2);
                unsigned char* const dp0 = dp + (rot0 & 3) * 16;
                unsigned char* const dp1 = dp + ((rot0 + 2) & 3) * 16;
#pragma unroll
                for (int x = 0; x < XP; ++x) {
                    const float v0 = MODE == 2 ? of[2 * x] * g0 : of[2 * x], v1 = MODE == 2 ? of[2 * x + 1] * g1 : of[2 * x + 1];
                    pf_half2 hi, lo;
                    hi[0] = (pf_half)v0; hi[1] = (pf_half)v1;
                    lo[0] = pf_split_lo(v0, hi[0]); lo[1] = pf_split_lo(v1, hi[1]);
                    amax = pf_amax(pf_amax(amax, v0), v1);
                    unsigned char* q = (x < 4 ? dp0 : dp1) + x * 64;
                    *reinterpret_cast<pf_half2*>(q) = hi;
                    *reinterpret_cast<pf_half2*>(q + 16384) = lo;
                }
            }
        };
        auto squeeze = [&](int tile) {                      // MODE 1 / 3: the NPART partial sums per channel of a finished tile, in a fixed order
            if constexpr (SQUEEZE) {
                if (t < 32) {
                    const float* ps = psum + (tile & 1) * (NPART * 32) + t;
                    float tot = 0.f;
#pragma unroll
                    for (int q = 0; q < NPART; ++q) tot += ps[q * 32];
                    const int c = (tb + tile) * 32 + t;
                    if (c < a.CEXP) a.gap_out[(size_t)face * a.CEXP + c] = tot / 256.f;
                }
            }
        };

        pf_wait_vm_barrier<0>();                            // W1(0), constants(0) have landed (and the input loads with them)
        expand(0);
        pf_wait_vm_barrier<0>();
        if (prof) c_pro += pf_clock() - q0;
        if constexpr (PROJECT) {
            for (int tile = 0; tile < T; ++tile) {
                // ======== phase a: project(tile - 1) | depthwise of tile -> D[tile & 1]; W1 / constants of tile + 1 on their way ==============
                const unsigned long long q1 = prof ? pf_clock() : 0;
                if (tile + 1 < T && !((pf_dbg(a) & 1) && tile > 0)) { dma_w1(tile + 1); dma_ct(tile + 1, face); }
                unsigned long long q2 = q1, q3 = q1;
                if (wave < NW / 2) {
                    if (tile >= 1) project(tile - 1);
                    if (prof) q2 = pf_clock();
                    depthwise(tile);
                    if (prof) { q3 = pf_clock(); c_mma += q2 - q1; c_dw += q3 - q2; }
                } else {
                    depthwise(tile);
                    if (prof) q2 = pf_clock();
                    if (tile >= 1) project(tile - 1);
                    if (prof) { q3 = pf_clock(); c_dw += q2 - q1; c_mma += q3 - q2; }
                }
                pf_wait_vm_barrier<0>();
                // ======== phase b: expand(tile + 1) -> E; W2 of tile on its way =============================================================
                const unsigned long long q4 = prof ? pf_clock() : 0;
                if (!((pf_dbg(a) & 1) && tile > 0)) dma_w2(tile);
                if (tile + 1 < T) expand(tile + 1);
                const unsigned long long q6 = prof ? pf_clock() : 0;
                pf_wait_vm_barrier<0>();
                if (prof) { c_wa += q4 - q3; c_exp += q6 - q4; c_wb += pf_clock() - q6; }
            }
        } else {
            for (int tile = 0; tile < T; ++tile) {
                // ======== one phase: expand(tile + 1) -> E[(tile + 1) & 1] | depthwise of tile; W1 / constants of tile + 2 on their way ==========
                const unsigned long long q1 = prof ? pf_clock() : 0;
                if (tile + 2 < T && !((pf_dbg(a) & 1) && tile > 0)) { dma_w1(tile + 2); dma_ct(tile + 2, face); }
                unsigned long long q2 = q1, q3 = q1;
                if (wave < NW / 2) {
                    if (tile + 1 < T) expand(tile + 1);
                    if (prof) q2 = pf_clock();
                    depthwise(tile);
                    if (prof) { q3 = pf_clock(); c_exp += q2 - q1; c_dw += q3 - q2; }
                } else {
                    depthwise(tile);
                    if (prof) q2 = pf_clock();
                    if (tile + 1 < T) expand(tile + 1);
                    if (prof) { q3 = pf_clock(); c_dw += q2 - q1; c_exp += q3 - q2; }
                }
                if (tile >= 1) squeeze(tile - 1);
                pf_wait_vm_barrier<0>();
                if (prof) c_wa += pf_clock() - q3;
            }
            squeeze(T - 1);
        }
        const unsigned long long q7 = prof ? pf_clock() : 0;
        if constexpr (PROJECT) {
            // ---- block output = acc * scale2 + bias (+ residual), no activation (timm InvertedResidual: the projection is linear).  The
            // residual vectors are requested BEFORE the last tile's projection -- all of them at once (the input fragments are dead by
            // now, so the registers are there): one round trip hidden behind the MFMAs.  The first cut added them load by load between
            // the stores (the compiler cannot move a load above a store that may alias it): twenty dependent round trips, 52 us per launch.
            const int tt = pf_opaque(t);
            const int pcol = tt & 15, kg = (tt >> 4) & 3, wv = tt >> 6;
            pf_f32x4 rv[MT][NTO];
            if (a.res) {
                const float* __restrict__ rrow = a.res + ((size_t)face * 256 + wv * 16 * MT + pcol) * a.resLd + kg * 4;
#pragma unroll
                for (int j = 0; j < NTO; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) rv[i][j] = *reinterpret_cast<const pf_f32x4*>(rrow + (size_t)i * 16 * a.resLd + j * 16);
            }
            project(T - 1);
            float* __restrict__ orow = a.out + ((size_t)face * 256 + wv * 16 * MT + pcol) * a.outLd + kg * 4;
            if (!(pf_dbg(a) & 16))
#pragma unroll
            for (int j = 0; j < NTO; ++j) {
                const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(a.b2 + j * 16 + kg * 4);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    pf_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(oacc[i][j][r], a.scale2, bv[r]);
                    if (a.res) v += rv[i][j];
                    *reinterpret_cast<pf_f32x4*>(orow + (size_t)i * 16 * a.outLd + j * 16) = v;
                }
            }
        }
        if (prof) c_epi += pf_clock() - q7;
        // the next face's first requests target stages last read before the loop's final barrier; what the trailing project / squeeze read
        // (W2, D, the partial sums) is next written at least one barrier into the next face
    }
    if (prof && lane == 0) {
        atomicAdd(a.prof + 0, c_pro); atomicAdd(a.prof + 1, c_mma); atomicAdd(a.prof + 2, c_wa); atomicAdd(a.prof + 3, c_dw);
        atomicAdd(a.prof + 4, c_exp); atomicAdd(a.prof + 5, c_wb); atomicAdd(a.prof + 6, c_epi); atomicAdd(a.prof + 7, 1ull);
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}

