// A whole inverted-residual block of the Student's encoder at 16 x 16 (timm MobileNetV3 stages 3-5 behind
// TRAIN/face_landmark/lib/core/base_trainer/model.py:252-264: expand 1x1 -> depthwise k x k -> [squeeze-excite] -> project 1x1
// [+ x]) with ONE persistent workgroup per face and the expanded tensor never in HBM (round 5).
//
// What it replaces (profiles/r04_run45_kernel_table.json, ms per 256 faces): per block one expand + depthwise launch
// (conv_gemm_split_kernel<..EPI_K>: 3 840 workgroups of one face x 64 expanded channels, every one a chain of ~8 memory round trips
// whose phases ADD UP -- skeleton 0.108 / operand fetches 0.090 / stores at the HBM roof 0.084 of 0.346 ms, r04_run20_expdw_ablations)
// that re-reads the block input once per 64 channels (15 x at 960) and writes the 960-channel f32 map (0.98 MB per face), then the gated
// projection that reads it back: ~7.5 x the block's algorithmic bytes.
//
// Here the face's INPUT is the stationary operand: 8 waves, wave w owns image rows 2w and 2w + 1 (two 16-pixel MFMA tiles) and keeps
// their pixel fragments of ALL input channels in registers as split f16 hi / lo (KS k-steps x 2 tiles x 8 VGPRs), loaded once per face.
// The expanded channels are walked 32 at a time ("tiles"), two barrier-separated phases per tile, each pairing a matrix-core job with
// a VALU job that does not depend on it:
//   phase a(t)  project(t - 1): out += W2[:, t - 1] . D[(t - 1) & 1] on the matrix cores, the accumulators (2 tiles x NTO x 4 VGPRs)
//               living in registers for the whole face
//             | depthwise of tile t on E out of LDS: thread = (channel pair, image row, half row), k x k x 16 f32 fma in the order of
//               the unfused kernel -- as SCALAR v_fma / v_fmac: this kernel lives in its own translation unit (mbx_launch.cpp, built
//               with -fno-slp-vectorize) because hipcc's SLP vectoriser packs the channel pair into v_pk_fma_f32, which measured
//               ~4x slower per flop here -- then + activation and MODE 1 (squeeze pass of an SE block): per-thread sums -> LDS ->
//               per-face channel means, nothing else; MODE 0 / 2: (x SE gate) -> split hi / lo -> the pixel-operand planes D[t & 1];
//   phase b(t)  expand(t + 1): E = act(W1[t + 1] . x + b1) (3 x v_mfma_f32_16x16x32_f16 per product, f32 accumulate; the 32 x (KS x 32)
//               pre-split weight rows from a 20 KB LDS stage) -> LDS as f32 [256 px][36].
// In phase a waves 0-3 run the matrix job first and the VALU job second, waves 4-7 the other way round: waves w and w + 4 share a
// SIMD, so its matrix pipe and its VALU are busy at the same time without instruction-level interleaving.  (A cut that carried the
// tap results over the barrier to finish them beside expand balanced the phases better on paper, and spilled the output accumulators
// to scratch inside the projection's MFMA chain: 16 more live registers next to 160 resident ones.)
// Everything that comes from memory inside the loop arrives by LDS-DMA issued at the START of a phase for the NEXT phase that reads
// it (W1 / taps / biases / gate of tile t + 1 during phase a(t), W2 of tile t during phase b(t)), so every barrier is a plain
// "vmcnt(0) + s_barrier": no hand-counted partial waits in this kernel.
// First cut (round 5, profiles/r05_run3_mbx_phase_cycles_first_cut.txt): expand | barrier | depthwise + project | barrier with the
// residual added in the epilogue ran 0.278 ms per 256 faces for a 160 -> 960 -> 160 block (the two launches it replaced: 0.254):
// v_pk_fma_f32 taps, a 52 us epilogue of twenty dependent residual round trips, twenty dependent input round trips in the prologue.
// An SE block is TWO launches around its two small FC launches: MODE 1 (expand + depthwise -> means only), then MODE 2, which
// RECOMPUTES expand + depthwise (the input is in registers, the weights in L2: no HBM bytes) and projects the gated result.  Per
// 160 -> 960 -> 160 block that is 2 x 24 + 24 us of matrix work instead of 0.5 GB of HBM traffic; algorithmic bytes only reach HBM:
// the 160-channel input in, the 160-channel output out.
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

template <int KS, int NTO, int K, int DIL, int MODE>
__global__ __launch_bounds__(512, 2) void mbx_kernel(MbxArgs a) {
    constexpr int PAD = DIL * (K - 1) / 2;
    constexpr int ES = 36;                                  // floats per E pixel row: the two half rows (8 columns apart) read disjoint banks (ds_read_b64), 8 pixels x 16-byte stores cover all 32
    constexpr int E_BYTES = 256 * ES * 4;
    constexpr int D_BYTES = 32768;                          // hi plane 256 x 64 B + lo plane
    constexpr int W1_BYTES = KS * 4096;
    constexpr int COUT = NTO * 16;
    constexpr int W2_BYTES = COUT * 128;
    constexpr int CT_FLOATS = (K * K + 2) * 32;
    constexpr int CT_SLOTS = ((CT_FLOATS / 4 + 63) / 64) * 64;   // 16-byte slots, whole waves
    constexpr int CT_BYTES = CT_SLOTS * 16 + 1024;          // + the gate's wave (32 floats used)
    constexpr int GATE_OFF = CT_SLOTS * 16;
    constexpr int NACC = MODE == 1 ? 1 : NTO;
    static_assert(PAD >= 1 && PAD <= 4 && (K == 3 || K == 5), "depthwise window");
    static_assert(E_BYTES + 2 * D_BYTES + W1_BYTES + W2_BYTES + 2 * CT_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[E_BYTES + 2 * D_BYTES + W1_BYTES + W2_BYTES + 2 * CT_BYTES];
    float* const es = reinterpret_cast<float*>(smem);
    unsigned char* const dbase = smem + E_BYTES;
    float* const psum = reinterpret_cast<float*>(dbase);    // MODE 1: [2][32 partials][32 channels] over the (unused) D planes
    unsigned char* const w1s = dbase + 2 * D_BYTES;
    unsigned char* const w2s = w1s + W1_BYTES;
    unsigned char* const cts = w2s + W2_BYTES;
    PF_EMU_POISON(smem);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int T = a.T;
    unsigned amax = 0;                                      // range guard (pf_common.h): everything this launch splits
    const unsigned amax_seen = pf_amax_seen(a.range_slot);

    // ---- LDS-DMA streams (16-byte slots, whole waves; the slot of lane l of a wave-instruction is base + 16 l) ------------------
    // Every thread-local index below is derived from an "opaque" copy of the thread index (pf_opaque: an empty asm the compiler must
    // assume changes it), once per call: left alone, the loop-invariant per-lane addresses of ALL phases (~40 of them, the DMA sources
    // as 64-bit pointers) are hoisted out of the tile loop, and next to 160 resident fragment / accumulator registers they spill
    // (first build: 616 bytes of scratch per lane).  Recomputing them costs a few dozen VALU instructions per phase.
    auto glds = [&](const void* sb, unsigned voff, void* dst) { pf_glds16_raw_soff<0>(sb, voff, dst); };
    auto dma_w1 = [&](int tile) {                           // slot -> [k-step][plane][row][position], chunk rotation on the SOURCE
        const int tt = pf_opaque(t);
        const unsigned char* sb = a.w1 + (size_t)tile * (32 * KS * 128);
#pragma unroll
        for (int r = 0; r < (KS * 256 + 511) / 512; ++r) {
            const int sl = r * 512 + tt;
            if (sl < KS * 256) {
                const int s = sl >> 8, plane = (sl >> 7) & 1, row = (sl >> 2) & 31;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                glds(sb, (unsigned)((row * KS + s) * 128 + plane * 64 + chunk * 16), w1s + (size_t)sl * 16);
            }
        }
    };
    auto dma_w2 = [&](int tile) {                           // slot -> [plane][row][position]
        const int tt = pf_opaque(t);
        const unsigned char* sb = a.w2 + (size_t)tile * 128;
#pragma unroll
        for (int r = 0; r < (COUT * 8 + 511) / 512; ++r) {
            const int sl = r * 512 + tt;
            if (sl < COUT * 8) {
                const int plane = sl >= COUT * 4 ? 1 : 0;
                const int row = (sl - plane * COUT * 4) >> 2;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                glds(sb, (unsigned)(row * T * 128 + plane * 64 + chunk * 16), w2s + (size_t)sl * 16);
            }
        }
    };
    auto dma_ct = [&](int tile, int face) {                 // taps | b1 | b_dw of the tile (+ the face's gate values of its 32 channels)
        const int tt = pf_opaque(t);
        unsigned char* dst = cts + (tile & 1) * CT_BYTES;
        if (tt < CT_SLOTS) {
            const int sl = tt < CT_FLOATS / 4 ? tt : 0;     // padding slots of the last wave re-read slot 0
            glds(a.ctile + (size_t)tile * CT_FLOATS, (unsigned)(sl * 16), dst + (size_t)tt * 16);
        } else if (MODE == 2 && tt < CT_SLOTS + 64) {       // the next wave: the gate values, slots GATE_OFF / 16 ... (8 distinct ones)
            glds(a.gate + (size_t)face * a.CEXP + tile * 32, (unsigned)((tt & 7) * 16), dst + (size_t)tt * 16);
        }
    };

    const bool prof = (pf_dbg(a) & 64) != 0;
    unsigned long long c_pro = 0, c_mma = 0, c_wa = 0, c_dwc = 0, c_dwf = 0, c_wb = 0, c_epi = 0;
    for (int face = blockIdx.x; face < a.B; face += gridDim.x) {
        const unsigned long long q0 = prof ? pf_clock() : 0;
        dma_w1(0);
        dma_ct(0, face);
        pf_f32x4 oacc[2][NACC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NACC; ++j) oacc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- the face's input -> split pixel fragments in registers (lane = pixel pcol of the row, k-group kg): one batch of KS x 2
        // unconditional 16-byte loads per image row (channels beyond inC read the pixel's first channels and are zeroed) -------------------
        pf_half8 xh[2][KS], xl[2][KS];
        {
            const int pcol = lane & 15, kg = lane >> 4;
            const float* xin = a.in + ((size_t)face * 256 + wave * 32 + pcol) * a.inLd;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
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
                        xl[i][s][e] = (pf_half)(v - (float)hv);
                        amax = pf_amax(amax, v);
                    }
                }
            }
        }

        // ---- expand(tile): E = act(W1[tile] . x + b1) ----------------------------------------------------------------------------------
        auto expand = [&](int tile) {
            const int tt = pf_opaque(t);
            const int pcol = tt & 15, kg = (tt >> 4) & 3, wv = tt >> 6;
            const float* ct = reinterpret_cast<const float*>(cts + (tile & 1) * CT_BYTES);
            pf_f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int off = s * 4096 + pf_lds_chunk_off(j * 16 + pcol, kg);
                    const pf_half8 wh = *reinterpret_cast<const pf_half8*>(w1s + off);
                    const pf_half8 wl = *reinterpret_cast<const pf_half8*>(w1s + 2048 + off);
                    if (pf_dbg(a) & 4) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wl, xh[i][s], acc[i][j]);      // small terms first
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wh, xl[i][s], acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wh, xh[i][s], acc[i][j]);
                    asm volatile("" ::: "memory");          // (register footprint: two weight fragments at a time)
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(ct + K * K * 32 + j * 16 + kg * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    pf_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[i][j][r], a.scale1, bv[r]);
                    pf_act_rh<4>(v, a.act);
                    *reinterpret_cast<pf_f32x4*>(es + (wv * 32 + i * 16 + pcol) * ES + j * 16 + kg * 4) = v;
                }
            }
        };
        // ---- project(tile): out += W2[:, tile] . D ----------------------------------------------------------------------------------------
        auto project = [&](int tile) {
            if constexpr (MODE != 1) {
                const int tt = pf_opaque(t);
                const int pcol = tt & 15, kg = (tt >> 4) & 3, wv = tt >> 6;
                const unsigned char* dsrc = dbase + (tile & 1) * D_BYTES;
                pf_half8 dh[2], dl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = pf_lds_chunk_off(wv * 32 + i * 16 + pcol, kg);
                    dh[i] = *reinterpret_cast<const pf_half8*>(dsrc + off);
                    dl[i] = *reinterpret_cast<const pf_half8*>(dsrc + 16384 + off);
                }
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const int off = pf_lds_chunk_off(j * 16 + pcol, kg);
                    const pf_half8 wh = *reinterpret_cast<const pf_half8*>(w2s + off);
                    const pf_half8 wl = *reinterpret_cast<const pf_half8*>(w2s + COUT * 64 + off);
                    if (pf_dbg(a) & 4) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i) oacc[i][j] = pf_mfma_16x16x32_f16(wl, dh[i], oacc[i][j]);     // small terms first
#pragma unroll
                    for (int i = 0; i < 2; ++i) oacc[i][j] = pf_mfma_16x16x32_f16(wh, dl[i], oacc[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) oacc[i][j] = pf_mfma_16x16x32_f16(wh, dh[i], oacc[i][j]);
                    asm volatile("" ::: "memory");          // one channel tile's weight fragments in flight at a time (register footprint)
                }
            }
        };
        // ---- depthwise taps of a tile: E -> 8 pixels x 2 channels per thread, bias included, in registers -----------------------------------
        auto depthwise = [&](int tile) {
            float of[16];                                   // of[2 x + c]: pixel 8 xhalf + x of row yrow, channel c2 + c
            const int tt = pf_opaque(t);
            const int c2 = (tt & 15) * 2, xhalf = (tt >> 4) & 1, yrow = tt >> 5;
            const float* ct = reinterpret_cast<const float*>(cts + (tile & 1) * CT_BYTES);
            const pf_f32x2 bd = *reinterpret_cast<const pf_f32x2*>(ct + (K * K + 1) * 32 + c2);
#pragma unroll
            for (int x = 0; x < 8; ++x) { of[2 * x] = bd[0]; of[2 * x + 1] = bd[1]; }
            const bool right_half = xhalf != 0;
            // the PAD columns beside this half: the other half's (columns 8 .. 8 + PAD - 1 for the left half, 8 - PAD .. 7 for the right
            // half); the columns on its outer side lie outside the image and read as zero
            const int side_col = right_half ? 8 - PAD : 8;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yy = yrow + ky * DIL - PAD;
                const bool rok = (unsigned)yy < 16u;
                const int yc = yy < 0 ? 0 : (yy > 15 ? 15 : yy);
                const float* erow = es + (yc * 16) * ES + c2;
                pf_f32x2 in[8 + 2 * PAD];                   // in[PAD + j] = column 8 xhalf + j, j = -PAD .. 7 + PAD
#pragma unroll
                for (int j = 0; j < 8; ++j) in[PAD + j] = *reinterpret_cast<const pf_f32x2*>(erow + (8 * xhalf + j) * ES);
#pragma unroll
                for (int s = 0; s < PAD; ++s) {
                    const pf_f32x2 sv = *reinterpret_cast<const pf_f32x2*>(erow + (side_col + s) * ES);
                    in[s] = right_half ? sv : pf_f32x2{0.f, 0.f};
                    in[PAD + 8 + s] = right_half ? pf_f32x2{0.f, 0.f} : sv;
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    pf_f32x2 w = *reinterpret_cast<const pf_f32x2*>(ct + (ky * K + kx) * 32 + c2);
                    if (!rok) w = pf_f32x2{0.f, 0.f};       // a filter row above / below the image (zero padding): its taps contribute nothing
                    if (pf_dbg(a) & 2) continue;
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        of[2 * x] = fmaf(w[0], in[x + kx * DIL][0], of[2 * x]);
                        of[2 * x + 1] = fmaf(w[1], in[x + kx * DIL][1], of[2 * x + 1]);
                    }
                }
                asm volatile("" ::: "memory");              // one filter row's LDS reads in flight at a time (register footprint)
            }
            // ---- activation -> sums (MODE 1) or gated split planes D[tile & 1] ----
            pf_act_rh<16>(of, a.act);
            if constexpr (MODE == 1) {
                pf_f32x2 rs = pf_f32x2{0.f, 0.f};
#pragma unroll
                for (int x = 0; x < 8; ++x) { rs[0] += of[2 * x]; rs[1] += of[2 * x + 1]; }
                *reinterpret_cast<pf_f32x2*>(psum + (tile & 1) * 1024 + (yrow * 2 + xhalf) * 32 + c2) = rs;
            } else {
                float g0 = 1.f, g1 = 1.f;
                if constexpr (MODE == 2) {
                    const pf_f32x2 g = *reinterpret_cast<const pf_f32x2*>(reinterpret_cast<const float*>(cts + (tile & 1) * CT_BYTES + GATE_OFF) + c2);
                    g0 = g[0]; g1 = g[1];
                }
                // pixel-operand row of pixel P0 + x (P0 = 16 yrow + 8 xhalf, a multiple of 8): the chunk rotation of pf_lds_chunk_off
                // depends on x only through x >> 2, so two base addresses + compile-time offsets cover the eight stores
                unsigned char* dp = dbase + (tile & 1) * D_BYTES + (yrow * 16 + 8 * xhalf) * 64 + (c2 & 7) * 2;
                unsigned char* const dp0 = dp + (((c2 >> 3)) & 3) * 16;
                unsigned char* const dp1 = dp + (((c2 >> 3) + 2) & 3) * 16;
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float v0 = MODE == 2 ? of[2 * x] * g0 : of[2 * x], v1 = MODE == 2 ? of[2 * x + 1] * g1 : of[2 * x + 1];
                    pf_half2 hi, lo;
                    hi[0] = (pf_half)v0; hi[1] = (pf_half)v1;
                    lo[0] = (pf_half)(v0 - (float)hi[0]); lo[1] = (pf_half)(v1 - (float)hi[1]);
                    amax = pf_amax(pf_amax(amax, v0), v1);
                    unsigned char* q = (x < 4 ? dp0 : dp1) + x * 64;
                    *reinterpret_cast<pf_half2*>(q) = hi;
                    *reinterpret_cast<pf_half2*>(q + 16384) = lo;
                }
            }
        };
        auto squeeze = [&](int tile) {                      // MODE 1: the 32 partial sums per channel of a finished tile, in a fixed order
            if (t < 32) {
                const float* ps = psum + (tile & 1) * 1024 + t;
                float tot = 0.f;
#pragma unroll
                for (int q = 0; q < 32; ++q) tot += ps[q * 32];
                const int c = tile * 32 + t;
                if (c < a.CEXP) a.gap_out[(size_t)face * a.CEXP + c] = tot / 256.f;
            }
        };

        pf_wait_vm_barrier<0>();                            // W1(0), constants(0) have landed (and the input / residual loads with them)
        expand(0);
        pf_wait_vm_barrier<0>();
        if (prof) c_pro += pf_clock() - q0;
        for (int tile = 0; tile < T; ++tile) {
            // ======== phase a: project(tile - 1) | depthwise of tile -> D[tile & 1]; W1 / constants of tile + 1 on their way ==================
            const unsigned long long q1 = prof ? pf_clock() : 0;
            if (tile + 1 < T && !((pf_dbg(a) & 1) && tile > 0)) { dma_w1(tile + 1); dma_ct(tile + 1, face); }
            unsigned long long q2 = q1, q3 = q1;
            if (wave < 4) {
                if (tile >= 1) project(tile - 1);
                if (prof) q2 = pf_clock();
                depthwise(tile);
                if (prof) { q3 = pf_clock(); c_mma += q2 - q1; c_dwc += q3 - q2; }
            } else {
                depthwise(tile);
                if (prof) q2 = pf_clock();
                if (tile >= 1) project(tile - 1);
                if (prof) { q3 = pf_clock(); c_dwc += q2 - q1; c_mma += q3 - q2; }
            }
            pf_wait_vm_barrier<0>();
            // ======== phase b: expand(tile + 1) -> E; W2 of tile on its way =================================================================
            const unsigned long long q4 = prof ? pf_clock() : 0;
            if constexpr (MODE != 1) { if (!((pf_dbg(a) & 1) && tile > 0)) dma_w2(tile); }
            if (tile + 1 < T) expand(tile + 1);
            if constexpr (MODE == 1) squeeze(tile);
            const unsigned long long q6 = prof ? pf_clock() : 0;
            pf_wait_vm_barrier<0>();
            if (prof) { c_wa += q4 - q3; c_dwf += q6 - q4; c_wb += pf_clock() - q6; }
        }
        const unsigned long long q7 = prof ? pf_clock() : 0;
        if constexpr (MODE != 1) {
            // ---- block output = acc * scale2 + bias (+ residual), no activation (timm InvertedResidual: the projection is linear).  The
            // residual vectors are requested BEFORE the last tile's projection -- all of them at once (the input fragments are dead by
            // now, so the registers are there): one round trip hidden behind 60 MFMAs.  The first cut added them load by load between
            // the stores (the compiler cannot move a load above a store that may alias it): twenty dependent round trips, 52 us per launch.
            const int tt = pf_opaque(t);
            const int pcol = tt & 15, kg = (tt >> 4) & 3, wv = tt >> 6;
            pf_f32x4 rv[2][NTO];
            if (a.res) {
                const float* __restrict__ rrow = a.res + ((size_t)face * 256 + wv * 32 + pcol) * a.resLd + kg * 4;
#pragma unroll
                for (int j = 0; j < NTO; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) rv[i][j] = *reinterpret_cast<const pf_f32x4*>(rrow + (size_t)i * 16 * a.resLd + j * 16);
            }
            project(T - 1);
            float* __restrict__ orow = a.out + ((size_t)face * 256 + wv * 32 + pcol) * a.outLd + kg * 4;
            if (!(pf_dbg(a) & 16))
#pragma unroll
            for (int j = 0; j < NTO; ++j) {
                const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(a.b2 + j * 16 + kg * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    pf_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(oacc[i][j][r], a.scale2, bv[r]);
                    if (a.res) v += rv[i][j];
                    *reinterpret_cast<pf_f32x4*>(orow + (size_t)i * 16 * a.outLd + j * 16) = v;
                }
            }
        }
        if (prof) c_epi += pf_clock() - q7;
        // the next face's first requests (W1 and constants of tile 0) target stages last read before the loop's final barrier; W2 and D,
        // which the trailing project reads, are next written two barriers into the next face
    }
    if (prof && lane == 0) {
        atomicAdd(a.prof + 0, c_pro); atomicAdd(a.prof + 1, c_mma); atomicAdd(a.prof + 2, c_wa); atomicAdd(a.prof + 3, c_dwc);
        atomicAdd(a.prof + 4, c_dwf); atomicAdd(a.prof + 5, c_wb); atomicAdd(a.prof + 6, c_epi); atomicAdd(a.prof + 7, 1ull);
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}
