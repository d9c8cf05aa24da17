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
// The expanded channels are walked 32 at a time ("tiles"); per tile
//   phase 1   expand: E = act(W1[tile] . x + b1) on the matrix cores (3 x v_mfma_f32_16x16x32_f16 per product, f32 accumulate), the
//             32 x (KS x 32) pre-split weight rows from a 20 KB LDS stage (LDS-DMA), E -> LDS as f32 [256 px][36];
//   barrier
//   phase 2   depthwise k x k (dilation DIL, zero padding) on E out of LDS, thread = (channel pair, image row, half row), f32 fma in
//             the order of the unfused kernel; + bias, activation; then
//               MODE 1 (squeeze pass of an SE block): per-thread sums -> LDS -> per-face channel means (the SE squeeze), nothing else;
//               MODE 0 / 2: (x SE gate) -> split hi / lo -> pixel-operand planes D[tile & 1] in LDS;
//             and, side by side with it, project(tile - 1): out += W2[:, tile - 1] . D[(tile - 1) & 1] on the matrix cores, the
//             accumulators (2 tiles x NTO x 4 VGPRs) living in registers for the whole face.  Waves 0-3 run project first and the
//             depthwise second, waves 4-7 the other way round: waves w and w + 4 share a SIMD, so its matrix pipe and its VALU are
//             busy at the same time without any instruction-level interleaving;
//   barrier
// Everything that comes from memory inside the loop arrives by LDS-DMA issued at the START of a phase for the NEXT phase that reads
// it (W1 / taps / biases / gate of tile + 1 during phase 2, W2 of tile - 1 during phase 1), so every barrier is a plain
// "vmcnt(0) + s_barrier": no hand-counted partial waits in this kernel, and a phase (>= 2 k cycles) hides the DMA's ~400.
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

struct MbxArgs {
    const float* in;           // [B][256][inLd]
    float* out;                // [B][256][outLd]           MODE 0 / 2
    const float* res;          // residual [B][256][resLd] or nullptr
    float* gap_out;            // [B][CEXP] channel means   MODE 1
    const float* gate;         // [B][CEXP] SE gate         MODE 2
    const unsigned char* w1;   // [32 T][KS][hi 32 | lo 32] f16, rows beyond CEXP zero
    const float* ctile;        // [T][K K + 2][32]: depthwise taps, expand bias, depthwise bias
    const unsigned char* w2;   // [COUT][T][hi 32 | lo 32] f16
    const float* b2;           // [COUT]
    int B, inC, inLd, outLd, resLd, T, CEXP, act;
    float scale1, scale2;      // 1 / (power-of-two weight scales)
    unsigned* range_slot;
};

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
    const int pcol = lane & 15, kg = lane >> 4;
    const int T = a.T;
    unsigned amax = 0;                                      // range guard (pf_common.h): everything this launch splits
    const unsigned amax_seen = pf_amax_seen(a.range_slot);

    // ---- LDS-DMA streams (16-byte slots, whole waves; the slot of lane l of a wave-instruction is base + 16 l) ------------------
    // Every thread-local index below is derived from an "opaque" copy of the thread index (pf_opaque: an empty asm the compiler must
    // assume changes it), once per call: left alone, the loop-invariant per-lane addresses of ALL phases (~40 of them, the DMA sources
    // as 64-bit pointers) are hoisted out of the tile loop, and next to 160 resident fragment / accumulator registers they spill
    // (first build: 616 bytes of scratch per lane).  Recomputing them costs a few dozen VALU instructions per phase.
    auto dma_w1 = [&](int tile) {                           // slot -> [k-step][plane][row][position], chunk rotation on the SOURCE
        const int tt = pf_opaque(t);
        const unsigned char* sb = a.w1 + (size_t)tile * (32 * KS * 128);
#pragma unroll
        for (int r = 0; r < (KS * 256 + 511) / 512; ++r) {
            const int sl = r * 512 + tt;
            if (sl < KS * 256) {
                const int s = sl >> 8, plane = (sl >> 7) & 1, row = (sl >> 2) & 31;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                pf_glds16_raw_soff<0>(sb, (unsigned)((row * KS + s) * 128 + plane * 64 + chunk * 16), w1s + (size_t)sl * 16);
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
                pf_glds16_raw_soff<0>(sb, (unsigned)(row * T * 128 + plane * 64 + chunk * 16), w2s + (size_t)sl * 16);
            }
        }
    };
    auto dma_ct = [&](int tile, int face) {                 // taps | b1 | b_dw of the tile (+ the face's gate values of its 32 channels)
        const int tt = pf_opaque(t);
        unsigned char* dst = cts + (tile & 1) * CT_BYTES;
        if (tt < CT_SLOTS) {
            const int sl = tt < CT_FLOATS / 4 ? tt : 0;     // padding slots of the last wave re-read slot 0
            pf_glds16_raw_soff<0>(a.ctile + (size_t)tile * CT_FLOATS, (unsigned)(sl * 16), dst + (size_t)tt * 16);
        }
        if constexpr (MODE == 2) {
            if ((tt >> 6) == 7) pf_glds16_raw_soff<0>(a.gate + (size_t)face * a.CEXP + tile * 32, (unsigned)((tt & 7) * 16), dst + GATE_OFF + (tt & 63) * 16);
        }
    };

    for (int face = blockIdx.x; face < a.B; face += gridDim.x) {
        dma_w1(0);
        dma_ct(0, face);
        // ---- the face's input -> split pixel fragments in registers (lane = pixel pcol of the row, k-group kg) -------------------
        pf_half8 xh[2][KS], xl[2][KS];
        {
            const float* xin = a.in + ((size_t)face * 256 + wave * 32 + pcol) * a.inLd + kg * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int c0 = s * 32 + kg * 8;
                    const float* p = xin + (size_t)i * 16 * a.inLd + s * 32;
                    pf_f32x4 v0 = pf_f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
                    if (c0 < a.inC) v0 = *reinterpret_cast<const pf_f32x4*>(p);
                    if (c0 + 4 < a.inC) v1 = *reinterpret_cast<const pf_f32x4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = e < 4 ? v0[e & 3] : v1[e & 3];
                        const pf_half hv = (pf_half)v;
                        xh[i][s][e] = hv;
                        xl[i][s][e] = (pf_half)(v - (float)hv);
                        amax = pf_amax(amax, v);
                    }
                }
        }
        pf_f32x4 oacc[2][MODE == 1 ? 1 : NTO];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < (MODE == 1 ? 1 : NTO); ++j) oacc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        pf_wait_vm_barrier<0>();

        // ---- project(tile): out += W2[:, tile] . D[tile & 1] ------------------------------------------------------------------------
        auto project = [&](int tile) {
            if constexpr (MODE != 1) {
                const int tt = pf_opaque(t);
                const int pcol = tt & 15, kg = (tt >> 4) & 3, wave = tt >> 6;
                const unsigned char* dp = dbase + (tile & 1) * D_BYTES;
                pf_half8 dh[2], dl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = pf_lds_chunk_off(wave * 32 + i * 16 + pcol, kg);
                    dh[i] = *reinterpret_cast<const pf_half8*>(dp + off);
                    dl[i] = *reinterpret_cast<const pf_half8*>(dp + 16384 + off);
                }
#pragma unroll
                for (int j = 0; j < NTO; ++j) {
                    const int off = pf_lds_chunk_off(j * 16 + pcol, kg);
                    const pf_half8 wh = *reinterpret_cast<const pf_half8*>(w2s + off);
                    const pf_half8 wl = *reinterpret_cast<const pf_half8*>(w2s + COUT * 64 + off);
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
        // ---- depthwise(tile): E -> bias, activation -> sums (MODE 1) or gated split planes D[tile & 1] ----------------------------------
        auto depthwise = [&](int tile) {
            const int tt = pf_opaque(t);
            const int c2 = (tt & 15) * 2, xhalf = (tt >> 4) & 1, yrow = tt >> 5;
            const float* ct = reinterpret_cast<const float*>(cts + (tile & 1) * CT_BYTES);
            const pf_f32x2 bd = *reinterpret_cast<const pf_f32x2*>(ct + (K * K + 1) * 32 + c2);
            pf_f32x2 o[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) o[x] = bd;
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
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        o[x][0] = fmaf(w[0], in[x + kx * DIL][0], o[x][0]);
                        o[x][1] = fmaf(w[1], in[x + kx * DIL][1], o[x][1]);
                    }
                }
                asm volatile("" ::: "memory");              // one filter row's LDS reads in flight at a time (register footprint)
            }
            float of[16];
#pragma unroll
            for (int x = 0; x < 8; ++x) { of[2 * x] = o[x][0]; of[2 * x + 1] = o[x][1]; }
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

        for (int tile = 0; tile <= T; ++tile) {
            // ======== phase 1: expand(tile) -> E; W2(tile - 1) on its way =========================================================================
            if constexpr (MODE != 1) { if (tile >= 1) dma_w2(tile - 1); }
            if (tile < T) {
                const int tt = pf_opaque(t);
                const int pcol = tt & 15, kg = (tt >> 4) & 3, wave = tt >> 6;
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
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wl, xh[i][s], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wh, xl[i][s], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][j] = pf_mfma_16x16x32_f16(wh, xh[i][s], acc[i][j]);
                        asm volatile("" ::: "memory");      // (register footprint: two weight fragments at a time)
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
                        *reinterpret_cast<pf_f32x4*>(es + (wave * 32 + i * 16 + pcol) * ES + j * 16 + kg * 4) = v;
                    }
                }
            }
            if constexpr (MODE == 1) {
                if (tile >= 1 && t < 32) {                  // the squeeze of tile - 1: 32 partial sums per channel in a fixed order
                    const float* ps = psum + ((tile - 1) & 1) * 1024 + t;
                    float tot = 0.f;
#pragma unroll
                    for (int q = 0; q < 32; ++q) tot += ps[q * 32];
                    const int c = (tile - 1) * 32 + t;
                    if (c < a.CEXP) a.gap_out[(size_t)face * a.CEXP + c] = tot / 256.f;
                }
                if (tile == T) break;
            }
            pf_wait_vm_barrier<0>();
            // ======== phase 2: depthwise(tile) beside project(tile - 1); W1 / constants of tile + 1 on their way ====================================
            if (tile + 1 < T) { dma_w1(tile + 1); dma_ct(tile + 1, face); }
            if (wave < 4) {
                if (tile >= 1) project(tile - 1);
                if (tile < T) depthwise(tile);
            } else {
                if (tile < T) depthwise(tile);
                if (tile >= 1) project(tile - 1);
            }
            pf_wait_vm_barrier<0>();
        }
        // ---- block output: + bias (+ residual), no activation (timm InvertedResidual: the projection is linear) ----------------------------
        if constexpr (MODE != 1) {
            const int tt = pf_opaque(t);
            const int pcol = tt & 15, kg = (tt >> 4) & 3, wave = tt >> 6;
            float* orow = a.out + ((size_t)face * 256 + wave * 32 + pcol) * a.outLd + kg * 4;
            const float* rrow = a.res ? a.res + ((size_t)face * 256 + wave * 32 + pcol) * a.resLd + kg * 4 : nullptr;
#pragma unroll
            for (int j = 0; j < NTO; ++j) {
                const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(a.b2 + j * 16 + kg * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    pf_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(oacc[i][j][r], a.scale2, bv[r]);
                    if (rrow) v += *reinterpret_cast<const pf_f32x4*>(rrow + (size_t)i * 16 * a.resLd + j * 16);
                    *reinterpret_cast<pf_f32x4*>(orow + (size_t)i * 16 * a.outLd + j * 16) = v;
                }
            }
        }
        // the next face's first DMA targets (W1, constants of tile 0) were last read before the loop's final barrier
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}
