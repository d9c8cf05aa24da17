// conv_stem + blocks.0.0 of the Student's encoder in ONE launch (round 6; timm MobileNetV3Features behind
// TRAIN/face_landmark/lib/core/base_trainer/model.py:252-264: 3x3 stride-2 conv 3 -> 16 + hard-swish, then the depthwise-separable block
// dw 3x3 + ReLU -> 1x1 16 -> 16 (linear) + x on the 128 x 128 map of a 256 x 256 crop).
//
// The two launches it replaces (stem_mfma_kernel, mbconv_wave_f32_kernel "no expand") are each bound by instruction issue, not by the
// 268 MB the first writes and the second reads back per 256 crops (0.094 + 0.160 ms against a 0.09 ms round trip at 6 TB/s;
// profiles/r05_run64_pmc_all_kernels.txt: VALU 65-82 % busy).  Round 4's lm_front_kernel fused three layers in 13 barrier-separated
// phases on a 64-pixel tile and lost (0.76 against 0.51 ms: k_front.h).  This one keeps the shape that works on this chip -- shallow,
// small LDS footprint, five workgroups per CU -- and has TWO barriers:
//   stage   the (2 TH + 5) x (2 TW + 5) pixel region of the uint8 crop -> f16 rows in LDS (a uint8 is exact in f16: no lo plane; the
//           float-input seam of pf_landmark_forward stages hi + lo planes);
//   stem    the (TH + 2) x (TW + 2) stem outputs the tile's depthwise taps touch, on the matrix cores (3 -> 16 as a 16 x 32 GEMM per 16
//           pixels, K order of ir.py::_stem_k_order, split-precision weights), hard-swish, -> LDS as f32 [pixel][16]; positions outside
//           the 128 x 128 map are the depthwise conv's ZERO padding, not stem outputs of padded pixels;
//   block   thread = (pixel of a 16-pixel MFMA tile, 4-channel group g): nine 16-byte LDS reads and 36 fma give the depthwise result of
//           channels 4 g .. 4 g + 3, which is exactly this lane's B operand of four v_mfma_f32_16x16x4_f32 steps when the pointwise
//           weights are indexed c = 4 g + kb (exact f32 products, like the kernel it replaces); + bias + the stem output (residual) from
//           LDS -> one 16-byte store.  The 128 x 128 x 16 stem map never exists in HBM.
// Redundant work: (TH + 2)(TW + 2) / (TH TW) = 1.33 x the stem's GEMM (tiny) and image fetches that hit L2.
#pragma once
#include "pf_common.h"
#include "k_det.h"
#ifndef FRONT2_ABL
#define FRONT2_ABL 0      // tools/ub_front2.hip timing ablations (1 no image loads, 2 no stem phase, 4 no block arithmetic, 8 no stores); 0 in the library
#endif

struct Front2Args {
    const void* in;            // u8 [B][H][W][3] (1/255 folded into w_u8) or f32 [B][3][H][W]
    float* out;                // [B][OH][OW][outLd]: blocks.0.0 output, 16 channels
    const pf_half* w_u8; const pf_half* w_f32; const float* b_stem;      // [16][1][hi 32 | lo 32] in ir.py _stem_k_order
    float s_u8, s_f32;
    const float* w_dw;         // [9][16] depthwise taps (BN folded)
    const float* b_dw;         // [16]
    const float* w_pw;         // [16 out][16 in] pointwise (BN folded)
    const float* b_pw;         // [16]
    int B, H, W, OH, OW, outLd, act_stem, tilesX;
    unsigned* range_slot;
};

template <bool F32IN>
__global__ __launch_bounds__(256) void lm_front2_kernel(Front2Args a) {
    constexpr int NTHR = 256, NW = 4, TH = 8, TW = 32;
    constexpr int SH = TH + 2, SW = TW + 2, SP = SH * SW;            // stem outputs per tile: 10 x 34
    constexpr int IRH = 2 * SH + 1, IRW = 2 * SW + 1;                // image region: 21 rows x 69 pixels
    constexpr int RS = 224;                                          // halves per LDS row (IRW * 3 + 3 + 1 <= RS, multiple of 8)
    __shared__ __attribute__((aligned(16))) pf_half s_ih[IRH * RS];
    __shared__ __attribute__((aligned(16))) pf_half s_il[F32IN ? IRH * RS : 8];
    __shared__ __attribute__((aligned(16))) float s_st[SP * 16];
    PF_EMU_POISON(s_ih); PF_EMU_POISON(s_il); PF_EMU_POISON(s_st);
    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen<false>(a.range_slot);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int oy0 = ((int)blockIdx.x / a.tilesX) * TH, ox0 = ((int)blockIdx.x % a.tilesX) * TW;
    const int iy0 = 2 * (oy0 - 1) - 1, ix0 = 2 * (ox0 - 1) - 1;      // image coordinates of the region's first pixel
    const int mis = (3 * ix0) & 3;                                   // 3 (ox0 is a multiple of 32): odd, so fragment reads are 4-byte aligned
    const int wb = 3 * ix0 - mis;
    const int frow = lane & 15, g = lane >> 4, g4 = g * 4;

    // ---- stage the image region -------------------------------------------------------------------------------------------------
    if constexpr (!F32IN) {
        const unsigned char* in8 = static_cast<const unsigned char*>(a.in) + (size_t)b * a.H * a.W * 3;
        constexpr int nwd = (IRW * 3 + 3 + 3) / 4;                   // aligned words covering one region row
        const int rowb = a.W * 3;
        constexpr int ITW = (IRH * nwd + NTHR - 1) / NTHR;
        unsigned wv[ITW];
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int i = tid + it * NTHR;
            const int ry = i / nwd, w = i - ry * nwd;
            const int iy = iy0 + ry, bw = wb + 4 * w;
            wv[it] = 0u;
            if (!(FRONT2_ABL & 1) && ry < IRH && (unsigned)iy < (unsigned)a.H && bw >= 0 && bw < rowb) wv[it] = *reinterpret_cast<const unsigned*>(in8 + (size_t)iy * rowb + bw);
        }
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int i = tid + it * NTHR;
            const int ry = i / nwd, w = i - ry * nwd;
            if (ry < IRH) {
                pf_half* q = s_ih + ry * RS + 4 * w + 1;
                q[0] = (pf_half)(unsigned short)(wv[it] & 0xffu);
                pf_half2 mid;
                mid[0] = (pf_half)(unsigned short)((wv[it] >> 8) & 0xffu);
                mid[1] = (pf_half)(unsigned short)((wv[it] >> 16) & 0xffu);
                *reinterpret_cast<pf_half2*>(q + 1) = mid;
                q[3] = (pf_half)(unsigned short)(wv[it] >> 24);
                amax = pf_amax(amax, (float)(wv[it] >> 24));       // (any byte: the guard only needs the order of magnitude, <= 255)
            }
        }
    } else {
        const float* inf = static_cast<const float*>(a.in) + (size_t)b * 3 * a.H * a.W;
        constexpr int row_elems = IRW * 3;
        for (int i = tid; i < IRH * row_elems; i += NTHR) {
            const int ry = i / row_elems, x3 = i - ry * row_elems;
            const int rx = x3 / 3, ci = x3 - rx * 3;
            const int iy = iy0 + ry, ix = ix0 + rx;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = inf[((size_t)ci * a.H + iy) * a.W + ix];
            const pf_half hv = (pf_half)v;
            s_ih[ry * RS + x3 + mis + 1] = hv;
            s_il[ry * RS + x3 + mis + 1] = pf_split_lo(v, hv);
            amax = pf_amax(amax, v);
        }
    }
    // constants of the block phase: requested now, needed after two barriers
    pf_f32x4 wdw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wdw[k] = *reinterpret_cast<const pf_f32x4*>(a.w_dw + k * 16 + g4);
    const pf_f32x4 bdw = *reinterpret_cast<const pf_f32x4*>(a.b_dw + g4);
    const pf_f32x4 wpw = *reinterpret_cast<const pf_f32x4*>(a.w_pw + frow * 16 + g4);     // A operands: W[m = frow][c = 4 g + kb], kb = 0 .. 3
    const pf_f32x4 bpw = *reinterpret_cast<const pf_f32x4*>(a.b_pw + g4);
    pf_half8 wh[1], wl[1];
    det_wfrag<1>(F32IN ? a.w_f32 : a.w_u8, 0, lane, wh, wl);
    const pf_f32x4 bst = *reinterpret_cast<const pf_f32x4*>(a.b_stem + g4);
    const float sc = F32IN ? a.s_f32 : a.s_u8;
    __syncthreads();

    // ---- stem conv on the (TH + 2) x (TW + 2) positions ---------------------------------------------------------------------------
    for (int mt = wave; mt < ((FRONT2_ABL & 2) ? 0 : (SP + 15) / 16); mt += NW) {
        const int p = mt * 16 + frow;
        const int pc = p < SP ? p : 0;
        const int sy = pc / SW, sx = pc - sy * SW;
        pf_half8 xh, xl;
        {
            const int base = (2 * sy + (g < 3 ? g : 0)) * RS + 6 * sx + mis + 1;
            if (g < 3) {
                const unsigned* q = reinterpret_cast<const unsigned*>(s_ih + base);
                unsigned u[4] = {q[0], q[1], q[2], q[3]};
                memcpy(&xh, u, 16);
                if constexpr (F32IN) {
                    const unsigned* ql = reinterpret_cast<const unsigned*>(s_il + base);
                    unsigned ul[4] = {ql[0], ql[1], ql[2], ql[3]};
                    memcpy(&xl, ul, 16);
                }
            } else {
                xh = pf_half8{s_ih[base + 8], s_ih[base + RS + 8], s_ih[base + 2 * RS + 8], (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0};
                if constexpr (F32IN) xl = pf_half8{s_il[base + 8], s_il[base + RS + 8], s_il[base + 2 * RS + 8], (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0};
            }
        }
        pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        acc = pf_mfma_16x16x32_f16(wl[0], xh, acc);
        if constexpr (F32IN) acc = pf_mfma_16x16x32_f16(wh[0], xl, acc);
        acc = pf_mfma_16x16x32_f16(wh[0], xh, acc);
        pf_f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[e], sc, bst[e]);
        mb_act<4>(v, a.act_stem);
        const int oy = oy0 - 1 + sy, ox = ox0 - 1 + sx;
        if (!((unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW)) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};      // the depthwise conv's zero padding
        if (p < SP) *reinterpret_cast<pf_f32x4*>(s_st + p * 16 + g4) = v;
    }
    __syncthreads();

    // ---- blocks.0.0: depthwise 3x3 + ReLU -> pointwise 16 -> 16 + residual ---------------------------------------------------------------
    float* out = a.out + (size_t)b * a.OH * a.OW * a.outLd;
#pragma unroll
    for (int i = 0; i < (TH * TW / 16) / NW; ++i) {
        const int mt = wave * ((TH * TW / 16) / NW) + i;             // 16 consecutive pixels of one tile row
        const int py = mt >> 1, px = (mt & 1) * 16 + frow;
        const float* sp = s_st + (py * SW + px) * 16 + g4;
        pf_f32x4 d = bdw;
        if (!(FRONT2_ABL & 4))
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const pf_f32x4 xv = *reinterpret_cast<const pf_f32x4*>(sp + (ky * SW + kx) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = fmaf(wdw[ky * 3 + kx][e], xv[e], d[e]);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaxf(d[e], 0.f);
        pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = pf_mfma_16x16x4_f32(wpw[kb], d[kb], acc);
        const pf_f32x4 res = *reinterpret_cast<const pf_f32x4*>(sp + (SW + 1) * 16);
        const int oy = oy0 + py, ox = ox0 + px;
        pf_f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[e] + bpw[e]) + res[e];
        if (!(FRONT2_ABL & 8) && oy < a.OH && ox < a.OW) *reinterpret_cast<pf_f32x4*>(out + ((size_t)oy * a.OW + ox) * a.outLd + g4) = v;
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}
