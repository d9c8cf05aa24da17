// The stem conv of the landmark networks on the matrix cores (round 4).  (Round 4's lm_front_kernel -- conv_stem + blocks.0.0 + blocks.1.0
// in one launch, 13 barrier-separated phases on a 64-pixel tile -- lived here too: correct, 0.76 against 0.51 ms for the three launches
// it replaced, never on by default; removed in round 6, when the shallow two-barrier fusion of conv_stem + blocks.0.0 (k_front2.h) took
// its place.  Its measurements stay in DESIGN.md section 9, round 4.)
#pragma once
#include "pf_common.h"
#include "k_det.h"

// ---- the stem conv alone on the matrix cores (round 4) ---------------------------------------------------------------------------------
// conv_stem of the Student (3 -> 16, hard-swish) and of the Teacher's HRNet (3 -> 64, relu): 3x3 stride 2 on the uint8 crop.  The VALU
// kernel (k_layers.h stem_conv_kernel: 27 byte loads and 27 x 16 FMAs per output pixel and 16-channel group) ran at 113 us (Student) /
// 700 us (Teacher) per 256 crops against 54 / 215 us for writing its output.  Same staging and K order as lm_front_kernel above, but a
// SHALLOW kernel: one barrier, small LDS footprint, many workgroups per CU -- the shape that works on this chip.
struct StemMfmaArgs {
    const void* in;           // u8 [B][H][W][3] (1/255 folded into w_u8) or f32 [B][3][H][W]
    float* out;               // [B][OH][OW][outLd], 16 * NT channels
    const pf_half* w_u8; const pf_half* w_f32; const float* bias;      // [16 NT][1][64] in ir.py _stem_k_order
    float s_u8, s_f32;
    int B, H, W, OH, OW, outLd, act, TH, TW, tilesX;
    unsigned* range_slot;
};

template <int NT, int MAXO, int MAXIH, int RS, bool F32IN>
__global__ __launch_bounds__(256) void stem_mfma_kernel(StemMfmaArgs a) {
    constexpr int NTHR = 256, NW = 4;
    __shared__ __attribute__((aligned(16))) pf_half s_ih[MAXIH * RS];
    __shared__ __attribute__((aligned(16))) pf_half s_il[F32IN ? MAXIH * RS : 8];
    PF_EMU_POISON(s_ih); PF_EMU_POISON(s_il);
    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen<false>(a.range_slot);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned m_tw = pf_div_magic(a.TW);               // p / TW of the per-lane tile arithmetic (pf_common.h pf_div_small)
    const int b = blockIdx.y;
    const int oy0 = ((int)blockIdx.x / a.tilesX) * a.TH, ox0 = ((int)blockIdx.x % a.tilesX) * a.TW;
    const int IRW = 2 * a.TW + 1, IRH = 2 * a.TH + 1;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const int mis = (3 * ix0) & 3;                            // odd: ox0 is a multiple of an even TW (host checks TW % 2 == 0)
    const int wb = 3 * ix0 - mis;
    const int frow = lane & 15, g = lane >> 4, g4 = g * 4;
    const int P = a.TH * a.TW;

    pf_half8 wh[NT][1], wl[NT][1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) det_wfrag<1>(F32IN ? a.w_f32 : a.w_u8, nt, lane, wh[nt], wl[nt]);
    const float sc = F32IN ? a.s_f32 : a.s_u8;

    if constexpr (!F32IN) {
        const unsigned char* in8 = static_cast<const unsigned char*>(a.in) + (size_t)b * a.H * a.W * 3;
        const int nwd = (IRW * 3 + mis + 3) / 4, rowb = a.W * 3;
        const unsigned m_nwd = pf_div_magic(nwd);
        constexpr int ITW = (MAXIH * (RS / 4) + NTHR - 1) / NTHR;
        unsigned wv[ITW];
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int i = tid + it * NTHR;
            const int ry = pf_div_small(i, m_nwd), w = i - ry * nwd;
            const int iy = iy0 + ry, bw = wb + 4 * w;
            wv[it] = 0u;
            if (ry < IRH && (unsigned)iy < (unsigned)a.H && bw >= 0 && bw < rowb) wv[it] = *reinterpret_cast<const unsigned*>(in8 + (size_t)iy * rowb + bw);
        }
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int i = tid + it * NTHR;
            const int ry = pf_div_small(i, m_nwd), w = i - ry * nwd;
            if (ry < IRH) {
                pf_half* q = s_ih + ry * RS + 4 * w + 1;
                q[0] = (pf_half)(unsigned short)(wv[it] & 0xffu);
                pf_half2 mid;
                mid[0] = (pf_half)(unsigned short)((wv[it] >> 8) & 0xffu);
                mid[1] = (pf_half)(unsigned short)((wv[it] >> 16) & 0xffu);
                *reinterpret_cast<pf_half2*>(q + 1) = mid;
                q[3] = (pf_half)(unsigned short)(wv[it] >> 24);
                amax = pf_amax(amax, (float)(wv[it] >> 24));
            }
        }
    } else {
        const float* inf = static_cast<const float*>(a.in) + (size_t)b * 3 * a.H * a.W;
        const int row_elems = IRW * 3;
        const unsigned m_row = pf_div_magic(row_elems);
        for (int i = tid; i < IRH * row_elems; i += NTHR) {
            const int ry = pf_div_small(i, m_row), x3 = i - ry * row_elems;
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
    __syncthreads();

    float* out = a.out + (size_t)b * a.OH * a.OW * a.outLd;
    for (int mt = wave; mt < (P + 15) / 16; mt += NW) {
        const int p = mt * 16 + frow;
        const int pc = p < P ? p : 0;
        const int py = pf_div_small(pc, m_tw), px = pc - py * a.TW;
        pf_half8 xh, xl;
        {
            const int base = (2 * py + (g < 3 ? g : 0)) * RS + 6 * px + mis + 1;
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
        const int oy = oy0 + py, ox = ox0 + px;
        const bool ok = p < P && oy < a.OH && ox < a.OW;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            acc = pf_mfma_16x16x32_f16(wl[nt][0], xh, acc);
            if constexpr (F32IN) acc = pf_mfma_16x16x32_f16(wh[nt][0], xl, acc);
            acc = pf_mfma_16x16x32_f16(wh[nt][0], xh, acc);
            const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(a.bias + nt * 16 + g4);
            pf_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[e], sc, bv[e]);
            mb_act<4>(v, a.act);
            if (ok) *reinterpret_cast<pf_f32x4*>(out + ((size_t)oy * a.OW + ox) * a.outLd + nt * 16 + g4) = v;
        }
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}
