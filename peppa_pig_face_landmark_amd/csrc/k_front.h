// Front end of the Student encoder in ONE launch (round 4): conv_stem (3x3 s2, 3 -> 16, hard-swish) -> blocks.0.0 (depthwise 3x3 +
// relu -> 1x1 16 -> 16, + x) -> blocks.1.0 (1x1 16 -> 64 + relu -> depthwise 3x3 s2 + relu -> 1x1 64 -> 24).
// timm mobilenetv3_large_100 features as the reference builds them (TRAIN/face_landmark/lib/core/base_trainer/model.py:252-264),
// restated in oracle/landmark_net.py.  The three launches this replaces moved the 128 x 128 x 16 maps (1 MB per face, f32) to
// HBM and back three times: ~1.2 GB per 256 faces for 0.5 ms; here only the uint8 crop comes in (196 KB per face) and the
// 64 x 64 x 24 map goes out (393 KB).  A workgroup owns a TH x TW tile of the 64 x 64 OUTPUT:
//   image region (4 TH + 7 rows) -> f16 rows in LDS (k_det.h det_stem_kernel's layout: whole 32-bit words in, K ordered
//   (ky, kx * 3 + ci) so a pixel's operand is four aligned ds_read_b32 per image row)
//   stem on the (2 TH + 3) x (2 TW + 3) region: MFMA, f32 result in LDS (zero outside the map = the depthwise conv's padding)
//   blocks.0.0: depthwise in f32 from LDS -> split planes -> 16 x 16 MFMA GEMM -> + stem (residual) -> split planes (y0)
//   blocks.1.0: per 16 expanded channels: MFMA expand on the (2 TH + 1) x (2 TW + 1) region -> relu, zero outside the map ->
//   f32 in LDS -> depthwise stride 2 -> split planes; then the 64 -> 24 projection GEMM and the only global store.
#pragma once
#include "k_det.h"

struct LmFrontArgs {
    const void* in;           // u8 [B][H][W][3] (1/255 folded into w_stem_u8) or f32 [B][3][H][W]
    float* out;               // [B][OH][OW][outLd], 24 channels (OH = H / 4)
    const pf_half* w_stem_u8; const pf_half* w_stem_f32; const float* b_stem;   // [16][1][64], K order of det_stem_kernel
    const float* w_dw0; const float* b_dw0;      // [9][16], [16]
    const pf_half* w_pw0; const float* b_pw0;    // [16][1][64] (K = 16 of 32)
    const pf_half* w_exp; const float* b_exp;    // [64][1][64]
    const float* w_dw1; const float* b_dw1;      // [9][64], [64]
    const pf_half* w_prj; const float* b_prj;    // [32 (24 used)][2][64]
    float s_stem_u8, s_stem_f32, s_pw0, s_exp, s_prj;
    int B, H, W, SH, SW, OH, OW, outLd, TH, TW, tilesX, act_stem;
    unsigned* range_slot;
    unsigned long long* prof;   // ablation build only (PEPPA_DBG & 4096): cycles per phase [8], tiles
};

// narrow planes: one 64-byte row per pixel = [hi ch 0-7 | hi ch 8-15 | lo ch 0-7 | lo ch 8-15], slots rotated like k_conv_gemm.h
__device__ __forceinline__ void lmf_park16(unsigned char* base, int row, int g, const pf_f32x4& v, unsigned& amax) {
    pf_half4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const pf_half hv = (pf_half)v[e];
        hi[e] = hv;
        lo[e] = pf_split_lo(v[e], hv);
        amax = pf_amax(amax, v[e]);
    }
    *reinterpret_cast<pf_half4*>(base + pf_lds_chunk_off(row, g >> 1) + (g & 1) * 8) = hi;
    *reinterpret_cast<pf_half4*>(base + pf_lds_chunk_off(row, 2 + (g >> 1)) + (g & 1) * 8) = lo;
}
// K = 16 GEMM tile on narrow planes: k groups 2 and 3 are zero
__device__ __forceinline__ pf_f32x4 lmf_tile16(const unsigned char* base, int row0, int lane, const pf_half8& wh, const pf_half8& wl) {
    const int row = row0 + (lane & 15), g = lane >> 4;
    pf_half8 xh = pf_half8{(pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0}, xl = xh;
    if (g < 2) {
        xh = *reinterpret_cast<const pf_half8*>(base + pf_lds_chunk_off(row, g));
        xl = *reinterpret_cast<const pf_half8*>(base + pf_lds_chunk_off(row, 2 + g));
    }
    pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    acc = pf_mfma_16x16x32_f16(wl, xh, acc);
    acc = pf_mfma_16x16x32_f16(wh, xl, acc);
    acc = pf_mfma_16x16x32_f16(wh, xh, acc);
    return acc;
}

// MAXO >= TH * TW (multiple of 16); MAXS >= (2 TH + 3)(2 TW + 3) rounded to 16; MAXY >= (2 TH + 1)(2 TW + 1) rounded to 16;
// MAXIH >= 4 TH + 7; RS halves per image row >= 12 TW + 26
template <int MAXO, int MAXS, int MAXY, int MAXIH, int RS, bool F32IN, int NTHR, int WPS>
__global__ __launch_bounds__(NTHR, WPS) void lm_front_kernel(LmFrontArgs a) {      // WPS = 4: two 512-thread workgroups per CU
    constexpr int NW = NTHR / 64;
    constexpr int IMG_B = MAXIH * RS * 2 * (F32IN ? 2 : 1), DPL_B = 2 * 2 * MAXO * 64;      // image rows | later: D planes (64 ch, tile rows)
    constexpr int PD_B = MAXY * 64;                                                          // ... and in between: blocks.0.0's depthwise output
    constexpr int A_B = (IMG_B > DPL_B ? IMG_B : DPL_B) > PD_B ? (IMG_B > DPL_B ? IMG_B : DPL_B) : PD_B;
    constexpr int ST_B = MAXS * 16 * 4, E_B = MAXY * 20 * 4;                                // stem f32 | later: E chunk f32 [row][16 + 4]
    constexpr int B_B = ST_B > E_B ? ST_B : E_B;
    __shared__ __attribute__((aligned(16))) unsigned char s_a[A_B];
    __shared__ __attribute__((aligned(16))) unsigned char s_b[B_B];
    __shared__ __attribute__((aligned(16))) unsigned char s_py[MAXY * 64];     // y0 (blocks.0.0 output), narrow planes
    __shared__ __attribute__((aligned(16))) float s_w[9 * 16 + 16 + 9 * 64 + 64];   // depthwise weights + biases of both blocks (loaded once: kept in
                                                                               // registers across the tile loop they cost 140 VGPRs and the second workgroup per CU)
    __shared__ unsigned char s_oks[MAXS];                                      // stem-region pixel inside the stem map?
    __shared__ unsigned char s_oky[MAXY];                                      // y0-region pixel inside the map?
    PF_EMU_POISON(s_a); PF_EMU_POISON(s_b); PF_EMU_POISON(s_py); PF_EMU_POISON(s_w); PF_EMU_POISON(s_oks); PF_EMU_POISON(s_oky);
    pf_half* s_ih = reinterpret_cast<pf_half*>(s_a);
    pf_half* s_il = s_ih + MAXIH * RS;                                         // float input only
    float* s_st = reinterpret_cast<float*>(s_b);                               // [stem-region row][16]
    float* s_e = reinterpret_cast<float*>(s_b);                                // [y0-region row][20], after blocks.0.0
    unsigned char* s_pd = s_a;                                                 // blocks.0.0 depthwise output (narrow planes), after the stem
    unsigned char* s_d = s_a;                                                  // D planes of blocks.1.0, after blocks.0.0

    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned m_tw = pf_div_magic(a.TW);               // p / TW of the per-lane tile arithmetic (pf_common.h pf_div_small)
    const int YRW = 2 * a.TW + 1, YRH = 2 * a.TH + 1, YR = YRH * YRW, MRY = (YR + 15) & ~15;      // y0 region
    const int SRW = YRW + 2, SRH = YRH + 2, SR = SRH * SRW, MRS = (SR + 15) & ~15;                // stem region
    const int IRW = 2 * SRW + 1, IRH = 2 * SRH + 1;                                               // image region
    const int P = a.TH * a.TW, MRD = (P + 15) & ~15;
    const int frow = lane & 15, g = lane >> 4, g4 = g * 4;
    // image column of region pixel (0, 0): ix0 = 4 ox0 - 5 (odd), so 3 ix0 = 1 (mod 4) for every tile: region byte e of a row lives
    // at LDS half e + mis + 1 (even at e = 6 rx), the aligned word that holds byte e = 0 starts at half 1
    constexpr int mis = 1;
    // persistent workgroups: tiles t = blockIdx.x, + gridDim.x, ...; the next tile's image words are requested before this one is computed
    const int tiles_y = (a.OH + a.TH - 1) / a.TH, tpf = a.tilesX * tiles_y, ntiles = tpf * a.B;
    const int nwd = (IRW * 3 + mis + 3) / 4;
    const unsigned m_nwd = pf_div_magic(nwd);
    const int rowb = a.W * 3;
    constexpr int ITW = (MAXIH * (RS / 4) + NTHR - 1) / NTHR;
    unsigned wv[ITW];
    auto request = [&](int t) {
        const int tb = t / tpf, tt = t - tb * tpf;
        const int iy0 = 4 * (tt / a.tilesX) * a.TH - 5, wb = 3 * (4 * (tt % a.tilesX) * a.TW - 5) - mis;
        const unsigned char* in8 = static_cast<const unsigned char*>(a.in) + (size_t)tb * a.H * a.W * 3;
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int i = tid + it * NTHR;
            const int ry = pf_div_small(i, m_nwd), w = i - ry * nwd;
            const int iy = iy0 + ry, bw = wb + 4 * w;
            wv[it] = 0u;
            if (ry < IRH && (unsigned)iy < (unsigned)a.H && bw >= 0 && bw < rowb) wv[it] = *reinterpret_cast<const unsigned*>(in8 + (size_t)iy * rowb + bw);
        }
    };
    if constexpr (!F32IN) { if ((int)blockIdx.x < ntiles) request(blockIdx.x); }
    for (int i = tid; i < 9 * 16 + 16 + 9 * 64 + 64; i += NTHR)
        s_w[i] = i < 144 ? a.w_dw0[i] : (i < 160 ? a.b_dw0[i - 144] : (i < 736 ? a.w_dw1[i - 160] : a.b_dw1[i - 736]));

    const float s_stem = F32IN ? a.s_stem_f32 : a.s_stem_u8;

    const bool prof = PF_ABLATE != 0 && a.prof != nullptr;
    unsigned long long tp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    unsigned long long c0 = prof ? pf_clock() : 0, c1;
#define LMF_TICK(k) if (prof) { c1 = pf_clock(); tp[k] += c1 - c0; c0 = c1; }
    const int b = t / tpf, tt = t - b * tpf;
    const int oy0 = (tt / a.tilesX) * a.TH, ox0 = (tt % a.tilesX) * a.TW;
    const int yy0 = 2 * oy0 - 1, yx0 = 2 * ox0 - 1;           // map coordinates of y0-region pixel (0, 0)
    const int sy0 = yy0 - 1, sx0 = yx0 - 1;                   // ... of stem-region pixel (0, 0)
    const int iy0 = 2 * sy0 - 1, ix0 = 2 * sx0 - 1;           // image coordinates of image-region pixel (0, 0)
    // The weight fragments are (re)read from L1 / L2 where each phase needs them.  They do not depend on the tile, so the compiler
    // would hoist every load out of this loop and keep ~100 registers of weights alive across it (191 VGPRs: one workgroup per
    // CU); an offset it cannot see through keeps the loads inside the iteration (<= 128 VGPRs, two workgroups per CU).
    int zt = 0;
#ifdef PF_SIMT_EMULATION
    asm volatile("" : "+r"(zt));
#else
    asm volatile("" : "+s"(zt));
#endif
    pf_half8 wsh[1], wsl[1];
    det_wfrag<1>((F32IN ? a.w_stem_f32 : a.w_stem_u8) + zt, 0, lane, wsh, wsl);
    const pf_f32x4 bsv = *reinterpret_cast<const pf_f32x4*>(a.b_stem + zt + g4);
    // ---- phase 0: image region -> f16 rows ------------------------------------------------------------------------------------
    if constexpr (!F32IN) {
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int i = tid + it * NTHR;
            const int ry = pf_div_small(i, m_nwd), w = i - ry * nwd;
            if (ry < IRH) {
                pf_half* q = s_ih + ry * RS + 4 * w + 1;          // halves 4 w + 1 .. 4 w + 4
                q[0] = (pf_half)(unsigned short)(wv[it] & 0xffu);
                pf_half2 mid;
                mid[0] = (pf_half)(unsigned short)((wv[it] >> 8) & 0xffu);
                mid[1] = (pf_half)(unsigned short)((wv[it] >> 16) & 0xffu);
                *reinterpret_cast<pf_half2*>(q + 1) = mid;
                q[3] = (pf_half)(unsigned short)(wv[it] >> 24);
                amax = pf_amax(amax, (float)(wv[it] >> 24));
            }
        }
        if (t + (int)gridDim.x < ntiles) request(t + gridDim.x);  // in flight across this tile's phases
    } else {
        const float* inf = static_cast<const float*>(a.in) + (size_t)b * 3 * a.H * a.W;
        const int row_elems = IRW * 3;
        for (int i = tid; i < IRH * row_elems; i += NTHR) {
            const int ry = i / row_elems, x3 = i - ry * row_elems;      // (i reaches 23 x 219 here: outside pf_div_small's exact range)
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
    for (int r = tid; r < MRS; r += NTHR) {
        const int ry = r / SRW, rx = r - ry * SRW;
        s_oks[r] = (r < SR && (unsigned)(sy0 + ry) < (unsigned)a.SH && (unsigned)(sx0 + rx) < (unsigned)a.SW) ? 1 : 0;
    }
    for (int r = tid; r < MRY; r += NTHR) {
        const int ry = r / YRW, rx = r - ry * YRW;
        s_oky[r] = (r < YR && (unsigned)(yy0 + ry) < (unsigned)a.SH && (unsigned)(yx0 + rx) < (unsigned)a.SW) ? 1 : 0;
    }
    const int c4 = tid & 3;                                   // depthwise role: 4 channels of a pixel
    __syncthreads();
    LMF_TICK(0)

    // ---- conv_stem on the stem region -> f32 [row][16], zero outside the map ---------------------------------------------------
    for (int mt = wave; mt < MRS / 16; mt += NW) {
        const int r = mt * 16 + frow;
        const int rc = r < SR ? r : 0;
        const int ry = rc / SRW, rx = rc - ry * SRW;
        pf_half8 xh, xl;
        {
            const int base = (2 * ry + (g < 3 ? g : 0)) * RS + 6 * rx + mis + 1;
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
        acc = pf_mfma_16x16x32_f16(wsl[0], xh, acc);
        if constexpr (F32IN) acc = pf_mfma_16x16x32_f16(wsh[0], xl, acc);
        acc = pf_mfma_16x16x32_f16(wsh[0], xh, acc);
        pf_f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[e], s_stem, bsv[e]);
        mb_act<4>(v, a.act_stem);
        if (!s_oks[r]) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<pf_f32x4*>(s_st + r * 16 + g4) = v;
    }
    __syncthreads();
    LMF_TICK(1)

    // ---- blocks.0.0 depthwise 3x3 + relu on the y0 region -> narrow planes ------------------------------------------------------
    {
    pf_f32x4 wd0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wd0[k] = *reinterpret_cast<const pf_f32x4*>(s_w + k * 16 + 4 * c4);
    const pf_f32x4 bd0 = *reinterpret_cast<const pf_f32x4*>(s_w + 144 + 4 * c4);
#pragma unroll 1
    for (int i = tid; i < MRY * 4; i += NTHR) {
        const int r = i >> 2;                                  // (c4 = i & 3 = tid & 3: NTHR % 4 == 0)
        pf_f32x4 sum = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < YR) {
            sum = bd0;
            const int ry = r / YRW, rx = r - ry * YRW;
            const float* e0 = s_st + (ry * SRW + rx) * 16 + 4 * c4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const pf_f32x4 xv = *reinterpret_cast<const pf_f32x4*>(e0 + (ky * SRW + kx) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[e] = fmaf(wd0[ky * 3 + kx][e], xv[e], sum[e]);
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] = sum[e] > 0.f ? sum[e] : 0.f;
        }
        lmf_park16(s_pd, r, c4, sum, amax);
    }
    }
    pf_half8 weh[1], wel[1], wph[1], wpl[1];
    det_wfrag<1>(a.w_pw0 + zt, 0, lane, wph, wpl);
    const pf_f32x4 bpv = *reinterpret_cast<const pf_f32x4*>(a.b_pw0 + zt + g4);
    __syncthreads();
    LMF_TICK(2)

    // ---- blocks.0.0 pointwise 16 -> 16 + residual -> y0 planes ---------------------------------------------------------------------
    for (int mt = wave; mt < MRY / 16; mt += NW) {
        const pf_f32x4 acc = lmf_tile16(s_pd, mt * 16, lane, wph[0], wpl[0]);
        const int r = mt * 16 + frow;
        const int rc = r < YR ? r : 0;
        const int ry = rc / YRW, rx = rc - ry * YRW;
        const pf_f32x4 res = *reinterpret_cast<const pf_f32x4*>(s_st + ((ry + 1) * SRW + rx + 1) * 16 + g4);
        pf_f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[e], a.s_pw0, bpv[e]) + res[e];
        if (r >= YR) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        lmf_park16(s_py, r, g, v, amax);
    }
    __syncthreads();
    LMF_TICK(3)

    // ---- blocks.1.0: expand 16 channels at a time -> relu, zero outside the map -> depthwise stride 2 + relu -> D planes --------
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        det_wfrag<1>(a.w_exp + zt, ch, lane, weh, wel);
        const pf_f32x4 bev = *reinterpret_cast<const pf_f32x4*>(a.b_exp + zt + ch * 16 + g4);
        for (int mt = wave; mt < MRY / 16; mt += NW) {
            const pf_f32x4 acc = lmf_tile16(s_py, mt * 16, lane, weh[0], wel[0]);
            const int r = mt * 16 + frow;
            pf_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = fmaf(acc[e], a.s_exp, bev[e]); v[e] = v[e] > 0.f ? v[e] : 0.f; }
            if (!s_oky[r]) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<pf_f32x4*>(s_e + r * 20 + g4) = v;
        }
        __syncthreads();
        LMF_TICK(4)
        pf_f32x4 wd1[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wd1[k] = *reinterpret_cast<const pf_f32x4*>(s_w + 160 + k * 64 + ch * 16 + 4 * c4);
        const pf_f32x4 bd1 = *reinterpret_cast<const pf_f32x4*>(s_w + 736 + ch * 16 + 4 * c4);
#pragma unroll 1
        for (int i = tid; i < MRD * 4; i += NTHR) {
            const int p = i >> 2;
            pf_f32x4 sum = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < P) {
                sum = bd1;
                const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
                const float* e0 = s_e + ((2 * py) * YRW + 2 * px) * 20 + 4 * c4;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const pf_f32x4 ev = *reinterpret_cast<const pf_f32x4*>(e0 + (ky * YRW + kx) * 20);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sum[e] = fmaf(wd1[ky * 3 + kx][e], ev[e], sum[e]);
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = sum[e] > 0.f ? sum[e] : 0.f;
            }
            det_park4(s_d, MRD, p, ch * 4 + c4, sum, amax);
        }
        __syncthreads();
        LMF_TICK(5)
    }

    // ---- blocks.1.0 projection 64 -> 24 -> global ---------------------------------------------------------------------------------------
    float* out = a.out + (size_t)b * a.OH * a.OW * a.outLd;
    for (int t = wave; t < (MRD / 16) * 2; t += NW) {
        const int mt = t >> 1, nt = t & 1;
        pf_half8 wjh[2], wjl[2];
        det_wfrag<2>(a.w_prj + zt, nt, lane, wjh, wjl);
        const pf_f32x4 acc = det_tile<2>(s_d, MRD, mt * 16, lane, wjh, wjl);
        const int p = mt * 16 + frow;
        const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
        const int oy = oy0 + py, ox = ox0 + px;
        const int n = nt * 16 + g4;
        if (p < P && oy < a.OH && ox < a.OW && n < 24) {
            const pf_f32x4 bj = *reinterpret_cast<const pf_f32x4*>(a.b_prj + zt + n);
            pf_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[e], a.s_prj, bj[e]);
            *reinterpret_cast<pf_f32x4*>(out + ((size_t)oy * a.OW + ox) * a.outLd + n) = v;
        }
    }
    __syncthreads();                                          // the next tile's rows overwrite everything
    LMF_TICK(6)
    if (prof) tp[8] += 1;
    }
#undef LMF_TICK
    pf_amax_commit(a.range_slot, amax, amax_seen);
    if (prof && tid == 0)
        for (int k = 0; k < 9; ++k) atomicAdd(a.prof + k, tp[k]);
}

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
