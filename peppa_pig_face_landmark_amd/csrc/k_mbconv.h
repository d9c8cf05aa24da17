// Fused MobileNetV3 inverted-residual block (timm InvertedResidual without SE, Student encoder
// model.py:252-264):   out = [x +] BN(pwl(act(BN(dw_kxk(act(BN(pw(x))))))))
// One launch per block: the expanded tensor (4-6x the block's input) lives only in LDS.
//
// Workgroup = 512 threads (8 waves) = one TH x TW tile of output pixels of one image:
//   0. the tile's input halo ((TH-1)*S + (K-1)*dil + 1)^2-ish pixels x Cin is staged once in LDS;
//   then, per chunk of 32 expanded channels:
//   1. expand 1x1:  E[halo px][32] = act(Wexp . X + b) on v_mfma_f32_16x16x4_f32 (pixels outside the image
//      are forced to 0: the depthwise conv zero-pads its *input*, i.e. the expanded map);
//   2. depthwise kxk (+bias, act) from E in LDS -> D[tile px][32];
//   3. project 1x1: acc[tile px][Cout] += Wpwl[:, chunk] . D on the same MFMA;
//   epilogue: + bias (+ residual x) -> NHWC store.
// Exact f32 arithmetic (the GEMMs here are tiny: the block is bandwidth bound, its traffic drops from
// x + 2E + 2D + out  to  ~1.4 x + out).
#pragma once
#include "pf_common.h"

struct MbconvArgs {
    const float* in;      // [B][inH][inW][inLd]
    float* out;           // [B][outH][outW][outLd]
    const float* res;     // residual (same pixel grid as out) or nullptr
    const float* w_exp;   // [MidPad][CP]   (row = expanded channel, K = input channel, zero padded)
    const float* b_exp;   // [MidPad]
    const float* w_dw;    // [K*K][MidPad]
    const float* b_dw;    // [MidPad]
    const float* w_pwl;   // [CoutPad][MidPad]
    const float* b_pwl;   // [CoutPad]
    int B, inH, inW, Cin, inLd, outH, outW, Cout, outLd, resLd;
    int MidPad, CoutPad, pad, act, tilesX;
};

// K: depthwise kernel size, S: stride, DIL: dilation, CP: padded input channels (multiple of 16), TH x TW: output tile
template <int K, int S, int DIL, int CP, int TH, int TW>
__global__ __launch_bounds__(512) void mbconv_fused_kernel(MbconvArgs a) {
    constexpr int P = TH * TW;                 // output pixels per tile (multiple of 16)
    constexpr int MP = P / 16;
    constexpr int HH = (TH - 1) * S + (K - 1) * DIL + 1;
    constexpr int HW = (TW - 1) * S + (K - 1) * DIL + 1;
    constexpr int HP = HH * HW;
    constexpr int HPP = (HP + 15) / 16 * 16;   // halo pixels padded to MFMA tiles
    constexpr int MH = HPP / 16;
    constexpr int XS = CP + 4;                 // LDS row strides (floats); +4 keeps 16-lane fragment reads conflict free
    constexpr int ES = 36;
    constexpr int MAXCO = 80;                  // project outputs handled per workgroup
    constexpr int MAXPAIR = (MP * (MAXCO / 16) + 7) / 8;
    static_assert(P % 16 == 0 && CP % 16 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float smem[HPP * XS + HPP * ES + P * ES + 32 * XS + MAXCO * ES];
    float* xs = smem;                 // [HPP][XS]   input halo tile
    float* es = xs + HPP * XS;        // [HPP][ES]   expanded chunk
    float* ds = es + HPP * ES;        // [P][ES]     depthwise output chunk
    float* we = ds + P * ES;          // [32][XS]    expand weights of the chunk
    float* wp = we + 32 * XS;         // [CoutPad][ES] project weights of the chunk

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / a.tilesX) * TH, tx0 = (blockIdx.x % a.tilesX) * TW;
    const int iy0 = ty0 * S - a.pad, ix0 = tx0 * S - a.pad;
    const float* in = a.in + (size_t)b * a.inH * a.inW * a.inLd;

    // ---- 0. input halo tile -> LDS (zeros outside the image / beyond Cin) -----------------------
    for (int i = t; i < HPP * (CP / 4); i += 512) {
        const int hp = i / (CP / 4), c4 = (i - hp * (CP / 4)) * 4;
        pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        if (hp < HP) {
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            if ((unsigned)iy < (unsigned)a.inH && (unsigned)ix < (unsigned)a.inW && c4 < a.Cin)
                v = *reinterpret_cast<const pf_f32x4*>(in + ((size_t)iy * a.inW + ix) * a.inLd + c4);
        }
        *reinterpret_cast<pf_f32x4*>(xs + hp * XS + c4) = v;
    }

    pf_f32x4 acc[MAXPAIR];
#pragma unroll
    for (int q = 0; q < MAXPAIR; ++q) acc[q] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int NTC = a.CoutPad / 16;
    const int frow = lane & 15, fk = (lane >> 4) * 4;

    for (int mc = 0; mc < a.MidPad; mc += 32) {
        // ---- weights of this chunk -> LDS -----------------------------------------------------------
        for (int i = t; i < 32 * (CP / 4); i += 512) {
            const int r = i / (CP / 4), c4 = (i - r * (CP / 4)) * 4;
            *reinterpret_cast<pf_f32x4*>(we + r * XS + c4) = *reinterpret_cast<const pf_f32x4*>(a.w_exp + (size_t)(mc + r) * CP + c4);
        }
        for (int i = t; i < a.CoutPad * 8; i += 512) {
            const int r = i >> 3, c4 = (i & 7) * 4;
            *reinterpret_cast<pf_f32x4*>(wp + r * ES + c4) = *reinterpret_cast<const pf_f32x4*>(a.w_pwl + (size_t)r * a.MidPad + mc + c4);
        }
        __syncthreads();
        // ---- 1. expand: (MH x 2) tiles of 16x16 over the 8 waves -------------------------------------
        for (int q = wave; q < MH * 2; q += 8) {
            const int mt = q >> 1, nt = q & 1;
            pf_f32x4 e = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k0 = 0; k0 < CP; k0 += 16) {
                const pf_f32x4 wv = *reinterpret_cast<const pf_f32x4*>(we + (nt * 16 + frow) * XS + k0 + fk);
                const pf_f32x4 xv = *reinterpret_cast<const pf_f32x4*>(xs + (mt * 16 + frow) * XS + k0 + fk);
#pragma unroll
                for (int j = 0; j < 4; ++j) e = pf_mfma_16x16x4_f32(wv[j], xv[j], e);
            }
            const int hp = mt * 16 + (lane & 15);
            const int n = nt * 16 + (lane >> 4) * 4;
            bool inside = false;
            if (hp < HP) {
                const int hy = hp / HW, hx = hp - hy * HW;
                inside = (unsigned)(iy0 + hy) < (unsigned)a.inH && (unsigned)(ix0 + hx) < (unsigned)a.inW;
            }
            pf_f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = inside ? pf_act(e[r] + a.b_exp[mc + n + r], a.act) : 0.f;
            *reinterpret_cast<pf_f32x4*>(es + hp * ES + n) = o;
        }
        __syncthreads();
        // ---- 2. depthwise: thread = (channel c, pixel group) -------------------------------------------
        {
            const int c = t & 31, pg = t >> 5;
            float wk[K * K];
#pragma unroll
            for (int k = 0; k < K * K; ++k) wk[k] = a.w_dw[(size_t)k * a.MidPad + mc + c];
            const float bd = a.b_dw[mc + c];
#pragma unroll
            for (int i = 0; i < P / 16; ++i) {
                const int p = pg + 16 * i;
                const int py = p / TW, px = p - py * TW;
                float s = bd;
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
                        s = fmaf(wk[ky * K + kx], es[((py * S + ky * DIL) * HW + px * S + kx * DIL) * ES + c], s);
                ds[p * ES + c] = pf_act(s, a.act);
            }
        }
        __syncthreads();
        // ---- 3. project: (MP x NTC) tiles over the 8 waves, K = 32 ----------------------------------------
#pragma unroll
        for (int qi = 0; qi < MAXPAIR; ++qi) {
            const int q = wave + 8 * qi;
            if (q < MP * NTC) {
                const int mt = q % MP, nt = q / MP;
#pragma unroll
                for (int k0 = 0; k0 < 32; k0 += 16) {
                    const pf_f32x4 wv = *reinterpret_cast<const pf_f32x4*>(wp + (nt * 16 + frow) * ES + k0 + fk);
                    const pf_f32x4 dv = *reinterpret_cast<const pf_f32x4*>(ds + (mt * 16 + frow) * ES + k0 + fk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[qi] = pf_mfma_16x16x4_f32(wv[j], dv[j], acc[qi]);
                }
            }
        }
        __syncthreads();
    }
    // ---- epilogue ----------------------------------------------------------------------------------
#pragma unroll
    for (int qi = 0; qi < MAXPAIR; ++qi) {
        const int q = wave + 8 * qi;
        if (q >= MP * NTC) continue;
        const int mt = q % MP, nt = q / MP;
        const int p = mt * 16 + (lane & 15);
        const int co = nt * 16 + (lane >> 4) * 4;
        const int oy = ty0 + p / TW, ox = tx0 + p % TW;
        if (oy >= a.outH || ox >= a.outW || co >= a.Cout) continue;
        const size_t pix = ((size_t)b * a.outH + oy) * a.outW + ox;
        pf_f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[qi][r] + a.b_pwl[co + r];
        if (a.res) {
            const pf_f32x4 rv = *reinterpret_cast<const pf_f32x4*>(a.res + pix * a.resLd + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        *reinterpret_cast<pf_f32x4*>(a.out + pix * a.outLd + co) = v;
    }
}

// ---- wave-level variant (high-resolution blocks: many pixels, few channels) ---------------------------
// Each WAVE owns a PH x PW patch of output pixels and runs the whole expand -> depthwise -> project chain
// on its own, in chunks of 16 expanded channels, with no workgroup barrier: the workgroup is just four
// independent waves, so a CU keeps 16+ of them in flight and their global-load latencies overlap.
//   * the patch's input halo is held in REGISTERS as MFMA pixel fragments (loaded once);
//   * weights come straight from global memory (a few KB shared by every wave: L1/L2 hits);
//   * E (halo x 16) and D (patch x 16) live in the wave's private 7 KB slice of LDS, unpadded: the
//     fragment stores/loads cover 1 KB contiguously and the depthwise reads hit 64 distinct banks.
template <int S, int CP, int PH, int PW>
__global__ __launch_bounds__(256) void mbconv_wave_kernel(MbconvArgs a) {
    constexpr int K = 3;
    constexpr int PP = PH * PW, MPW = PP / 16;
    constexpr int HH = (PH - 1) * S + K, HW = (PW - 1) * S + K, HP = HH * HW;
    constexpr int MH = (HP + 15) / 16, HPP = MH * 16;
    constexpr int KK = CP / 16;
    constexpr int MAXNT = 2;                   // Cout <= 32
    static_assert(PP % 16 == 0 && (PW == 4 || PW == 8), "patch shape");

    __shared__ __attribute__((aligned(16))) float smem[4][(HPP + PP) * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* es = smem[wave];
    float* ds = es + HPP * 16;

    const int patchesX = (a.outW + PW - 1) / PW, patchesY = (a.outH + PH - 1) / PH;
    const int pid = blockIdx.x * 4 + wave;
    if (pid >= patchesX * patchesY) return;    // whole wave leaves; no workgroup barrier below
    const int b = blockIdx.y;
    const int oy0 = (pid / patchesX) * PH, ox0 = (pid % patchesX) * PW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const float* in = a.in + (size_t)b * a.inH * a.inW * a.inLd;
    const int frow = lane & 15, fk = (lane >> 4) * 4;

    // ---- input halo as MFMA pixel fragments: lane = (pixel frow of tile mt, channels kk*16+fk..+3) ----
    pf_f32x4 xf[MH][KK];
    unsigned inside = 0;
#pragma unroll
    for (int mt = 0; mt < MH; ++mt) {
        const int hp = mt * 16 + frow;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)a.inH && (unsigned)ix < (unsigned)a.inW;
        inside |= ok ? (1u << mt) : 0u;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            xf[mt][kk] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok && kk * 16 + fk < a.Cin)
                xf[mt][kk] = *reinterpret_cast<const pf_f32x4*>(in + ((size_t)iy * a.inW + ix) * a.inLd + kk * 16 + fk);
        }
    }
    pf_f32x4 acc[MPW][MAXNT];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int j = 0; j < MAXNT; ++j) acc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int NTC = a.CoutPad / 16;
    const int dc = lane & 15, dg = lane >> 4;  // depthwise role: channel dc, pixel phase dg

#pragma unroll 1
    for (int mc = 0; mc < a.MidPad; mc += 16) {
        // ---- 1. expand -> E ---------------------------------------------------------------------------------
        pf_f32x4 wv[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            wv[kk] = *reinterpret_cast<const pf_f32x4*>(a.w_exp + (size_t)(mc + frow) * CP + kk * 16 + fk);
        const pf_f32x4 be = *reinterpret_cast<const pf_f32x4*>(a.b_exp + mc + fk);
#pragma unroll
        for (int mt = 0; mt < MH; ++mt) {
            pf_f32x4 e = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) e = pf_mfma_16x16x4_f32(wv[kk][j], xf[mt][kk][j], e);
            const bool ok = (inside >> mt) & 1u;
            pf_f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = ok ? pf_act(e[r] + be[r], a.act) : 0.f;
            *reinterpret_cast<pf_f32x4*>(es + (mt * 16 + frow) * 16 + fk) = o;
        }
        pf_wave_sync();
        // ---- 2. depthwise 3x3 -> D ------------------------------------------------------------------------------
        {
            float wk[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = a.w_dw[(size_t)k * a.MidPad + mc + dc];
            const float bd = a.b_dw[mc + dc];
#pragma unroll
            for (int i = 0; i < PP / 4; ++i) {
                const int px = PW == 4 ? dg : dg + 4 * (i & 1);
                const int py = PW == 4 ? i : i >> 1;
                float s = bd;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        s = fmaf(wk[ky * 3 + kx], es[((py * S + ky) * HW + px * S + kx) * 16 + dc], s);
                ds[(py * PW + px) * 16 + dc] = pf_act(s, a.act);
            }
        }
        pf_wave_sync();
        // ---- 3. project ---------------------------------------------------------------------------------------
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt) {
            if (nt < NTC) {
                const pf_f32x4 pv = *reinterpret_cast<const pf_f32x4*>(a.w_pwl + (size_t)(nt * 16 + frow) * a.MidPad + mc + fk);
#pragma unroll
                for (int mt = 0; mt < MPW; ++mt) {
                    const pf_f32x4 dv = *reinterpret_cast<const pf_f32x4*>(ds + (mt * 16 + frow) * 16 + fk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt][nt] = pf_mfma_16x16x4_f32(pv[j], dv[j], acc[mt][nt]);
                }
            }
        }
    }
    // ---- epilogue ------------------------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < MPW; ++mt)
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt) {
            const int p = mt * 16 + frow;
            const int co = nt * 16 + fk;
            const int oy = oy0 + p / PW, ox = ox0 + p % PW;
            if (nt >= NTC || oy >= a.outH || ox >= a.outW || co >= a.Cout) continue;
            const size_t pix = ((size_t)b * a.outH + oy) * a.outW + ox;
            pf_f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r] + a.b_pwl[co + r];
            if (a.res) {
                const pf_f32x4 rv = *reinterpret_cast<const pf_f32x4*>(a.res + pix * a.resLd + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
            *reinterpret_cast<pf_f32x4*>(a.out + pix * a.outLd + co) = v;
        }
}
