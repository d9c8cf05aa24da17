// Fused MobileNetV3 inverted-residual block (timm InvertedResidual without SE, Student encoder
// model.py:252-264):   out = [x +] BN(pwl(act(BN(dw_3x3(act(BN(pw(x))))))))
// One launch per block: the expanded tensor (3-6x the block's input) never reaches HBM; the block's
// traffic drops from  x + 2E + 2D + out  to  ~(1.3-2.2) x + out.
//
// Each WAVE owns a PH x PW patch of output pixels and runs the whole expand -> depthwise -> project chain
// on its own with no workgroup barrier (a workgroup is four independent waves; wave-level LDS ordering
// only), in sub-chunks of 16 expanded channels:
//   * the patch's input halo is held in REGISTERS as split-precision MFMA pixel fragments (hi/lo f16,
//     k_conv_gemm.h), loaded and split once;
//   * expand: 3 x v_mfma_f32_16x16x32_f16 per 32 input channels and 16x16 tile -> +bias, act, zero
//     outside the image (the depthwise conv zero-pads the EXPANDED map) -> E[halo][16] f32 in LDS;
//   * depthwise 3x3 (+bias, act) in f32 from LDS -> D[patch][32] f32 in LDS;
//   * every 32 expanded channels: project, D split on the fly, accumulators in registers;
//   * weights come straight from global memory (a few KB shared by all waves: L2/L1 hits); each set is
//     requested right after the previous one's last use, so its latency hides behind the next phase.
// MSPLIT = 4 (low-resolution blocks, too few patches to fill the chip): the four waves of a workgroup
// share ONE patch and take every fourth 32-channel chunk; their partial projections are summed in LDS.
#pragma once
#include "pf_common.h"

struct MbconvArgs {
    const float* in;      // [B][inH][inW][inLd]
    float* out;           // [B][outH][outW][outLd]
    const float* res;     // residual (same pixel grid as out) or nullptr
    const pf_half* w_exp; // [MidPad][KS][hi 32 | lo 32]  f16 of w * 2^s  (K = input channel, zero padded)
    const float* b_exp;   // [MidPad]
    const float* w_dw;    // [9][MidPad]
    const float* b_dw;    // [MidPad]
    const pf_half* w_pwl; // [CoutPad][MidPad/32][hi 32 | lo 32]
    const float* b_pwl;   // [CoutPad]
    const float* w_exp32; // exact-f32 variant: [Mid16][CP] / [CoutPad][Mid16] f32 (MidPad == Mid16 there)
    const float* w_pwl32;
    float scale_exp, scale_pwl;   // 2^-s of the two weight sets
    int B, inH, inW, Cin, inLd, outH, outW, Cout, outLd, resLd;
    int Mid16, MidPad, CoutPad, pad, act;   // Mid16: expanded channels rounded to 16; MidPad: to 32
    // ShuffleNetV2 unit (yolov5-face backbone, stride 1): expand act = `act`, depthwise act = act_dw (none),
    // output act = act_out; the result goes to every outCs-th channel of `out` (channel shuffle folded into the
    // store) and the pass-through half of the unit's input is copied to the channels in between.
    int act_dw, act_out, outCs;
    unsigned* range_slot;   // f32s range guard: max |v| (raw bits) over the block input and the depthwise output it splits, or nullptr
    const float* pass_src;  // [B][outH][outW][passLd], passC channels -> pass_dst channel c*outCs, or nullptr
    float* pass_dst;
    int passLd, passC;
};

__device__ __forceinline__ void pf_split8(const pf_f32x4& v0, const pf_f32x4& v1, pf_half8& hi, pf_half8& lo, unsigned& amax) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? v0[e & 3] : v1[e & 3];
        const pf_half hv = (pf_half)v;
        hi[e] = hv;
        lo[e] = pf_split_lo(v, hv);
        amax = pf_amax(amax, v);                 // range guard (pf_common.h)
    }
}

// S: stride, KS: input channels / 32 (rounded up), PH x PW: patch, MAXNT: output channels / 16 (rounded up)
template <int S, int KS, int PH, int PW, int MAXNT, int MSPLIT>
__global__ __launch_bounds__(256, MSPLIT > 1 ? 2 : 4) void mbconv_wave_kernel(MbconvArgs a) {
    unsigned amax = 0;                         // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    constexpr int PP = PH * PW, MPW = PP / 16;
    constexpr int HH = (PH - 1) * S + 3, HW = (PW - 1) * S + 3, HP = HH * HW;
    constexpr int MH = (HP + 15) / 16, HPP = MH * 16;
    constexpr int DS = 36;                     // D row stride: 16-lane fragment reads hit 64 distinct banks
    // E pixel stride (floats).  The depthwise reads are ds_read_b32 of lanes (channel dc, pixel phase dg): two phases share a
    // 32-lane bank group, S pixels apart, so S * EST must be 16 mod 32 for them to land on different bank halves: 16 at
    // stride 1; at stride 2 the natural 16 puts them on the SAME banks (SQ_LDS_BANK_CONFLICT = 62-64 % of the LDS-active
    // cycles of the stride-2 kernels in round 1's layout), 24 does not.
    constexpr int EST = S == 2 ? 24 : 16;
    constexpr int WORK = HPP * EST + PP * DS;
    constexpr int RED = MSPLIT > 1 ? MPW * MAXNT * 256 : 0;
    constexpr int WSZ = WORK > RED ? WORK : RED;
    static_assert(PP % 16 == 0 && (PW == 4 || PW == 8) && (MSPLIT == 1 || MSPLIT == 4), "patch shape");

    __shared__ __attribute__((aligned(16))) float smem[4][WSZ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* es = smem[wave];                    // [HPP][EST]  expanded sub-chunk
    float* ds = es + HPP * EST;                // [PP][DS]   depthwise output, 32 channels

    const int patchesX = (a.outW + PW - 1) / PW, patchesY = (a.outH + PH - 1) / PH;
    const int pid = MSPLIT > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (pid >= patchesX * patchesY) return;    // MSPLIT == 1: a whole wave leaves, nobody waits for it
    const int b = blockIdx.y;
    const int oy0 = (pid / patchesX) * PH, ox0 = (pid % patchesX) * PW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const float* in = a.in + (size_t)b * a.inH * a.inW * a.inLd;
    const int frow = lane & 15, kg = lane >> 4;
    const int fk = kg * 4;                     // accumulator layout: channels fk..fk+3 of pixel frow
    const int k8 = kg * 8;                     // operand layout: k elements k8..k8+7 of row frow

    // ---- input halo -> split MFMA pixel fragments -----------------------------------------------------
    pf_half8 xh[MH][KS], xl[MH][KS];
    unsigned inside = 0;
#pragma unroll
    for (int mt = 0; mt < MH; ++mt) {
        const int hp = mt * 16 + frow;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)a.inH && (unsigned)ix < (unsigned)a.inW;
        inside |= ok ? (1u << mt) : 0u;
        const float* px = in + ((size_t)(ok ? iy : 0) * a.inW + (ok ? ix : 0)) * a.inLd;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pf_f32x4 v0 = pf_f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (ok && ks * 32 + k8 < a.Cin) {  // Cin % 8 == 0
                v0 = *reinterpret_cast<const pf_f32x4*>(px + ks * 32 + k8);
                v1 = *reinterpret_cast<const pf_f32x4*>(px + ks * 32 + k8 + 4);
            }
            pf_split8(v0, v1, xh[mt][ks], xl[mt][ks], amax);
        }
    }
    pf_f32x4 acc[MPW][MAXNT];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int j = 0; j < MAXNT; ++j) acc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int NTC = a.CoutPad / 16;
    const int dc = lane & 15, dg = lane >> 4;  // depthwise role: channel dc, pixel phase dg
    const int step = 32 * MSPLIT;
    const int mc0 = MSPLIT > 1 ? wave * 32 : 0;

    // weight sets of the NEXT sub-chunk are fetched right after the current ones were consumed
    pf_half8 wh[KS], wl[KS];
    pf_f32x4 be;
    float wk[9], bd;
    // uniform (scalar) base + one 32-bit lane offset per stream: keeps the address arithmetic out of the VGPRs
    const unsigned woff = (unsigned)(frow * KS * 64 + k8);
    const unsigned poff = (unsigned)(frow * (a.MidPad / 32) * 64 + k8);
    auto fetch_expand = [&](int m) {
        if (m >= a.Mid16) return;
        const pf_half* base = a.w_exp + (size_t)m * (KS * 64);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wh[ks] = *reinterpret_cast<const pf_half8*>(base + woff + ks * 64);
            wl[ks] = *reinterpret_cast<const pf_half8*>(base + woff + ks * 64 + 32);
        }
        be = *reinterpret_cast<const pf_f32x4*>(a.b_exp + m + (unsigned)fk);
    };
    auto fetch_dw = [&](int m) {
        if (m >= a.Mid16) return;
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = (a.w_dw + (k * a.MidPad + m))[(unsigned)dc];
        bd = (a.b_dw + m)[(unsigned)dc];
    };
    fetch_expand(mc0);
    fetch_dw(mc0);

#pragma unroll 1
    for (int mc = mc0; mc < a.MidPad; mc += step) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m = mc + 16 * half;
            if (m >= a.Mid16) {                // padding sub-chunk (only ever the second half): D = 0
#pragma unroll
                for (int i = 0; i < PP / 4; ++i) ds[(dg + 4 * i) * DS + 16 + dc] = 0.f;
                pf_wave_sync();
                continue;
            }
            const int mnext = (half == 0 && m + 16 < a.Mid16) ? m + 16 : mc + step;
            // ---- expand -> E -----------------------------------------------------------------------------
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) {
                pf_f32x4 e = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    e = pf_mfma_16x16x32_f16(wl[ks], xh[mt][ks], e);
                    e = pf_mfma_16x16x32_f16(wh[ks], xl[mt][ks], e);
                    e = pf_mfma_16x16x32_f16(wh[ks], xh[mt][ks], e);
                }
                const bool ok = (inside >> mt) & 1u;
                pf_f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaf(e[r], a.scale_exp, be[r]);
                mb_act<4>(o, a.act);
                *reinterpret_cast<pf_f32x4*>(es + (mt * 16 + frow) * EST + fk) = ok ? o : pf_f32x4{0.f, 0.f, 0.f, 0.f};
            }
            fetch_expand(mnext);
            pf_wave_sync();
            // ---- depthwise 3x3 -> D ------------------------------------------------------------------------
            float dv[PP / 4];
#pragma unroll
            for (int i = 0; i < PP / 4; ++i) {
                const int px = PW == 4 ? dg : dg + 4 * (i & 1);
                const int py = PW == 4 ? i : i >> 1;
                float s = bd;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        s = fmaf(wk[ky * 3 + kx], es[((py * S + ky) * HW + px * S + kx) * EST + dc], s);
                dv[i] = s;
            }
            mb_act<PP / 4>(dv, a.act_dw);
#pragma unroll
            for (int i = 0; i < PP / 4; ++i) {
                const int px = PW == 4 ? dg : dg + 4 * (i & 1);
                const int py = PW == 4 ? i : i >> 1;
                ds[(py * PW + px) * DS + 16 * half + dc] = dv[i];
            }
            fetch_dw(mnext);
            pf_wave_sync();
        }
        // ---- project 32 expanded channels --------------------------------------------------------------------
        pf_half8 dh[MPW], dl[MPW];
#pragma unroll
        for (int mt = 0; mt < MPW; ++mt) {
            const float* dp = ds + (mt * 16 + frow) * DS + k8;
            pf_split8(*reinterpret_cast<const pf_f32x4*>(dp), *reinterpret_cast<const pf_f32x4*>(dp + 4), dh[mt], dl[mt], amax);
        }
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt)
            if (nt < NTC) {             // one output tile's weights at a time (8 tiles x hi/lo would be 64 registers)
                const pf_half* base = a.w_pwl + ((size_t)nt * 16 * (a.MidPad / 32) + mc / 32) * 64;
                const pf_half8 ph = *reinterpret_cast<const pf_half8*>(base + poff);
                const pf_half8 pl = *reinterpret_cast<const pf_half8*>(base + poff + 32);
#pragma unroll
                for (int mt = 0; mt < MPW; ++mt) {
                    acc[mt][nt] = pf_mfma_16x16x32_f16(pl, dh[mt], acc[mt][nt]);
                    acc[mt][nt] = pf_mfma_16x16x32_f16(ph, dl[mt], acc[mt][nt]);
                    acc[mt][nt] = pf_mfma_16x16x32_f16(ph, dh[mt], acc[mt][nt]);
                }
            }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    auto store_tile = [&](int mt, int nt, pf_f32x4 sum) {
        const int p = mt * 16 + frow;
        const int co = nt * 16 + fk;
        const int oy = oy0 + p / PW, ox = ox0 + p % PW;
        if (oy >= a.outH || ox >= a.outW || co >= a.Cout) return;
        const size_t pix = ((size_t)b * a.outH + oy) * a.outW + ox;
        pf_f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(sum[r], a.scale_pwl, a.b_pwl[co + r]);
        if (a.res) {
            const pf_f32x4 rv = *reinterpret_cast<const pf_f32x4*>(a.res + pix * a.resLd + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        mb_act<4>(v, a.act_out);
        if (a.outCs == 1) {
            *reinterpret_cast<pf_f32x4*>(a.out + pix * a.outLd + co) = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.out[pix * a.outLd + (size_t)(co + r) * a.outCs] = v[r];
        }
    };
    // pass-through half of a ShuffleV2 unit: this patch's pixels, channel c -> c * outCs of pass_dst
    auto copy_pass = [&](int part, int parts) {
        if (!a.pass_src) return;
        for (int q = part; q < MPW * (a.passC / 16); q += parts) {
            const int mt = q % MPW, cg = q / MPW;
            const int p = mt * 16 + frow;
            const int oy = oy0 + p / PW, ox = ox0 + p % PW;
            if (oy >= a.outH || ox >= a.outW) continue;
            const size_t pix = ((size_t)b * a.outH + oy) * a.outW + ox;
            const int c = cg * 16 + fk;
            const pf_f32x4 v = *reinterpret_cast<const pf_f32x4*>(a.pass_src + pix * a.passLd + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) a.pass_dst[pix * a.outLd + (size_t)(c + r) * a.outCs] = v[r];
        }
    };
    pf_amax_commit(a.range_slot, amax, amax_seen);
    if constexpr (MSPLIT == 1) {
#pragma unroll
        for (int mt = 0; mt < MPW; ++mt)
#pragma unroll
            for (int nt = 0; nt < MAXNT; ++nt)
                if (nt < NTC) store_tile(mt, nt, acc[mt][nt]);
        copy_pass(0, 1);
    } else {
        // the four waves hold partial sums of the same tiles: exchange through LDS, tile q summed by wave q % 4
        pf_wave_sync();
#pragma unroll
        for (int mt = 0; mt < MPW; ++mt)
#pragma unroll
            for (int nt = 0; nt < MAXNT; ++nt)
                *reinterpret_cast<pf_f32x4*>(smem[wave] + (mt * MAXNT + nt) * 256 + lane * 4) = acc[mt][nt];
        __syncthreads();
        for (int q = wave; q < MPW * NTC; q += 4) {
            const int mt = q % MPW, nt = q / MPW;
            pf_f32x4 sum = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += *reinterpret_cast<const pf_f32x4*>(smem[w] + (mt * MAXNT + nt) * 256 + lane * 4);
            store_tile(mt, nt, sum);
        }
        copy_pass(wave, 4);
    }
}

// ---- exact-f32 variant for the two high-resolution blocks (16 / 24 input channels, 24 outputs) ---------------
// Same wave-per-patch structure, v_mfma_f32_16x16x4_f32 on f32 fragments: with so few channels the split
// path's hi/lo fragments double the register footprint and spill (0.42 vs 0.26 ms on block 1.0); the block is
// latency/bandwidth bound, not matrix bound.  16 expanded channels per step.  Launch bound = 3 waves per SIMD:
// capped at 128 VGPRs (4 waves) the 24-channel variant spills 14 registers and runs 20 % slower.
// NOEXP: depthwise-separable block (timm DepthwiseSeparableConv, encoder block 0): no expand conv, the
// "expanded" map is the input itself (the halo fragments are stored to LDS as they are).
// SP (f32s programs, round 3): the same kernel with its matrix products on v_mfma_f32_16x16x16_f16, split precision.  The
// f32 variant turned out to be bound by the f32 matrix pipe itself -- v_mfma_f32_16x16x4_f32 takes 32 cycles, four of them
// per 16-deep K block: 12 k cycles of MFMA issue per patch and SIMD are exactly what block 1.0 measured -- whereas K = 16 is
// the native depth of the 16x16x16 f16 instruction (4 f16 per lane, the SAME lane / k mapping as four 16x16x4 f32 steps,
// and hi + lo fragments in the registers one f32 fragment took): 3 x 16 cycles instead of 4 x 32 per K block.  Weights are
// scaled by a power of two and split when fetched (a.scale_exp / a.scale_pwl = 2^-s undo the scaling), activations are split
// once per patch / per 16 depthwise channels.
// ACT >= 0: the block's activation at compile time (round 5).  With the run-time switch every 16-pixel tile of the expand loop sat
// between two scalar branches, so each tile's three dependent MFMAs ran to completion (s_nop 6) before its epilogue started and no
// tile overlapped the next; without branches the compiler interleaves them.  The Student's three blocks here are all ReLU.
template <int S, int CP, int PH, int PW, bool NOEXP = false, bool SP = false, int ACT = -1>
__global__ __launch_bounds__(256, 3) void mbconv_wave_f32_kernel(MbconvArgs a) {
    unsigned amax = 0;                         // range guard (pf_common.h): only the SP variant splits anything
    const unsigned amax_seen = SP ? pf_amax_seen(a.range_slot) : 0u;
    constexpr int PP = PH * PW, MPW = PP / 16;
    constexpr int HH = (PH - 1) * S + 3, HW = (PW - 1) * S + 3, HP = HH * HW;
    constexpr int MH = (HP + 15) / 16, HPP = MH * 16;
    constexpr int KK = CP / 16;
    constexpr int MAXNT = 2;                   // Cout <= 32
    constexpr int EST = S == 2 ? 24 : 16;      // E pixel stride, see mbconv_wave_kernel
    static_assert(PP % 16 == 0 && (PW == 4 || PW == 8), "patch shape");

    __shared__ __attribute__((aligned(16))) float smem[4][HPP * EST + PP * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* es = smem[wave];                    // [HPP][EST]
    float* ds = es + HPP * EST;                // [PP][16]

    const int patchesX = (a.outW + PW - 1) / PW, patchesY = (a.outH + PH - 1) / PH;
    const int pid = blockIdx.x * 4 + wave;
    if (pid >= patchesX * patchesY) return;
    const int b = blockIdx.y;
    const int oy0 = (pid / patchesX) * PH, ox0 = (pid % patchesX) * PW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const float* in = a.in + (size_t)b * a.inH * a.inW * a.inLd;
    const int frow = lane & 15, fk = (lane >> 4) * 4;
    auto split4 = [&](const pf_f32x4& v, pf_half4& hi, pf_half4& lo, bool track) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];
            const pf_half hv = (pf_half)x;
            hi[e] = hv;
            lo[e] = pf_split_lo(x, hv);
            if (track) amax = pf_amax(amax, x);
        }
    };

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
    // SP: the halo fragments as f16 hi / lo (they overlay the registers of xf, which is dead afterwards unless NOEXP)
    pf_half4 xh[SP && !NOEXP ? MH : 1][KK], xl[SP && !NOEXP ? MH : 1][KK];
    if constexpr (SP && !NOEXP) {
#pragma unroll
        for (int mt = 0; mt < MH; ++mt)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) split4(xf[mt][kk], xh[mt][kk], xl[mt][kk], true);
    }
    pf_f32x4 acc[MPW][MAXNT];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int j = 0; j < MAXNT; ++j) acc[i][j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int NTC = a.CoutPad / 16;
    const int dc = lane & 15, dg = lane >> 4;
    const unsigned woff = (unsigned)(frow * CP + fk);
    const unsigned poff = (unsigned)(frow * a.Mid16 + fk);

    // every weight set is requested right after the previous one's last use (latency hides behind the next phase)
    // SP: the 16 bytes of four f32 weights hold their scaled split instead -- [hi x 4 | lo x 4] f16, made at pack time (ir.py::mbconv:
    // same arithmetic as split4 below) -- so no wave splits weights (it was 66-88 of a chunk's ~330 VALU instructions, in every one of
    // the 65 536 waves of a launch, and these kernels are VALU-bound: profiles/r05_run37_pmc_all_kernels.txt)
    pf_f32x4 wv[KK], be, pv[MAXNT];
    pf_half8 wv8[SP ? KK : 1], pv8[SP ? MAXNT : 1];
    float wk[9], bd;
    auto fetch_expand = [&](int m) {
        if (m >= a.Mid16) return;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if constexpr (SP) wv8[kk] = *reinterpret_cast<const pf_half8*>(a.w_exp32 + (size_t)m * CP + woff + kk * 16);
            else wv[kk] = *reinterpret_cast<const pf_f32x4*>(a.w_exp32 + (size_t)m * CP + woff + kk * 16);
        }
        be = *reinterpret_cast<const pf_f32x4*>(a.b_exp + m + (unsigned)fk);
    };
    auto fetch_dw = [&](int m) {
        if (m >= a.Mid16) return;
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = (a.w_dw + (k * a.MidPad + m))[(unsigned)dc];
        bd = (a.b_dw + m)[(unsigned)dc];
    };
    auto fetch_project = [&](int m) {
        if (m >= a.Mid16) return;
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt)
            if (nt < NTC) {
                if constexpr (SP) pv8[nt] = *reinterpret_cast<const pf_half8*>(a.w_pwl32 + (size_t)nt * 16 * a.Mid16 + m + poff);
                else pv[nt] = *reinterpret_cast<const pf_f32x4*>(a.w_pwl32 + (size_t)nt * 16 * a.Mid16 + m + poff);
            }
    };
    if constexpr (!NOEXP) fetch_expand(0);
    fetch_dw(0);
    fetch_project(0);

#pragma unroll 1
    for (int mc = 0; mc < a.Mid16; mc += 16) {
#pragma unroll
        for (int mt = 0; mt < MH; ++mt) {
            if constexpr (NOEXP) {
                static_assert(!NOEXP || CP == 16, "depthwise-separable variant: 16 channels");
                *reinterpret_cast<pf_f32x4*>(es + (mt * 16 + frow) * EST + fk) = xf[mt][0];   // zero outside the image already
            } else {
                pf_f32x4 e = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (SP) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const pf_half4 wh = pf_half4{wv8[kk][0], wv8[kk][1], wv8[kk][2], wv8[kk][3]};
                        const pf_half4 wl = pf_half4{wv8[kk][4], wv8[kk][5], wv8[kk][6], wv8[kk][7]};
                        e = pf_mfma_16x16x16_f16(wl, xh[mt][kk], e);
                        e = pf_mfma_16x16x16_f16(wh, xl[mt][kk], e);
                        e = pf_mfma_16x16x16_f16(wh, xh[mt][kk], e);
                    }
                    e *= a.scale_exp;
                } else {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                        for (int j = 0; j < 4; ++j) e = pf_mfma_16x16x4_f32(wv[kk][j], xf[mt][kk][j], e);
                }
                pf_f32x4 o = e + be;
                if constexpr (ACT >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = pf_act_c<ACT>(o[r]);
                } else pf_act_rh<4>(o, a.act);
                *reinterpret_cast<pf_f32x4*>(es + (mt * 16 + frow) * EST + fk) = ((inside >> mt) & 1u) ? o : pf_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (!NOEXP) fetch_expand(mc + 16);
        pf_wave_sync();
        float dv[PP / 4];
#pragma unroll
        for (int i = 0; i < PP / 4; ++i) {
            const int px = PW == 4 ? dg : dg + 4 * (i & 1);
            const int py = PW == 4 ? i : i >> 1;
            float s = bd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    s = fmaf(wk[ky * 3 + kx], es[((py * S + ky) * HW + px * S + kx) * EST + dc], s);
            dv[i] = s;
        }
        if constexpr (ACT >= 0) {
#pragma unroll
            for (int i = 0; i < PP / 4; ++i) dv[i] = pf_act_c<ACT>(dv[i]);
        } else pf_act_rh<PP / 4>(dv, a.act);
#pragma unroll
        for (int i = 0; i < PP / 4; ++i) {
            const int px = PW == 4 ? dg : dg + 4 * (i & 1);
            const int py = PW == 4 ? i : i >> 1;
            ds[(py * PW + px) * 16 + dc] = dv[i];
        }
        fetch_dw(mc + 16);
        pf_wave_sync();
#pragma unroll
        for (int mt = 0; mt < MPW; ++mt) {
            const pf_f32x4 d4 = *reinterpret_cast<const pf_f32x4*>(ds + (mt * 16 + frow) * 16 + fk);
            if constexpr (SP) {
                pf_half4 dh, dl;
                split4(d4, dh, dl, true);
#pragma unroll
                for (int nt = 0; nt < MAXNT; ++nt)
                    if (nt < NTC) {
                        const pf_half4 ph = pf_half4{pv8[nt][0], pv8[nt][1], pv8[nt][2], pv8[nt][3]};
                        const pf_half4 pl = pf_half4{pv8[nt][4], pv8[nt][5], pv8[nt][6], pv8[nt][7]};
                        acc[mt][nt] = pf_mfma_16x16x16_f16(pl, dh, acc[mt][nt]);
                        acc[mt][nt] = pf_mfma_16x16x16_f16(ph, dl, acc[mt][nt]);
                        acc[mt][nt] = pf_mfma_16x16x16_f16(ph, dh, acc[mt][nt]);
                    }
            } else {
#pragma unroll
                for (int nt = 0; nt < MAXNT; ++nt)
                    if (nt < NTC) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[mt][nt] = pf_mfma_16x16x4_f32(pv[nt][j], d4[j], acc[mt][nt]);
                    }
            }
        }
        fetch_project(mc + 16);
    }
    if constexpr (SP) pf_amax_commit(a.range_slot, amax, amax_seen);
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
            for (int r = 0; r < 4; ++r) v[r] = (SP ? acc[mt][nt][r] * a.scale_pwl : acc[mt][nt][r]) + a.b_pwl[co + r];
            if (a.res) {
                const pf_f32x4 rv = *reinterpret_cast<const pf_f32x4*>(a.res + pix * a.resLd + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
            *reinterpret_cast<pf_f32x4*>(a.out + pix * a.outLd + co) = v;
        }
}
