// Workgroup-level fused kernels of the yolov5n-0.5 face detector (f32s programs), round 4.
//
// The detector (reference seam: Skps/core/api/face_detector.py:29-31, one ONNX session call per frame) is ~60 small
// layers on 96x160 ... 12x20 maps with 16 ... 256 channels: per lane-step of 32 frames it is 11 GFLOP -- nothing for the
// matrix cores -- and used to cost ~1.5 ms in ~55 launches, every one a chain of dependent phases executed by four-wave
// workgroups (the three-lane kernel trace, profiles/r04_run1_lane_trace_3lanes.md, charges those launches with a third of
// the summed kernel time under contention).  The kernels here run a whole ShuffleNetV2 unit / C3 block per launch with
// ONE pass over HBM and ONE short chain of workgroup-wide phases:
//
//   det_unit_kernel<C, K1, S>   ShuffleV2Block (models/common.py of yolov5-face; oracle/detector_net.py:_shuffle_block):
//        stride 1:  out = shuffle(cat(x1, silu(pw2(dw3x3(silu(pw1(x2)))))))
//        stride 2:  out = shuffle(cat(silu(pw(dw3x3s2(x))), silu(pw2(dw3x3s2(silu(pw1(x)))))))
//   det_c3_kernel<CIN>          C3 (n = 1, no shortcut) [+ one trailing 1x1 conv, + the Detect conv and its decode]
//
// Common structure: a workgroup owns a TH x TW tile of OUTPUT pixels of one frame.
//   phase 0  the input region (tile + the 3x3 halo) is read once from global memory, split into f16 hi / lo and parked in
//            LDS as MFMA pixel-operand planes ([32-channel chunk][hi | lo][row][64 B], the chunk rotation of k_conv_gemm.h);
//   GEMMs    every 1x1 conv is a split-precision MFMA GEMM (3 x v_mfma_f32_16x16x32_f16 per 32 k) whose pixel operand comes
//            from such planes and whose weight fragments (pre-split, a few KB) are read straight from L2 -- requested one
//            phase ahead, so their latency hides behind the phase in front of them;
//   dw 3x3   reads the expanded map from LDS as f32 (thread = 4 channels x one pixel), adds the bias, splits, parks planes;
//   store    the two halves of the channel shuffle are interleaved in registers: 32 contiguous bytes per lane.
// Zero padding of the EXPANDED map (the depthwise conv pads its own input) = region pixels outside the image are forced
// to zero after the first GEMM's bias / activation.
#pragma once
#include "k_conv_gemm.h"
#include "k_mbconv.h"

// LDS holds whatever the previous workgroup left there; the emulator's "LDS" is zero-initialised host memory, which hides reads of
// rows nobody wrote (round 4's first GPU run of the C3 kernel: NaN patterns in the padding rows of one plane reached the range
// guard).  The CPU tier therefore poisons the arrays of these kernels with 0xFF bytes (NaN as f16 and as f32) at kernel entry.
#ifdef PF_SIMT_EMULATION
#define PF_EMU_POISON(arr)                                        \
    do {                                                          \
        if (threadIdx.x == 0) memset((void*)(arr), 0xFF, sizeof(arr)); \
        __syncthreads();                                          \
    } while (0)
#else
#define PF_EMU_POISON(arr) do { } while (0)
#endif

// ---- pixel-operand planes in LDS ----------------------------------------------------------------------------------------
// rows = pixels (MR of them, a multiple of 16), 32 channels per chunk; chunk kc: hi plane at (2 kc) * MR * 64, lo plane behind it
__device__ __forceinline__ unsigned char* det_plane(unsigned char* base, int MR, int kc) { return base + (size_t)(2 * kc) * MR * 64; }

// 8 consecutive channels (c8 = channel / 8) of pixel row `row`
__device__ __forceinline__ void det_park8(unsigned char* base, int MR, int row, int c8, const pf_f32x4& v0, const pf_f32x4& v1, unsigned& amax) {
    pf_half8 hi, lo;
    pf_split8(v0, v1, hi, lo, amax);
    unsigned char* p = det_plane(base, MR, c8 >> 2) + pf_lds_chunk_off(row, c8 & 3);
    *reinterpret_cast<pf_half8*>(p) = hi;
    *reinterpret_cast<pf_half8*>(p + (size_t)MR * 64) = lo;
}

// 4 consecutive channels (c4 = channel / 4)
__device__ __forceinline__ void det_park4(unsigned char* base, int MR, int row, int c4, const pf_f32x4& v, unsigned& amax) {
    pf_half4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const pf_half hv = (pf_half)v[e];
        hi[e] = hv;
        lo[e] = pf_split_lo(v[e], hv);
        amax = pf_amax(amax, v[e]);
    }
    unsigned char* p = det_plane(base, MR, c4 >> 3) + pf_lds_chunk_off(row, (c4 >> 1) & 3) + (c4 & 1) * 8;
    *reinterpret_cast<pf_half4*>(p) = hi;
    *reinterpret_cast<pf_half4*>(p + (size_t)MR * 64) = lo;
}

__device__ __forceinline__ void det_frag(const unsigned char* base, int MR, int kc, int row, int g, pf_half8& h, pf_half8& l) {
    const unsigned char* p = base + (size_t)(2 * kc) * MR * 64 + pf_lds_chunk_off(row, g);
    h = *reinterpret_cast<const pf_half8*>(p);
    l = *reinterpret_cast<const pf_half8*>(p + (size_t)MR * 64);
}

// weight fragments of output-channel tile nt: rows [N][KS][hi 32 | lo 32] f16 (ir.py _split_rows)
template <int KS>
__device__ __forceinline__ void det_wfrag(const pf_half* w, int nt, int lane, pf_half8 (&wh)[KS], pf_half8 (&wl)[KS]) {
    const pf_half* p = w + ((size_t)(nt * 16 + (lane & 15)) * KS) * 64 + (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wh[ks] = *reinterpret_cast<const pf_half8*>(p + ks * 64);
        wl[ks] = *reinterpret_cast<const pf_half8*>(p + ks * 64 + 32);
    }
}

// one 16 x 16 output tile: D[n][px] over KS k-steps of planes `x` (pixel rows row0 .. row0 + 15)
template <int KS>
__device__ __forceinline__ pf_f32x4 det_tile(const unsigned char* x, int MR, int row0, int lane, const pf_half8 (&wh)[KS], const pf_half8 (&wl)[KS]) {
    pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int row = row0 + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        pf_half8 xh, xl;
        det_frag(x, MR, ks, row, g, xh, xl);
        acc = pf_mfma_16x16x32_f16(wl[ks], xh, acc);      // small terms first
        acc = pf_mfma_16x16x32_f16(wh[ks], xl, acc);
        acc = pf_mfma_16x16x32_f16(wh[ks], xh, acc);
    }
    return acc;
}

// SiLU with the hardware's exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of libm expf and an IEEE division:
// ~6 instructions instead of ~40 behind every accumulator register of these latency-bound kernels; the result differs from
// x / (1 + expf(-x)) by a few 1e-7 relative, three orders below the detector's parity bar.
__device__ __forceinline__ float det_silu(float x) {
#ifdef PF_SIMT_EMULATION
    return x / (1.f + expf(-x));
#else
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
#endif
}
__device__ __forceinline__ pf_f32x4 det_silu4(const pf_f32x4& acc, float scale, const pf_f32x4& b) {
    pf_f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = det_silu(fmaf(acc[r], scale, b[r]));
    return v;
}

// ---- ShuffleV2Block ---------------------------------------------------------------------------------------------------------
struct DetUnitArgs {
    const float* in;      // [B][inH][inW][inLd]; stride 1: channels [0, C) pass through, [C, 2C) feed branch 2; stride 2: Cin channels
    float* out;           // [B][outH][outW][outLd], 2C channels written (shuffled)
    const pf_half* w1;    // branch2 1x1 #1  [C][K1/32][64]   (BN folded, scaled by a power of two, split)
    const float* b1;      // [C]
    const float* wd;      // branch2 depthwise [9][C], bd [C]
    const float* bd;
    const pf_half* w2;    // branch2 1x1 #2  [C][C/32][64]
    const float* b2;
    const float* wd1;     // stride 2: branch1 depthwise [9][K1] (zero padded), bd1 [K1]
    const float* bd1;
    const pf_half* w3;    // stride 2: branch1 1x1  [C][K1/32][64]
    const float* b3;
    float s1, s2, s3;     // 2^-s of the three weight sets
    int B, inH, inW, inLd, Cin, outH, outW, outLd, TH, TW, tilesX, tpf;     // tpf: tiles per frame
    unsigned* range_slot;
    unsigned long long* prof;   // ablation build only (PEPPA_DBG & 4096): [5] cycles of phase 0 / GEMM 1 / depthwise / last GEMMs, workgroups
};

// WPS: waves per SIMD the register allocation must leave room for (workgroups per CU x NTHR / 256)
template <int C, int K1, int S, int MAXR, int NTHR, int WPS>
__global__ __launch_bounds__(NTHR, WPS) void det_unit_kernel(DetUnitArgs a) {
    constexpr int KS1 = K1 / 32, KSC = C / 32, ES = C + 4, NW = NTHR / 64, NTC = C / 16, MG = NW / NTC;
    constexpr int MAXD = S == 1 ? MAXR : ((MAXR / 4 + 15) / 16) * 16;
    constexpr int XB = KS1 * 2 * MAXR * 64, DB = KSC * 2 * MAXD * 64, D1B = S == 2 ? KS1 * 2 * MAXD * 64 : 16;
    constexpr int AB = XB > DB ? XB : DB;
    static_assert(NW % NTC == 0 && (C % 32) == 0 && (K1 % 32) == 0, "wave -> channel-tile assignment");
    __shared__ __attribute__((aligned(16))) unsigned char s_a[AB];     // X planes; later the branch-2 depthwise output's planes
    __shared__ __attribute__((aligned(16))) unsigned char s_d1[D1B];   // stride 2: planes of the branch-1 depthwise output
    __shared__ __attribute__((aligned(16))) float s_e[MAXR * ES];      // expanded map, f32, [region pixel][C + 4]
    __shared__ unsigned char s_in[MAXR];                               // region pixel inside the image?
    PF_EMU_POISON(s_a); PF_EMU_POISON(s_d1); PF_EMU_POISON(s_e); PF_EMU_POISON(s_in);

    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    const bool prof = PF_ABLATE != 0 && a.prof != nullptr;             // constant false in the production library
    const unsigned long long t0 = prof ? pf_clock() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned m_tw = pf_div_magic(a.TW);               // p / TW of the per-lane tile arithmetic (pf_common.h pf_div_small)
    // dispatch order (tiles of a frame on consecutive workgroups = on different XCDs).  Giving each XCD a contiguous run of tiles, so
    // that shared halo rows hit its own L2, measured 3-15 % SLOWER per launch (profiles/r04_run16_*): these launches are
    // latency-bound, not fabric-bound
    const int tile = blockIdx.x;
    const int b = tile / a.tpf, tt = tile - b * a.tpf;
    const int oy0 = (tt / a.tilesX) * a.TH, ox0 = (tt % a.tilesX) * a.TW;
    const int RW = (a.TW - 1) * S + 3, RH = (a.TH - 1) * S + 3, R = RH * RW, MR = (R + 15) & ~15;
    const int P = a.TH * a.TW, MRD = (P + 15) & ~15;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const float* in = a.in + (size_t)b * a.inH * a.inW * a.inLd;
    const int xoff = S == 1 ? C : 0;                                   // branch 2 reads the second half of a stride-1 unit's input
    const int nt = wave % NTC, mg = wave / NTC, g4 = (lane >> 4) * 4;

    // ---- phase 0: every global load of the launch's first half is issued before anything waits ----------------------------
    // (a load issued inside a loop that also parks its result costs the loop one full memory latency per iteration: ~5 k
    // cycles apiece when every workgroup of the launch starts at once -- the first cut of this kernel spent 10 k cycles here)
    constexpr int C8 = K1 / 8, IT0 = (MAXR * C8 + NTHR - 1) / NTHR;
    pf_f32x4 st[IT0][2];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int i = tid + it * NTHR;
        const int r = i / C8, c8 = i - r * C8;
        const int ry = r / RW, rx = r - ry * RW;
        const int iy = iy0 + ry, ix = ix0 + rx;
        const bool ok = r < R && (unsigned)iy < (unsigned)a.inH && (unsigned)ix < (unsigned)a.inW;
        st[it][0] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        st[it][1] = st[it][0];
        if (ok && 8 * c8 < a.Cin) {
            const float* px = in + ((size_t)iy * a.inW + ix) * a.inLd + xoff + 8 * c8;
            st[it][0] = *reinterpret_cast<const pf_f32x4*>(px);
            st[it][1] = *reinterpret_cast<const pf_f32x4*>(px + 4);
        }
    }
    pf_half8 w1h[KS1], w1l[KS1];
    det_wfrag<KS1>(a.w1, nt, lane, w1h, w1l);
    const pf_f32x4 b1v = *reinterpret_cast<const pf_f32x4*>(a.b1 + nt * 16 + g4);
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int i = tid + it * NTHR;
        if (i < MR * C8) {
            const int r = i / C8, c8 = i - r * C8;
            const int ry = r / RW, rx = r - ry * RW;
            det_park8(s_a, MR, r, c8, st[it][0], st[it][1], amax);
            if (c8 == 0) s_in[r] = (r < R && (unsigned)(iy0 + ry) < (unsigned)a.inH && (unsigned)(ix0 + rx) < (unsigned)a.inW) ? 1 : 0;
        }
    }
    // the pass-through half of this wave's output tiles: in flight across GEMM 1 and the depthwise phase
    constexpr int C4 = C / 4;
    const int dc4 = tid % C4;
    constexpr int MAXT = (MAXD / 16 + MG - 1) / MG;
    pf_f32x4 evn[S == 1 ? MAXT : 1];
    if constexpr (S == 1) {
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            const int p = (mg + j * MG) * 16 + (lane & 15);
            const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
            const int oy = oy0 + py, ox = ox0 + px;
            evn[j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < P && oy < a.outH && ox < a.outW) evn[j] = *reinterpret_cast<const pf_f32x4*>(in + ((size_t)oy * a.inW + ox) * a.inLd + nt * 16 + g4);
        }
    }
    __syncthreads();
    const unsigned long long t1 = prof ? pf_clock() : 0;

    // ---- GEMM 1: E = silu(W1 x + b1) on every region pixel, zero outside the image; two tiles per step ----------------------
    for (int mt = mg; mt < MR / 16; mt += 2 * MG) {
        const bool two = mt + MG < MR / 16;
        const pf_f32x4 acc0 = det_tile<KS1>(s_a, MR, mt * 16, lane, w1h, w1l);
        pf_f32x4 acc1 = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        if (two) acc1 = det_tile<KS1>(s_a, MR, (mt + MG) * 16, lane, w1h, w1l);
        const int r0 = mt * 16 + (lane & 15), r1 = r0 + MG * 16;
        pf_f32x4 v0 = det_silu4(acc0, a.s1, b1v);
        if (!s_in[r0]) v0 = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<pf_f32x4*>(s_e + r0 * ES + nt * 16 + g4) = v0;
        if (two) {
            pf_f32x4 v1 = det_silu4(acc1, a.s1, b1v);
            if (!s_in[r1]) v1 = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<pf_f32x4*>(s_e + r1 * ES + nt * 16 + g4) = v1;
        }
    }
    if constexpr (S == 2) {
        // branch 1: depthwise 3x3 stride 2 on the block input, read back from the X planes as hi + lo (22 significand bits: what
        // every split-precision conv sees of its input) -> planes of their own (the X planes are still being read by GEMM 1)
        constexpr int K4 = K1 / 4;
        const int c4 = tid % K4;
        pf_f32x4 w9[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w9[k] = *reinterpret_cast<const pf_f32x4*>(a.wd1 + k * K1 + 4 * c4);
        const pf_f32x4 bb = *reinterpret_cast<const pf_f32x4*>(a.bd1 + 4 * c4);
        const unsigned char* xp = det_plane(s_a, MR, c4 >> 3);
        const int xo = (c4 & 1) * 8, xs = (c4 >> 1) & 3;
        for (int p = tid / K4; p < MRD; p += NTHR / K4) {
            pf_f32x4 sum = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < P && 4 * c4 < a.Cin) {
                sum = bb;
                const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const unsigned char* q = xp + pf_lds_chunk_off((py * 2 + ky) * RW + px * 2 + kx, xs) + xo;
                        const pf_half4 h = *reinterpret_cast<const pf_half4*>(q), l = *reinterpret_cast<const pf_half4*>(q + (size_t)MR * 64);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum[r] = fmaf(w9[ky * 3 + kx][r], (float)h[r] + (float)l[r], sum[r]);
                    }
            }
            det_park4(s_d1, MRD, p, c4, sum, amax);
        }
    }
    // depthwise weights of branch 2 (in flight across the barrier) and the weights of the last GEMM(s) (across the depthwise phase)
    pf_f32x4 wdv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wdv[k] = *reinterpret_cast<const pf_f32x4*>(a.wd + k * C + 4 * dc4);
    const pf_f32x4 bdv = *reinterpret_cast<const pf_f32x4*>(a.bd + 4 * dc4);
    pf_half8 w2h[KSC], w2l[KSC];
    det_wfrag<KSC>(a.w2, nt, lane, w2h, w2l);
    const pf_f32x4 b2v = *reinterpret_cast<const pf_f32x4*>(a.b2 + nt * 16 + g4);
    pf_half8 w3h[S == 2 ? KS1 : 1], w3l[S == 2 ? KS1 : 1];
    pf_f32x4 b3v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (S == 2) {
        det_wfrag<KS1>(a.w3, nt, lane, w3h, w3l);
        b3v = *reinterpret_cast<const pf_f32x4*>(a.b3 + nt * 16 + g4);
    }
    __syncthreads();
    const unsigned long long t2 = prof ? pf_clock() : 0;

    // ---- depthwise 3x3 (stride S) on E -> D planes (over the X planes, which nobody reads any more) ----------------------
    unsigned char* dplanes = s_a;
    for (int p = tid / C4; p < MRD; p += NTHR / C4) {
        pf_f32x4 sum = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        if (p < P) {
            sum = bdv;
            const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
            const float* e0 = s_e + ((py * S) * RW + px * S) * ES + 4 * dc4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const pf_f32x4 ev = *reinterpret_cast<const pf_f32x4*>(e0 + (ky * RW + kx) * ES);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[r] = fmaf(wdv[ky * 3 + kx][r], ev[r], sum[r]);
                }
        }
        det_park4(dplanes, MRD, p, dc4, sum, amax);
    }
    __syncthreads();
    const unsigned long long t3 = prof ? pf_clock() : 0;

    // ---- last GEMM(s) + channel shuffle + store ------------------------------------------------------------------------------
    float* out = a.out + (size_t)b * a.outH * a.outW * a.outLd;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        const int mt = mg + j * MG;
        if (mt >= MRD / 16) break;
        const int p = mt * 16 + (lane & 15);
        const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
        const int oy = oy0 + py, ox = ox0 + px;
        const bool ok = p < P && oy < a.outH && ox < a.outW;
        const int n = nt * 16 + g4;
        const pf_f32x4 acc2 = det_tile<KSC>(dplanes, MRD, mt * 16, lane, w2h, w2l);
        pf_f32x4 even;
        if constexpr (S == 2) even = det_silu4(det_tile<KS1>(s_d1, MRD, mt * 16, lane, w3h, w3l), a.s3, b3v);
        else even = evn[j];
        const pf_f32x4 odd = det_silu4(acc2, a.s2, b2v);
        if (ok) {
            float* o = out + ((size_t)oy * a.outW + ox) * a.outLd + 2 * n;
            *reinterpret_cast<pf_f32x4*>(o) = pf_f32x4{even[0], odd[0], even[1], odd[1]};
            *reinterpret_cast<pf_f32x4*>(o + 4) = pf_f32x4{even[2], odd[2], even[3], odd[3]};
        }
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
    if (prof && tid == 0) {
        const unsigned long long t4 = pf_clock();
        atomicAdd(a.prof + 0, t1 - t0); atomicAdd(a.prof + 1, t2 - t1); atomicAdd(a.prof + 2, t3 - t2); atomicAdd(a.prof + 3, t4 - t3);
        atomicAdd(a.prof + 4, 1ull);
    }
}

// ---- C3 (n = 1, shortcut = False) [+ trailing 1x1 conv | + Detect conv and decode] -----------------------------------------------
// yolov5-face models/common.py C3 / models/yolo.py Detect, restated in oracle/detector_net.py::_c3 / detect_decode:
//   y1 = silu(cv1 x); y1 = silu(m.cv2_3x3(silu(m.cv1 y1))); y2 = silu(cv2 x); out = silu(cv3 cat(y1, y2))
// The block's input is the concatenation of up to two tensors, the first optionally a nearest x2 upsample (the PAN head's
// Upsample + Concat, never materialised): channels [0, CA) come from srcA at (y >> upA, x >> upA), [CA, CIN) from srcB.
//   TAIL 1: out2 = silu(W_e out + b_e)                       (model.11 behind model.10; the C3's own output is optional)
//   TAIL 2: raw = W_e out + b_e (48 = 3 anchors x 16), decoded in the epilogue into the (15120, 16) rows of
//           face_detector.py:31 -- one accumulator tile IS one anchor, a lane owns 4 of a row's 16 columns.
struct DetC3Args {
    const float* srcA; const float* srcB;   // [B][H >> upA][W >> upA][ldA], [B][H][W][ldB]
    float* out;                             // C3 output [B][H][W][outLd] (64 channels) or nullptr
    float* out2;                            // TAIL 1: [B][H][W][out2Ld] 64 channels; TAIL 2: raw Detect conv output (48 channels) or nullptr
    float* rows;                            // TAIL 2: [B][nrows_total][16]
    const pf_half* wA; const float* bA;     // [cv1 ; cv2]  [64][CIN/32][64]
    const pf_half* wB; const float* bB;     // m.cv1        [32][1][64]
    const pf_half* wC; const float* bC;     // m.cv2 3x3    [32][9][1][64]
    const pf_half* wD; const float* bD;     // cv3          [64][2][64]
    const pf_half* wE; const float* bE;     // tail         [64 | 48][2][64]
    const float* anchors;                   // TAIL 2: [3][2]
    float sA, sB, sC, sD, sE, det_stride;
    int B, H, W, CA, upA, ldA, ldB, outLd, out2Ld, TH, TW, tilesX, tpf, row0, nrows_total;
    unsigned* range_slot;
};

template <int CIN, int TAIL, int MAXR, int NTHR>
__global__ __launch_bounds__(NTHR) void det_c3_kernel(DetC3Args a) {
    constexpr int KSA = CIN / 32, NW = NTHR / 64;
    constexpr int XB = KSA * 2 * MAXR * 64;          // X planes; later Z (1 chunk, region rows) + OUT (2 chunks, tile rows)
    constexpr int YB = 2 * MAXR * 64;                // y1 planes (1 chunk, region rows)
    constexpr int CB = 2 * 2 * MAXR * 64;            // cat(y1', y2) planes (2 chunks, tile rows <= MAXR)
    static_assert(NW % 4 == 0 && XB >= 2 * MAXR * 64 + 2 * 2 * MAXR * 64, "LDS plan");
    __shared__ __attribute__((aligned(16))) unsigned char s_x[XB];
    __shared__ __attribute__((aligned(16))) unsigned char s_y[YB];
    __shared__ __attribute__((aligned(16))) unsigned char s_c[CB];
    __shared__ unsigned char s_in[MAXR];
    PF_EMU_POISON(s_x); PF_EMU_POISON(s_y); PF_EMU_POISON(s_c); PF_EMU_POISON(s_in);

    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned m_tw = pf_div_magic(a.TW);               // p / TW of the per-lane tile arithmetic (pf_common.h pf_div_small)
    const int tile = blockIdx.x;                                     // dispatch order, see det_unit_kernel
    const int b = tile / a.tpf, tt = tile - b * a.tpf;
    const int oy0 = (tt / a.tilesX) * a.TH, ox0 = (tt % a.tilesX) * a.TW;
    const int RW = a.TW + 2, R = (a.TH + 2) * RW, MR = (R + 15) & ~15;
    const int P = a.TH * a.TW, MRD = (P + 15) & ~15;
    const int g4 = (lane >> 4) * 4;
    unsigned char* s_z = s_x;                                        // after GEMM A
    unsigned char* s_o = s_x + (size_t)2 * MR * 64;                  // after GEMM A, behind Z

    // weights of the first GEMM: requested before the input
    const int ntA = wave & 3;
    pf_half8 wAh[KSA], wAl[KSA];
    det_wfrag<KSA>(a.wA, ntA, lane, wAh, wAl);
    const pf_f32x4 bAv = *reinterpret_cast<const pf_f32x4*>(a.bA + ntA * 16 + g4);

    // ---- phase 0: concatenated input region -> split planes (all loads issued before the first one is waited for) ---------------
    {
        const int hA = a.H >> a.upA, wA_ = a.W >> a.upA;
        const float* srcA = a.srcA + (size_t)b * hA * wA_ * a.ldA;
        const float* srcB = a.srcB ? a.srcB + (size_t)b * a.H * a.W * a.ldB : nullptr;
        constexpr int C8 = CIN / 8, IT0 = (MAXR * C8 + NTHR - 1) / NTHR;
        pf_f32x4 st[IT0][2];
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int i = tid + it * NTHR;
            const int r = i / C8, c8 = i - r * C8;
            const int ry = r / RW, rx = r - ry * RW;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            st[it][0] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            st[it][1] = st[it][0];
            if (r < R && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                const float* px = 8 * c8 < a.CA ? srcA + ((size_t)(iy >> a.upA) * wA_ + (ix >> a.upA)) * a.ldA + 8 * c8
                                                : srcB + ((size_t)iy * a.W + ix) * a.ldB + (8 * c8 - a.CA);
                st[it][0] = *reinterpret_cast<const pf_f32x4*>(px);
                st[it][1] = *reinterpret_cast<const pf_f32x4*>(px + 4);
            }
        }
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int i = tid + it * NTHR;
            if (i < MR * C8) {
                const int r = i / C8, c8 = i - r * C8;
                const int ry = r / RW, rx = r - ry * RW;
                det_park8(s_x, MR, r, c8, st[it][0], st[it][1], amax);
                if (c8 == 0) s_in[r] = (r < R && (unsigned)(oy0 - 1 + ry) < (unsigned)a.H && (unsigned)(ox0 - 1 + rx) < (unsigned)a.W) ? 1 : 0;
            }
        }
    }
    // rows [P, MRD) of the y2 half of the cat planes are pixel padding no epilogue writes: zero them (LDS is not)
    for (int i = tid; i < (MRD - P) * 4; i += NTHR) {
        unsigned char* q = det_plane(s_c, MRD, 1) + pf_lds_chunk_off(P + (i >> 2), i & 3);
        const pf_half8 z = pf_half8{(pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0};
        *reinterpret_cast<pf_half8*>(q) = z;
        *reinterpret_cast<pf_half8*>(q + (size_t)MRD * 64) = z;
    }
    // weights of GEMM B and of the 3x3 conv: in flight across GEMM A
    const int ntB = wave & 1;
    pf_half8 wBh[1], wBl[1];
    det_wfrag<1>(a.wB, ntB, lane, wBh, wBl);
    const pf_f32x4 bBv = *reinterpret_cast<const pf_f32x4*>(a.bB + ntB * 16 + g4);
    __syncthreads();

    // ---- GEMM A: [y1 ; y2] = silu([cv1 ; cv2] x) on the region ------------------------------------------------------------------
    for (int mt = wave >> 2; mt < MR / 16; mt += NW / 4) {
        const pf_f32x4 acc = det_tile<KSA>(s_x, MR, mt * 16, lane, wAh, wAl);
        const pf_f32x4 v = det_silu4(acc, a.sA, bAv);
        const int r = mt * 16 + (lane & 15);
        if (ntA < 2) {
            det_park4(s_y, MR, r, ntA * 4 + (lane >> 4), v, amax);
        } else {
            const int ry = r / RW, rx = r - ry * RW;
            if (r < R && ry >= 1 && ry <= a.TH && rx >= 1 && rx <= a.TW)
                det_park4(s_c, MRD, (ry - 1) * a.TW + (rx - 1), 8 + (ntA - 2) * 4 + (lane >> 4), v, amax);
        }
    }
    pf_half8 wCh[9], wCl[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const pf_half* p = a.wC + ((size_t)(ntB * 16 + (lane & 15)) * 9 + t) * 64 + (lane >> 4) * 8;
        wCh[t] = *reinterpret_cast<const pf_half8*>(p);
        wCl[t] = *reinterpret_cast<const pf_half8*>(p + 32);
    }
    const pf_f32x4 bCv = *reinterpret_cast<const pf_f32x4*>(a.bC + ntB * 16 + g4);
    __syncthreads();

    // ---- GEMM B: z = silu(m.cv1 y1) on the region, zero outside the image (the 3x3 conv pads ITS input) -------------------------
    for (int mt = wave >> 1; mt < MR / 16; mt += NW / 2) {
        const pf_f32x4 acc = det_tile<1>(s_y, MR, mt * 16, lane, wBh, wBl);
        pf_f32x4 v = det_silu4(acc, a.sB, bBv);
        const int r = mt * 16 + (lane & 15);
        if (!s_in[r]) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        det_park4(s_z, MR, r, ntB * 4 + (lane >> 4), v, amax);
    }
    const int ntD = wave & 3;
    pf_half8 wDh[2], wDl[2];
    det_wfrag<2>(a.wD, ntD, lane, wDh, wDl);
    const pf_f32x4 bDv = *reinterpret_cast<const pf_f32x4*>(a.bD + ntD * 16 + g4);
    __syncthreads();

    // ---- GEMM C: y1' = silu(m.cv2 (3x3, pad 1) z) on the tile ----------------------------------------------------------------------
    for (int mt = wave >> 1; mt < MRD / 16; mt += NW / 2) {
        const int p = mt * 16 + (lane & 15);
        const int pc = p < P ? p : 0;
        const int py = pf_div_small(pc, m_tw), px = pc - py * a.TW;
        pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            pf_half8 zh, zl;
            det_frag(s_z, MR, 0, (py + t / 3) * RW + px + t % 3, lane >> 4, zh, zl);
            acc = pf_mfma_16x16x32_f16(wCl[t], zh, acc);
            acc = pf_mfma_16x16x32_f16(wCh[t], zl, acc);
            acc = pf_mfma_16x16x32_f16(wCh[t], zh, acc);
        }
        det_park4(s_c, MRD, p, ntB * 4 + (lane >> 4), det_silu4(acc, a.sC, bCv), amax);
    }
    __syncthreads();

    // ---- GEMM D: out = silu(cv3 cat) ---------------------------------------------------------------------------------------------------
    float* outp = a.out ? a.out + (size_t)b * a.H * a.W * a.outLd : nullptr;
    for (int mt = wave >> 2; mt < MRD / 16; mt += NW / 4) {
        pf_f32x4 v = det_silu4(det_tile<2>(s_c, MRD, mt * 16, lane, wDh, wDl), a.sD, bDv);
        const int p = mt * 16 + (lane & 15);
        if (p >= P) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};          // pixel padding of the last tile
        const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
        const int oy = oy0 + py, ox = ox0 + px;
        if (outp && p < P && oy < a.H && ox < a.W) *reinterpret_cast<pf_f32x4*>(outp + ((size_t)oy * a.W + ox) * a.outLd + ntD * 16 + g4) = v;
        if constexpr (TAIL != 0) det_park4(s_o, MRD, p, ntD * 4 + (lane >> 4), v, amax);
    }
    if constexpr (TAIL != 0) {
        constexpr int NTE = TAIL == 1 ? 4 : 3;
        const int ntE = wave % NTE;
        const bool works = wave < (NW / NTE) * NTE;
        pf_half8 wEh[2], wEl[2];
        det_wfrag<2>(a.wE, ntE, lane, wEh, wEl);
        const pf_f32x4 bEv = *reinterpret_cast<const pf_f32x4*>(a.bE + ntE * 16 + g4);
        __syncthreads();
        if (works) {
            for (int mt = wave / NTE; mt < MRD / 16; mt += NW / NTE) {
                const pf_f32x4 acc = det_tile<2>(s_o, MRD, mt * 16, lane, wEh, wEl);
                const int p = mt * 16 + (lane & 15);
                const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
                const int oy = oy0 + py, ox = ox0 + px;
                if (!(p < P && oy < a.H && ox < a.W)) continue;
                if constexpr (TAIL == 1) {
                    *reinterpret_cast<pf_f32x4*>(a.out2 + (((size_t)b * a.H + oy) * a.W + ox) * a.out2Ld + ntE * 16 + g4) = det_silu4(acc, a.sE, bEv);
                } else {
                    pf_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[r], a.sE, bEv[r]);
                    if (a.out2) *reinterpret_cast<pf_f32x4*>(a.out2 + (((size_t)b * a.H + oy) * a.W + ox) * a.out2Ld + ntE * 16 + g4) = v;
                    // Detect decode (k_layers.h detect_decode_kernel, same arithmetic): this lane holds columns g4 .. g4 + 3 of anchor ntE
                    const float aw = a.anchors[ntE * 2], ah = a.anchors[ntE * 2 + 1];
                    const float gx = (float)ox, gy = (float)oy;
                    pf_f32x4 o;
                    const int q = lane >> 4;
                    if (q == 0) {
                        const float s0 = pf_act(v[0], PF_ACT_SIGMOID), s1 = pf_act(v[1], PF_ACT_SIGMOID);
                        const float s2 = pf_act(v[2], PF_ACT_SIGMOID), s3 = pf_act(v[3], PF_ACT_SIGMOID);
                        o = pf_f32x4{(s0 * 2.f - 0.5f + gx) * a.det_stride, (s1 * 2.f - 0.5f + gy) * a.det_stride,
                                     (s2 * 2.f) * (s2 * 2.f) * aw, (s3 * 2.f) * (s3 * 2.f) * ah};
                    } else if (q == 1) {
                        o = pf_f32x4{pf_act(v[0], PF_ACT_SIGMOID), v[1] * aw + gx * a.det_stride, v[2] * ah + gy * a.det_stride, v[3] * aw + gx * a.det_stride};
                    } else if (q == 2) {
                        o = pf_f32x4{v[0] * ah + gy * a.det_stride, v[1] * aw + gx * a.det_stride, v[2] * ah + gy * a.det_stride, v[3] * aw + gx * a.det_stride};
                    } else {
                        o = pf_f32x4{v[0] * ah + gy * a.det_stride, v[1] * aw + gx * a.det_stride, v[2] * ah + gy * a.det_stride, pf_act(v[3], PF_ACT_SIGMOID)};
                    }
                    float* row = a.rows + ((size_t)b * a.nrows_total + a.row0 + ((size_t)ntE * a.H + oy) * a.W + ox) * 16;
                    *reinterpret_cast<pf_f32x4*>(row + g4) = o;
                }
            }
        }
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}

// ---- StemBlock ---------------------------------------------------------------------------------------------------------------------
// yolov5-face models/common.py StemBlock (oracle/detector_net.py::detector_features):
//   s1 = silu(conv3x3 s2 (x));  s2 = silu(conv3x3 s2 (silu(conv1x1 16->8 (s1))));  out = silu(conv1x1 (cat(s2, maxpool2x2(s1))))
// Five launches used to move the 192 x 320 x 16 stem map (3.9 MB per frame, f32) to HBM and back three times; here a workgroup
// owns a TH x TW tile of the 96 x 160 OUTPUT: the (4 TH + 3) x (4 TW + 3) input pixels it needs are staged once as f16 hi / lo
// (a uint8 pixel is exact in f16: lo = 0), every conv is a split-precision MFMA GEMM on LDS-resident operands
//   stem_1 : K = 27 (tap, colour) padded to 32, pixel fragments gathered from the staged image (im2col in registers)
//   stem_2a: K = 16, N = 8               stem_2b: K = 9 taps x 8 channels = 72 padded to 96      stem_3: K = 32
// and only the 16-channel result leaves the CU.  The max-pool picks the hi / lo PAIR of the largest hi + lo.
struct DetStemArgs {
    const void* in;           // u8 [B][H][W][3] (1/255 folded into w1_u8) or f32 [B][3][H][W]
    float* out;               // [B][OH][OW][outLd], 16 channels
    const pf_half* w1_u8; const pf_half* w1_f32; const float* b1;      // [16][1][64]
    const pf_half* w2a; const float* b2a;                             // [16 (8 used)][1][64]
    const pf_half* w2b; const float* b2b;                             // [16][3][64], k = tap * 8 + c
    const pf_half* w3; const float* b3;                               // [16][1][64], k = [stem_2b 16 | pool 16]
    float s1_u8, s1_f32, s2a, s2b, s3;
    int in_f32_nchw, B, H, W, SH, SW, OH, OW, outLd, TH, TW, tilesX;
    unsigned* range_slot;
};

// MAXO: largest TH * TW; the stem_1 region of such a tile has at most (2 TH + 1)(2 TW + 1) <= MAXS pixels; the image region is
// IRH = 4 TH + 3 <= MAXIH rows of IRW = 4 TW + 3 pixels.  Image rows live in LDS as f16 (a uint8 is exact: no lo plane on that
// path), RS halves apart, laid out so that a stem_1 pixel's nine (kx, colour) bytes of one image row are 9 consecutive halves
// starting on a 4-byte boundary: the K axis is ordered (ky, kx * 3 + ci) -- k groups 0..2 = the first 8 of the 9 halves of
// rows ky = 0..2 (four aligned ds_read_b32 each), group 3 = the ninth half of the three rows -- and the image reaches LDS as
// whole 32-bit words (W % 4 == 0: a word of an image row is inside or outside the image as a whole).  The first cut gathered
// (tap, colour) elements one ds_read_u16 at a time from bytes it had loaded one at a time behind two integer divisions each:
// 190 us per 32 frames, instruction-issue bound (11 us of HBM traffic).
template <int MAXO, int MAXS, int MAXIH, int RS, bool F32IN, int NTHR>
__global__ __launch_bounds__(NTHR) void det_stem_kernel(DetStemArgs a) {
    constexpr int NW = NTHR / 64;
    __shared__ __attribute__((aligned(16))) pf_half s_ih[MAXIH * RS];          // staged image rows (hi)
    __shared__ __attribute__((aligned(16))) pf_half s_il[F32IN ? MAXIH * RS : 8];   // lo halves, float input only
    __shared__ __attribute__((aligned(16))) unsigned char s_p1[MAXS * 64];     // stem_1: [row][hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] (slots rotated)
    __shared__ __attribute__((aligned(16))) unsigned char s_p2[MAXS * 32];     // stem_2a: [row][hi 0-7 | lo 0-7]
    __shared__ __attribute__((aligned(16))) unsigned char s_cat[2 * MAXO * 64];// cat(stem_2b, pool): one 32-channel chunk, hi / lo planes
    __shared__ unsigned char s_ok[MAXS];                                       // stem_1 region pixel inside the stem_1 map?
    PF_EMU_POISON(s_ih); PF_EMU_POISON(s_il); PF_EMU_POISON(s_p1); PF_EMU_POISON(s_p2); PF_EMU_POISON(s_cat); PF_EMU_POISON(s_ok);

    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned m_tw = pf_div_magic(a.TW);               // p / TW of the per-lane tile arithmetic (pf_common.h pf_div_small)
    const int SRW = 2 * a.TW + 1, SRH = 2 * a.TH + 1, S1R = SRH * SRW, MR1 = (S1R + 15) & ~15;
    const int IRW = 2 * SRW + 1, IRH = 2 * SRH + 1;
    const int P = a.TH * a.TW, MRD = (P + 15) & ~15;
    const int frow = lane & 15, g = lane >> 4, g4 = g * 4;
    // PERSISTENT workgroups (round 4, second cut): a workgroup walks tiles t = blockIdx.x, + gridDim.x, ... of all frames and
    // requests the NEXT tile's image words before it computes the current one -- with one tile per workgroup every workgroup
    // opened with a ~5 us wait for its 4 KB of pixels (133 us per 32 frames at three workgroups per CU).
    const int tiles_y = (a.OH + a.TH - 1) / a.TH, tpf = a.tilesX * tiles_y, ntiles = tpf * a.B;
    const int nwd = (IRW * 3 + 3 + 3) / 4;                    // aligned words covering one region row (the row starts at byte 3 of a word)
    const unsigned m_nwd = pf_div_magic(nwd);
    const int rowb = a.W * 3;
    constexpr int ITW = (MAXIH * (RS / 4) + NTHR - 1) / NTHR;
    unsigned wv[ITW];
    auto request = [&](int t) {                               // uint8 input: the image words of tile t -> registers
        const int tb = t / tpf, tt = t - tb * tpf;
        const int iy0 = 4 * (tt / a.tilesX) * a.TH - 3, wb = 3 * (4 * (tt % a.tilesX) * a.TW - 3) - 3;
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

    // weights: all four sets are a few hundred bytes per lane -- requested up front
    pf_half8 w1h[1], w1l[1], w2ah[1], w2al[1], w2bh[3], w2bl[3], w3h[1], w3l[1];
    det_wfrag<1>(F32IN ? a.w1_f32 : a.w1_u8, 0, lane, w1h, w1l);
    det_wfrag<1>(a.w2a, 0, lane, w2ah, w2al);
    det_wfrag<3>(a.w2b, 0, lane, w2bh, w2bl);
    det_wfrag<1>(a.w3, 0, lane, w3h, w3l);
    const pf_f32x4 b1v = *reinterpret_cast<const pf_f32x4*>(a.b1 + g4), b2av = *reinterpret_cast<const pf_f32x4*>(a.b2a + g4);
    const pf_f32x4 b2bv = *reinterpret_cast<const pf_f32x4*>(a.b2b + g4), b3v = *reinterpret_cast<const pf_f32x4*>(a.b3 + g4);
    const float s1 = F32IN ? a.s1_f32 : a.s1_u8;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / tpf, tt = t - b * tpf;
    const int oy0 = (tt / a.tilesX) * a.TH, ox0 = (tt % a.tilesX) * a.TW;
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;           // stem_1 map coordinates of stem_1-region pixel (0, 0)
    const int iy0 = 2 * sy0 - 1, ix0 = 2 * sx0 - 1;           // image coordinates of image-region pixel (0, 0): 3 ix0 = 3 (mod 4).  LDS
    // half index of region byte e: e + 4 (the aligned word holding byte e = 0 starts at half 1): byte 6 rx -> half 6 rx + 4, 4-byte aligned
    // ---- phase 0: image region -> f16 rows (zero outside the image: the conv's padding) ----------------------------------------
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
                amax = pf_amax(amax, (float)(wv[it] >> 24));       // (any byte: the guard only needs the order of magnitude, <= 255)
            }
        }
        if (t + (int)gridDim.x < ntiles) request(t + gridDim.x);  // in flight across this tile's five phases
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
            s_ih[ry * RS + x3 + 4] = hv;
            s_il[ry * RS + x3 + 4] = pf_split_lo(v, hv);
            amax = pf_amax(amax, v);
        }
    }
    for (int r = tid; r < MR1; r += NTHR) {
        const int ry = r / SRW, rx = r - ry * SRW;
        s_ok[r] = (r < S1R && (unsigned)(sy0 + ry) < (unsigned)a.SH && (unsigned)(sx0 + rx) < (unsigned)a.SW) ? 1 : 0;
    }
    __syncthreads();

    // ---- stem_1 on the stem_1 region -> s_p1 ------------------------------------------------------------------------------------
    for (int mt = wave; mt < MR1 / 16; mt += NW) {
        const int r = mt * 16 + frow;
        const int rc = r < S1R ? r : 0;
        const int ry = rc / SRW, rx = rc - ry * SRW;
        // this lane's 8 k values: groups 0..2 = halves 0..7 of image row 2 ry + g, group 3 = half 8 of the three rows
        pf_half8 xh, xl;
        {
            const int base = (2 * ry + (g < 3 ? g : 0)) * RS + 6 * rx + 4;
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
        acc = pf_mfma_16x16x32_f16(w1l[0], xh, acc);
        if constexpr (F32IN) acc = pf_mfma_16x16x32_f16(w1h[0], xl, acc);
        acc = pf_mfma_16x16x32_f16(w1h[0], xh, acc);
        const pf_f32x4 v = det_silu4(acc, s1, b1v);
        pf_half4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const pf_half hv = (pf_half)v[e];
            hi[e] = hv;
            lo[e] = pf_split_lo(v[e], hv);
            amax = pf_amax(amax, v[e]);
        }
        // channels g4 .. g4 + 3: slot (g >> 1) holds hi of channels 8 (g >> 1) .. + 7, slot 2 + (g >> 1) the lo halves
        *reinterpret_cast<pf_half4*>(s_p1 + pf_lds_chunk_off(r, g >> 1) + (g & 1) * 8) = hi;
        *reinterpret_cast<pf_half4*>(s_p1 + pf_lds_chunk_off(r, 2 + (g >> 1)) + (g & 1) * 8) = lo;
    }
    __syncthreads();

    // ---- stem_2a (16 -> 8) on the stem_1 region -> s_p2, zero outside the stem_1 map (stem_2b pads ITS input) ----------------
    for (int mt = wave; mt < MR1 / 16; mt += NW) {
        const int r = mt * 16 + frow;
        pf_half8 xh = pf_half8{(pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0}, xl = xh;
        if (g < 2) {
            xh = *reinterpret_cast<const pf_half8*>(s_p1 + pf_lds_chunk_off(r, g));
            xl = *reinterpret_cast<const pf_half8*>(s_p1 + pf_lds_chunk_off(r, 2 + g));
        }
        pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        acc = pf_mfma_16x16x32_f16(w2al[0], xh, acc);
        acc = pf_mfma_16x16x32_f16(w2ah[0], xl, acc);
        acc = pf_mfma_16x16x32_f16(w2ah[0], xh, acc);
        if (g < 2) {                                  // output channels 0 .. 7 exist
            pf_f32x4 v = det_silu4(acc, a.s2a, b2av);
            if (!s_ok[r]) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            pf_half4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const pf_half hv = (pf_half)v[e];
                hi[e] = hv;
                lo[e] = pf_split_lo(v[e], hv);
                amax = pf_amax(amax, v[e]);
            }
            *reinterpret_cast<pf_half4*>(s_p2 + r * 32 + g * 8) = hi;
            *reinterpret_cast<pf_half4*>(s_p2 + r * 32 + 16 + g * 8) = lo;
        }
    }
    __syncthreads();

    // ---- stem_2b (3x3 stride 2 on s_p2) -> cat channels 0 .. 15; max-pool 2x2 of stem_1 -> cat channels 16 .. 31 -----------------
    for (int mt = wave; mt < MRD / 16; mt += NW) {
        const int p = mt * 16 + frow;
        const int pc = p < P ? p : 0;
        const int py = pf_div_small(pc, m_tw), px = pc - py * a.TW;
        pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int tap = 4 * ks + g < 9 ? 4 * ks + g : 8;         // k groups past tap 8 carry zero weights
            const unsigned char* q = s_p2 + ((2 * py + tap / 3) * SRW + 2 * px + tap % 3) * 32;
            const pf_half8 xh = *reinterpret_cast<const pf_half8*>(q), xl = *reinterpret_cast<const pf_half8*>(q + 16);
            acc = pf_mfma_16x16x32_f16(w2bl[ks], xh, acc);
            acc = pf_mfma_16x16x32_f16(w2bh[ks], xl, acc);
            acc = pf_mfma_16x16x32_f16(w2bh[ks], xh, acc);
        }
        pf_f32x4 v = det_silu4(acc, a.s2b, b2bv);
        if (p >= P) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        det_park4(s_cat, MRD, p, g, v, amax);
    }
    for (int i = tid; i < MRD * 4; i += NTHR) {
        const int p = i >> 2, cg = i & 3;                     // 4 channels 4 cg .. 4 cg + 3 of output pixel p
        pf_half4 bh = pf_half4{(pf_half)0, (pf_half)0, (pf_half)0, (pf_half)0}, bl = bh;
        if (p < P) {
            const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
            float best[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int r = (2 * py + 1 + (d >> 1)) * SRW + 2 * px + 1 + (d & 1);
                if (!s_ok[r]) continue;                       // ceil_mode: windows over the map's edge take what exists
                const pf_half4 h = *reinterpret_cast<const pf_half4*>(s_p1 + pf_lds_chunk_off(r, cg >> 1) + (cg & 1) * 8);
                const pf_half4 l = *reinterpret_cast<const pf_half4*>(s_p1 + pf_lds_chunk_off(r, 2 + (cg >> 1)) + (cg & 1) * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = (float)h[e] + (float)l[e];
                    if (f > best[e]) { best[e] = f; bh[e] = h[e]; bl[e] = l[e]; }
                }
            }
        }
        unsigned char* q = det_plane(s_cat, MRD, 0) + pf_lds_chunk_off(p, 2 + (cg >> 1)) + (cg & 1) * 8;      // channels 16 + 4 cg ..
        *reinterpret_cast<pf_half4*>(q) = bh;
        *reinterpret_cast<pf_half4*>(q + (size_t)MRD * 64) = bl;
    }
    __syncthreads();

    // ---- stem_3 (32 -> 16) -> global ------------------------------------------------------------------------------------------------------
    float* out = a.out + (size_t)b * a.OH * a.OW * a.outLd;
    for (int mt = wave; mt < MRD / 16; mt += NW) {
        const pf_f32x4 v = det_silu4(det_tile<1>(s_cat, MRD, mt * 16, lane, w3h, w3l), a.s3, b3v);
        const int p = mt * 16 + frow;
        const int py = pf_div_small(p, m_tw), px = p - py * a.TW;
        const int oy = oy0 + py, ox = ox0 + px;
        if (p < P && oy < a.OH && ox < a.OW) *reinterpret_cast<pf_f32x4*>(out + ((size_t)oy * a.OW + ox) * a.outLd + g4) = v;
    }
    __syncthreads();                                          // the next tile's rows overwrite everything
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}
