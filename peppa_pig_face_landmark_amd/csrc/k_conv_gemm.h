// Implicit-GEMM convolution on the CDNA4 matrix cores (1x1 and kxk dense convs, NHWC).
//
// Replaces what onnxruntime's CPU conv kernels compute for the reference
// (Skps/core/api/onnx_model_base.py:23-24) for every dense conv of the landmark regressor
// (TRAIN/face_landmark/lib/core/base_trainer/model.py: 1x1 expand/project convs of the encoder,
// ASPP :70-83, DecoderBlock conv1/conv2 :146-172, hm head :271) and of the detector.
//
// GEMM view:  D[n][m] = sum_k  W[n][k] * X[m][k]
//   m = output pixel (b, oy, ox) flattened,  n = output channel,  k = (tap, input channel)
//   MFMA "A" operand = weight rows, "B" operand = pixels, so each lane ends up owning 4 consecutive
//   output channels of one pixel (contiguous in NHWC -> one 8/16-byte store per accumulator).
//
// Tiling: block = 256 threads = 4 waves arranged WARPS_M x WARPS_N over a BM(pixels) x BN(channels)
// tile; K advances 64 bytes per step (32 f16 / 16 f32 per row) through two LDS stages
// (register-staged global->LDS copy overlapping the MFMAs of the previous stage).  The 16-byte
// chunk index inside a 64-byte LDS row is rotated by 2*(row>>2) so that the four 16-lane groups
// of a ds_read_b128 fragment read hit 16 distinct bank slots.
//
// Fused epilogue: + bias[n] (BN folded) (+ per-face bias) (+ residual) -> activation -> store,
// optional SE gate on the input channels (applied while staging), optional per-(face,channel)
// running arg-max for the heat-map head (COTRAIN.postp, model.py:520-522).
#pragma once
#include "pf_common.h"
#include <type_traits>
#include <utility>

struct ConvGemmArgs {
    const void* in;
    const void* wt;       // [Npad][KH*KW][Cpad], element type T, zero padded
    const float* bias;    // [Npad]
    void* out;
    const void* res;      // residual (same pixel indexing as out) or nullptr
    const float* gate;    // [B][inC] multiplicative gate on input channels, or nullptr
    const float* fbias;   // [B][Npad] per-face bias, or nullptr
    float* amax_val;      // [B][amaxN][nslots] partial maxima, or nullptr
    int* amax_idx;
    int B, inH, inW, inC, inLd;
    int outH, outW, N, Npad, outLd, outCs;  // channel n is stored at element n*outCs of the pixel row
    int outCpad;          // channels [N, outCpad) of the output view are written as zeros (vector padding)
    int resLd;
    int KH, KW, stride, pad, dil, Cpad;
    int act;
    int amaxN;
    int head_segs;        // pw_head_kernel (k_pwhead.h): work items per face
    int store_out;
    float acc_scale;      // split-precision kernels: 1 / (power-of-two weight scale); 1 otherwise
    // fused "upsample x2 (bilinear) + concat + depthwise 3x3 + BN" producer of the pixel operand
    // (DecoderBlock.forward model.py:184-189 + SeparableConv2d.conv_dw :21-27), STAGE == 1 kernels only
    const float* up_lo;   // [B][loH][loW][loLd]  channels [0, C1)   -> upsampled
    const float* up_skip; // [B][2loH][2loW][skipLd] channels [C1, C1+C2)
    const float* dw_w;    // [16 position classes][9][C1]: upsample (x) depthwise collapsed onto the low-res grid
    const float* dw_w2;   // [9][C2] plain depthwise weights of the skip channels (BN folded)
    const float* dw_b;    // [C1+C2]
    int loH, loW, C1, loLd, skipLd;
    // fused "pointwise expand -> depthwise kxk" (EPI != 0 kernels): dw_w2 = [K*K][N] depthwise weights, dw_b = [N]
    // bias, both BN folded; the depthwise output goes to `out`, its per-face channel means (SE squeeze) to gap_out
    float* gap_out;       // [B][N] or nullptr
    unsigned* range_slot; // f32s range guard: max |v| (raw bits) over everything this launch splits into f16 hi / lo, or nullptr
    // timing ablations of conv3x3_halo_split_kernel (PEPPA_DBG bit mask, 0 in production; results are WRONG when set):
    // 1 = weights fetched for the first K step only, 16 = no MFMAs, 32 = no output stores, 64 = input patch staged for the
    // first channel chunk only, 128 = no per-tap barrier; conv_gemm_split_kernel: 16 as above, 256 = operands (pixels AND weights)
    // fetched for the first K step only, 512 = no split / LDS store of the pixel operand, 1024 = no depthwise taps in the fused
    // expand + depthwise epilogue.  Tables: profiles/r02_*ablations.md.
    int dbg;
};

template <typename T> struct ConvMma;
template <> struct ConvMma<pf_half> {
    static constexpr int KSUB = 1;   // one v_mfma_f32_16x16x32_f16 consumes the whole 64-byte K step
    __device__ static __forceinline__ pf_f32x4 step(pf_half8 w, pf_half8 x, pf_f32x4 c, int) {
        return pf_mfma_16x16x32_f16(w, x, c);
    }
};
template <> struct ConvMma<float> {
    static constexpr int KSUB = 4;   // four v_mfma_f32_16x16x4_f32 per 64-byte K step
    __device__ static __forceinline__ pf_f32x4 step(pf_f32x4 w, pf_f32x4 x, pf_f32x4 c, int ks) {
        return pf_mfma_16x16x4_f32(w[ks], x[ks], c);
    }
};

__device__ __forceinline__ int pf_lds_chunk_off(int row, int chunk) {
    return row * 64 + (((chunk + 2 * (row >> 2)) & 3) << 4);
}

// ---- fused epilogue shared by the direct and the split-precision kernels ----------------------
// acc[j][i][r] = D[channel n0 + wn*WN + 16j + 4*(lane>>4) + r][pixel m0 + wm*WM + 16i + (lane&15)]
template <typename T, int BM, int BN, int WARPS_M, int WARPS_N>
__device__ __forceinline__ void conv_gemm_epilogue(const ConvGemmArgs& a, pf_f32x4 (&acc)[BN / WARPS_N / 16][BM / WARPS_M / 16],
                                                   int m0, int n0, int wm, int wn, int lane, int M, int OHW, float acc_scale) {
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    T* __restrict__ out = static_cast<T*>(a.out);
    const T* __restrict__ res = static_cast<const T*>(a.res);
    const int pcol = lane & 15;        // pixel within the 16-wide sub-tile
    const int crow = (lane >> 4) * 4;  // first of the 4 channels this lane owns
    const bool want_amax = a.amax_val != nullptr;

    // every bias of this lane in one round trip (Npad is a multiple of 16: a lane's four channels are inside or outside together)
    pf_f32x4 bvv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WN + j * 16 + crow;
        bvv[j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        if (n + 3 < a.Npad) bvv[j] = *reinterpret_cast<const pf_f32x4*>(a.bias + n);
    }
    const bool res_vec = res != nullptr && sizeof(T) == 4 && (a.resLd & 3) == 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WN + j * 16 + crow;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = bvv[j][r];
        // the residual vectors of this channel tile's MT pixel sub-tiles: one round trip per channel tile, not one per sub-tile
        pf_f32x4 rvv[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WM + i * 16 + pcol;
            rvv[i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (res_vec && m < M && n + 3 < a.N) rvv[i] = *reinterpret_cast<const pf_f32x4*>(reinterpret_cast<const float*>(res) + (size_t)m * a.resLd + n);
        }
        float best_v[4];
        int best_i[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { best_v[r] = -3.0e38f; best_i[r] = 0x7fffffff; }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WM + i * 16 + pcol;
            const bool mok = m < M;
            const int b = mok ? m / OHW : 0;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r] * acc_scale + bv[r];
            if (a.fbias && mok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < a.Npad) v[r] += a.fbias[(size_t)b * a.Npad + n + r];
            }
            if (res && mok) {
                if (res_vec && n + 3 < a.N) {      // one 16-byte load (views start on vector boundaries), requested above
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rvv[i][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) v[r] += (float)res[(size_t)m * a.resLd + n + r];
                }
            }
            pf_act_n<4>(v, a.act);
            if (want_amax && mok) {
                const int local = m - b * OHW;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (v[r] > best_v[r]) { best_v[r] = v[r]; best_i[r] = local; }
            }
            if (a.store_out && mok && !(pf_dbg(a) & 32)) {
                T* o = out + (size_t)m * a.outLd;
                if (a.outCs == 1 && n + 3 < a.N) {
                    if constexpr (sizeof(T) == 2) {
                        pf_half4 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pk[r] = (pf_half)v[r];
                        *reinterpret_cast<pf_half4*>(o + n) = pk;
                    } else {
                        pf_f32x4 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pk[r] = v[r];
                        *reinterpret_cast<pf_f32x4*>(o + n) = pk;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r < a.N) o[(size_t)(n + r) * a.outCs] = (T)v[r];
                        else if (n + r < a.outCpad) o[(size_t)(n + r) * a.outCs] = (T)0.f;
                    }
                }
            }
        }
        if (want_amax) {
            // all BM pixels of this block belong to one face (host guarantees OHW % BM == 0)
#pragma unroll
            for (int mask = 1; mask < 16; mask <<= 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ov = pf_shfl_xor_f32(best_v[r], mask);
                    const int oi = pf_shfl_xor_i32(best_i[r], mask);
                    if (ov > best_v[r] || (ov == best_v[r] && oi < best_i[r])) { best_v[r] = ov; best_i[r] = oi; }
                }
            }
            if (pcol == 0 && m0 < M) {
                const int b = m0 / OHW;
                const int blocks_per_face = OHW / BM;
                const int nslots = blocks_per_face * WARPS_M;
                const int slot = ((m0 - b * OHW) / BM) * WARPS_M + wm;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < a.amaxN) {
                        const size_t o = ((size_t)b * a.amaxN + n + r) * nslots + slot;
                        a.amax_val[o] = best_v[r];
                        a.amax_idx[o] = best_i[r];
                    }
                }
            }
        }
    }
}

// ---- arg-max-only epilogue of the heat-map score head (COTRAIN.postp, model.py:520-522) ----------------------------------
// The generic epilogue above serves every conv of both networks through run-time switches (residual, per-face bias, five
// activations, strided stores, optional arg-max); on the score head -- bias only, nothing stored, 128-pixel tiles that
// never straddle a face -- those switches and the ds_bpermute shuffles of its arg-max were most of the kernel (SQ
// counters: VALU 56 % busy, MFMA 15 %).  This one does exactly the head's work: bias, running (max, first index) per
// lane, then a 16-lane reduction by DPP row exchanges.  Host guarantees: OHW % BM == 0 (so every pixel of the tile exists and
// belongs to one face), no residual / per-face bias / gate / activation, store_out == 0.
template <int BM, int BN, int WARPS_M, int WARPS_N>
__device__ __forceinline__ void conv_gemm_argmax_epilogue(const ConvGemmArgs& a, pf_f32x4 (&acc)[BN / WARPS_N / 16][BM / WARPS_M / 16],
                                                          int m0, int n0, int wm, int wn, int lane, int OHW, float acc_scale) {
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    const int pcol = lane & 15;
    const int crow = (lane >> 4) * 4;
    const int b = m0 / OHW;
    const int local0 = m0 - b * OHW + wm * WM + pcol;
    const int nslots = (OHW / BM) * WARPS_M;
    const int slot = ((m0 - b * OHW) / BM) * WARPS_M + wm;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WN + j * 16 + crow;
        float best_v[4];
        int best_i[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bv = (n + r < a.Npad) ? a.bias[n + r] : 0.f;
            best_v[r] = fmaf(acc[j][0][r], acc_scale, bv);
            best_i[r] = local0;
#pragma unroll
            for (int i = 1; i < MT; ++i) {           // ascending pixel index: strict > keeps the first maximum
                const float v = fmaf(acc[j][i][r], acc_scale, bv);
                if (v > best_v[r]) { best_v[r] = v; best_i[r] = local0 + i * 16; }
            }
        }
#define PF_AMAX_STEP(STEP)                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                              \
        const float ov = pf_row_xchg_f32<STEP>(best_v[r]);                                                    \
        const int oi = pf_row_xchg_i32<STEP>(best_i[r]);                                                      \
        if (ov > best_v[r] || (ov == best_v[r] && oi < best_i[r])) { best_v[r] = ov; best_i[r] = oi; }        \
    }
        PF_AMAX_STEP(0) PF_AMAX_STEP(1) PF_AMAX_STEP(2) PF_AMAX_STEP(3)
#undef PF_AMAX_STEP
        if (pcol == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < a.amaxN) {
                    const size_t o = ((size_t)b * a.amaxN + n + r) * nslots + slot;
                    a.amax_val[o] = best_v[r];
                    a.amax_idx[o] = best_i[r];
                }
        }
    }
}

// KS = 1: pointwise conv (1x1, stride 1, no padding) -- tap arithmetic compiled out;
// KS = 3: general kxk conv (any kernel size / stride / dilation / padding).
template <typename T, int BM, int BN, int WARPS_M, int WARPS_N, int KS>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvGemmArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;       // elements per 16-byte chunk
    constexpr int KE = 4 * VE;            // elements per 64-byte K step
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int XROWS = BM / 64;                     // pixel rows staged per thread
    constexpr int WROWS = (BN + 63) / 64;              // weight rows staged per thread
    constexpr int STAGE_BYTES = (BM + BN) * 64;
    static_assert(WARPS_M * WARPS_N == 4, "4 waves per block");
    static_assert(BM % 64 == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    // XCD-aware tile order for kxk convs: workgroup b runs on XCD b % 8 (private 4 MiB L2 each), so
    // give every XCD a contiguous run of pixel tiles -- vertically adjacent tiles, which share their
    // halo rows, then meet in the same L2.  Pure speed choice; any mapping is correct.
    int mtile = blockIdx.x;
    if (KS != 1 && (gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = mtile * BM;
    const int n0 = blockIdx.y * BN;
    const int OHW = a.outH * a.outW;
    const int M = a.B * OHW;
    const T* __restrict__ in = static_cast<const T*>(a.in);
    const T* __restrict__ wt = static_cast<const T*>(a.wt);

    // ---- per-thread staging geometry -------------------------------------------------
    const int chunk = t & 3;
    const int srow = t >> 2;  // 0..63
    int xb[XROWS], xiy0[XROWS], xix0[XROWS];
    bool xvalid[XROWS];
#pragma unroll
    for (int r = 0; r < XROWS; ++r) {
        const int m = m0 + srow + 64 * r;
        xvalid[r] = m < M;
        const int mm = xvalid[r] ? m : 0;
        const int b = mm / OHW;
        const int rem = mm - b * OHW;
        const int oy = rem / a.outW;
        const int ox = rem - oy * a.outW;
        xb[r] = b;
        xiy0[r] = KS == 1 ? oy : oy * a.stride - a.pad;
        xix0[r] = KS == 1 ? ox : ox * a.stride - a.pad;
    }
    const int taps = KS == 1 ? 1 : a.KH * a.KW;
    const int cchunks = a.Cpad / KE;
    const int nk = taps * cchunks;
    const size_t wrow_stride = (size_t)taps * a.Cpad;

    vec_t xreg[XROWS], wreg[WROWS];

    auto load_tile = [&](int tap, int cc) {
        const int ky = KS == 1 ? 0 : tap / a.KW;
        const int kx = KS == 1 ? 0 : tap - ky * a.KW;
        const int kelem = cc * KE + chunk * VE;
        const bool kok = kelem < a.inC;
#pragma unroll
        for (int r = 0; r < XROWS; ++r) {
            const int iy = KS == 1 ? xiy0[r] : xiy0[r] + ky * a.dil;
            const int ix = KS == 1 ? xix0[r] : xix0[r] + kx * a.dil;
            const bool ok = xvalid[r] && kok && (KS == 1 || ((unsigned)iy < (unsigned)a.inH && (unsigned)ix < (unsigned)a.inW));
            vec_t v = pf_zero_vec<T>();
            if (ok) {
                const size_t off = ((size_t)(xb[r] * a.inH + iy) * a.inW + ix) * a.inLd + kelem;
                v = pf_ldv<T>(in + off);
                if (a.gate) {
                    const float* g = a.gate + (size_t)xb[r] * a.inC + kelem;
#pragma unroll
                    for (int e = 0; e < VE; ++e) v[e] = (T)((float)v[e] * g[e]);
                }
            }
            xreg[r] = v;
        }
#pragma unroll
        for (int r = 0; r < WROWS; ++r) {
            const int row = srow + 64 * r;
            const int n = n0 + row;
            vec_t v = pf_zero_vec<T>();
            if (row < BN && n < a.Npad) v = pf_ldv<T>(wt + (size_t)n * wrow_stride + (size_t)tap * a.Cpad + kelem);
            wreg[r] = v;
        }
    };
    auto store_tile = [&](int stage) {
        unsigned char* xs = smem + stage * STAGE_BYTES;
        unsigned char* ws = xs + BM * 64;
#pragma unroll
        for (int r = 0; r < XROWS; ++r)
            *reinterpret_cast<vec_t*>(xs + pf_lds_chunk_off(srow + 64 * r, chunk)) = xreg[r];
#pragma unroll
        for (int r = 0; r < WROWS; ++r) {
            const int row = srow + 64 * r;
            if (row < BN) *reinterpret_cast<vec_t*>(ws + pf_lds_chunk_off(row, chunk)) = wreg[r];
        }
    };

    pf_f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop ---------------------------------------------------------------------
    int tap = 0, cc = 0;
    load_tile(tap, cc);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            // channel-chunk outer, tap inner: the KH*KW shifted reads of one 64-byte channel chunk are
            // issued back to back, so the halo re-reads hit L1/L2 instead of going back to the fabric
            if (++tap == taps) { tap = 0; ++cc; }
            load_tile(tap, cc);
        }
        const unsigned char* xs = smem + cur * STAGE_BYTES;
        const unsigned char* ws = xs + BM * 64;
        vec_t xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
            xf[i] = *reinterpret_cast<const vec_t*>(xs + pf_lds_chunk_off(wm * WM + i * 16 + frow, fchunk));
#pragma unroll
        for (int j = 0; j < NT; ++j)
            wf[j] = *reinterpret_cast<const vec_t*>(ws + pf_lds_chunk_off(wn * WN + j * 16 + frow, fchunk));
        // issue order: all (j,i) accumulators for one k sub-step before the next sub-step, so that
        // consecutive MFMAs never depend on each other (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle
        // dependent latency)
#pragma unroll
        for (int ks = 0; ks < ConvMma<T>::KSUB; ++ks)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[j][i] = ConvMma<T>::step(wf[j], xf[i], acc[j][i], ks);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    conv_gemm_epilogue<T, BM, BN, WARPS_M, WARPS_N>(a, acc, m0, n0, wm, wn, lane, M, OHW, 1.0f);
}

// ---- fused depthwise epilogue (MobileNetV3 inverted residual, expand -> depthwise, model.py:252-264) -------
// The workgroup's BM pixels are BM / (H*W) WHOLE images (host guarantees H*W divides BM, W <= 16), so the
// expanded tile act(acc) can stay in LDS and the k x k depthwise conv (+bias, act) reads it from there:
// the expanded tensor -- the largest one of the block -- never exists in HBM.  Thread = (channel, image row):
// per filter row it loads the W-pixel input row once and slides the filter along it in registers.  Also
// emits the per-face channel means of the depthwise output (the SE squeeze), complete because a workgroup
// owns whole images.
// WS = compile-time image width (16: the shape of every such layer of the Student at 256 x 256; 0 = read it from the
// arguments): with the width known the per-pixel "x < W" selects, the row / image index divisions and the padding tests
// fold away -- the epilogue is VALU-bound (57 % VALU busy, 12 % MFMA by SQ counters), so instruction count is its time.
template <int BM, int BN, int WARPS_M, int WARPS_N, int K, int DIL, int WS>
__device__ __forceinline__ void expdw_epilogue(const ConvGemmArgs& a, pf_f32x4 (&acc)[BN / WARPS_N / 16][BM / WARPS_M / 16],
                                               unsigned char* smem, int m0, int n0, int wm, int wn, int t, int M) {
    constexpr int NTHR = WARPS_M * WARPS_N * 64;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int ES = BN + 4;                 // E row stride (floats)
    constexpr int NG = NTHR / BN;              // row groups
    constexpr int MAXF = 4;                    // images per workgroup
    constexpr int PAD = DIL * (K - 1) / 2;
    constexpr int MAXW = 16;
    static_assert(NTHR % BN == 0, "thread = (channel, row group)");
    float* es = reinterpret_cast<float*>(smem);          // [BM][ES]
    float* sums = es + BM * ES;                           // [NG][MAXF][BN]
    const int lane = t & 63;
    const int pcol = lane & 15, crow = (lane >> 4) * 4;
    const int c = t % BN, g = t / BN;
    const int n = n0 + c;
    const bool cok = n < a.N;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nl = wn * WN + j * 16 + crow;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (n0 + nl + r < a.Npad) ? a.bias[n0 + nl + r] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            pf_f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[j][i][r], a.acc_scale, bv[r]);
            pf_act_rh<4>(v, a.act);
            *reinterpret_cast<pf_f32x4*>(es + (wm * WM + i * 16 + pcol) * ES + nl) = v;
        }
    }
    if constexpr (WS == 16 && BM == 256 && BN == 64 && NTHR == 512) {
        // One 16 x 16 image per workgroup: thread = (channel PAIR, image row).  Two adjacent channels of a pixel are one aligned
        // 8-byte LDS word, so every tap is a v_pk_fma_f32 on a ds_read_b64 operand -- half the VALU and LDS instructions of the
        // one-channel-per-thread form below (the launch is VALU-bound: 57 % VALU busy against 12 % matrix pipe).  The filter taps
        // sit in LDS (6.4 KB at 5 x 5) instead of 2 x 25 registers.  Same fma order per output as the generic path.
        constexpr int KK = K * K;
        float* wks = sums + 16 * BN;                     // [K * K][BN]
        static_assert((BM * ES + 16 * BN + KK * BN) * 4 <= 80 * 1024, "E tile + row sums + taps: two workgroups per CU");
        {
            float wv[(KK * BN + NTHR - 1) / NTHR];
#pragma unroll
            for (int i = 0; i < (KK * BN + NTHR - 1) / NTHR; ++i) {
                const int id = t + i * NTHR;
                const int k = id / BN, cc = id - k * BN;
                wv[i] = (id < KK * BN && n0 + cc < a.N) ? a.dw_w2[(size_t)k * a.N + n0 + cc] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < (KK * BN + NTHR - 1) / NTHR; ++i)
                if (t + i * NTHR < KK * BN) wks[t + i * NTHR] = wv[i];
        }
        const int c2 = (t & 31) * 2, row = t >> 5;
        const int n2 = n0 + c2;
        pf_f32x2 bd2;
        bd2[0] = n2 < a.N ? a.dw_b[n2] : 0.f;
        bd2[1] = n2 + 1 < a.N ? a.dw_b[n2 + 1] : 0.f;
        __syncthreads();
        pf_f32x2 o[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) o[x] = bd2;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yy = row + ky * DIL - PAD;
            if ((unsigned)yy >= 16u || (pf_dbg(a) & 1024)) continue;
            const float* erow = es + (yy * 16) * ES + c2;
            pf_f32x2 in[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) in[x] = *reinterpret_cast<const pf_f32x2*>(erow + x * ES);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const pf_f32x2 w = *reinterpret_cast<const pf_f32x2*>(wks + (ky * K + kx) * BN + c2);
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int xx = x + kx * DIL - PAD;       // compile-time register index
                    if (xx >= 0 && xx < 16) o[x] = __builtin_elementwise_fma(w, in[xx], o[x]);
                }
            }
            asm volatile("" ::: "memory");       // one filter row's LDS reads in flight at a time (register footprint)
        }
        float of[32];
#pragma unroll
        for (int x = 0; x < 16; ++x) { of[2 * x] = o[x][0]; of[2 * x + 1] = o[x][1]; }
        pf_act_rh<32>(of, a.act);
        const int m = m0 + row * 16;
        pf_f32x2 rs = pf_f32x2{0.f, 0.f};
        float* out = static_cast<float*>(a.out);
        if (m < M && !(pf_dbg(a) & 32)) {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                float* po = out + (size_t)(m + x) * a.outLd + n2;
                if (n2 + 1 < a.N) *reinterpret_cast<pf_f32x2*>(po) = pf_f32x2{of[2 * x], of[2 * x + 1]};
                else if (n2 < a.N) po[0] = of[2 * x];
                rs[0] += of[2 * x];
                rs[1] += of[2 * x + 1];
            }
        }
        if (a.gap_out) {
            *reinterpret_cast<pf_f32x2*>(sums + row * BN + c2) = rs;
            __syncthreads();
            if (t < BN && n0 + t < a.N) {
                float tot = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) tot += sums[r * BN + t];
                const int b = m0 / 256;
                if (b < a.B) a.gap_out[(size_t)b * a.N + n0 + t] = tot / 256.f;
            }
        }
        return;
    }
    float wk[K * K];                           // requested before the barrier (accumulators are dead by now)
#pragma unroll
    for (int k = 0; k < K * K; ++k) wk[k] = cok ? a.dw_w2[(size_t)k * a.N + n] : 0.f;
    const float bd = cok ? a.dw_b[n] : 0.f;
    __syncthreads();
    const int W = WS ? WS : a.outW, H = WS ? WS : a.outH, OHW = H * W;
    const int rows = BM / W;
    float fsum[MAXF];
#pragma unroll
    for (int f = 0; f < MAXF; ++f) fsum[f] = 0.f;
    float* out = static_cast<float*>(a.out);
    for (int r = g; r < rows; r += NG) {
        const int f = r / H, y = r - f * H;
        float o[MAXW];
#pragma unroll
        for (int x = 0; x < MAXW; ++x) o[x] = bd;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yy = y + ky * DIL - PAD;
            if ((unsigned)yy >= (unsigned)H) continue;
            const float* erow = es + ((f * H + yy) * W) * ES + c;
            float in[MAXW];
#pragma unroll
            for (int x = 0; x < MAXW; ++x) in[x] = x < W ? erow[x * ES] : 0.f;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float w = wk[ky * K + kx];
#pragma unroll
                for (int x = 0; x < MAXW; ++x) {
                    const int xx = x + kx * DIL - PAD;       // compile-time register index
                    if (xx >= 0 && xx < MAXW) o[x] = fmaf(w, in[xx], o[x]);   // in[xx] is 0 beyond the image width
                }
            }
            asm volatile("" ::: "memory");       // one filter row's LDS reads in flight at a time (register footprint)
        }
        pf_act_rh<MAXW>(o, a.act);
        const int m = m0 + r * W;                             // first pixel of the row
        if (cok && m < M) {
            float rs = 0.f;
#pragma unroll
            for (int x = 0; x < MAXW; ++x)
                if (x < W) {
                    out[(size_t)(m + x) * a.outLd + n] = o[x];
                    rs += o[x];
                }
#pragma unroll
            for (int ff = 0; ff < MAXF; ++ff)
                if (ff == f) fsum[ff] += rs;
        }
    }
    if (a.gap_out) {
#pragma unroll
        for (int f = 0; f < MAXF; ++f) sums[(g * MAXF + f) * BN + c] = fsum[f];
        __syncthreads();
        const int faces = BM / OHW;
        if (t < faces * BN) {
            const int f = t / BN, cc = t - f * BN;
            float tot = 0.f;
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) tot += sums[(gg * MAXF + f) * BN + cc];
            const int b = m0 / OHW + f;
            if (b < a.B && n0 + cc < a.N) a.gap_out[(size_t)b * a.N + n0 + cc] = tot / (float)OHW;
        }
    }
}

// =============================================================================================
// Split-precision variant: f32 tensors in HBM, f16 matrix cores, f32-grade results.
//
// Every f32 operand is written as hi + lo with hi = f16(v), lo = f16(v - hi) (22 significand bits) and
// the product is accumulated as  wh*xh + wh*xl + wl*xh  on v_mfma_f32_16x16x32_f16 (f32 accumulate; the
// dropped wl*xl term is 2^-22 relative).  Three f16 MFMAs replace eight v_mfma_f32_16x16x4_f32 per 32 k,
// i.e. ~5x the matrix throughput of the exact-f32 path at the same accuracy (measured against float64:
// both 3.9e-7 of the output range on the hero layer's shape).  Weights are split at pack time and scaled
// by a per-layer power of two so that their lo parts stay clear of f16 subnormals (undone by acc_scale);
// activations are split while they are staged into LDS.
//
// K step = 32 elements: per row 64 B of hi + 64 B of lo in LDS (same chunk rotation as above).
// Weight rows in HBM: [taps][Cpad/32][hi: 32 x f16 | lo: 32 x f16].
// STAGE = 0: the pixel operand is read from a.in.  STAGE = 1 (pointwise only): it is produced on the
// fly -- bilinear x2 upsample of up_lo / pass-through of up_skip, depthwise 3x3 (+bias) -- so the
// concatenated and the depthwise tensors never exist in HBM.
// EPI_K != 0 (pointwise only): the epilogue is the fused depthwise EPI_K x EPI_K conv (dilation EPI_DIL) above.
// NK > 0 (plain pointwise convs only; host: Cpad == 32 NK): the K loop is unrolled completely and runs TWO steps ahead -- see the
// PW2 path in the body.
template <int N, typename F, int... I> __device__ __forceinline__ void pf_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void pf_static_for(F&& f) { pf_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int BM, int BN, int WARPS_M, int WARPS_N, int KS, int STAGE = 0, int EPI_K = 0, int EPI_DIL = 1, int EPI_W = 0, int NK = 0>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64, BN >= 256 ? WARPS_M * WARPS_N / 4 : WARPS_M * WARPS_N / 2) void conv_gemm_split_kernel(ConvGemmArgs a) {
    // second launch bound = waves per SIMD for two resident workgroups per CU (<= 128 VGPRs at 8 waves);
    // 256-channel tiles hold 64 accumulators + 64 weight-fragment registers and run one workgroup per CU;
    // the fused-depthwise variants keep an 80 KB tile in LDS (one workgroup per CU) and may use 256
    constexpr int NTHR = WARPS_M * WARPS_N * 64;       // 256 or 512 threads (8 waves hide the staging latency)
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int XUNITS = (BM * 4 + NTHR - 1) / NTHR; // (row, 8-float unit) pairs staged per thread
    constexpr int XROWSTEP = NTHR / 4;                 // rows covered by one pass of the block
    constexpr int WCHUNKS = (BN * 8 + NTHR - 1) / NTHR;  // 16-byte weight chunks staged per thread
    constexpr int PLANE_X = BM * 64, PLANE_W = BN * 64;
    constexpr int W_BYTES = WCHUNKS * NTHR * 16;         // weight planes (hi | lo), rounded up to whole LDS-DMA passes
    constexpr int STAGE_BYTES = 2 * PLANE_X + W_BYTES;
    static_assert((NTHR == 256 || NTHR == 512) && WM % 16 == 0 && WN % 16 == 0 && WM > 0 && WN > 0, "tile shape");
    static_assert((BM * 4) % NTHR == 0, "pixel tile must split evenly over the block");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    int mtile = blockIdx.x;
    if (KS != 1 && (gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = mtile * BM;
    const int n0 = blockIdx.y * BN;
    const int OHW = a.outH * a.outW;
    const int M = a.B * OHW;
    const float* __restrict__ in = static_cast<const float*>(a.in);
    const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);

    // pixel staging: unit u of this thread = row (t>>2) + XROWSTEP*u, floats [8*(t&3), 8*(t&3)+8) of the K step
    const int xc = t & 3;
    const int xrow0 = t >> 2;
    int xb[XUNITS], xiy0[XUNITS], xix0[XUNITS];
    bool xvalid[XUNITS];
#pragma unroll
    for (int u = 0; u < XUNITS; ++u) {
        const int m = m0 + xrow0 + XROWSTEP * u;
        xvalid[u] = m < M;
        const int mm = xvalid[u] ? m : 0;
        const int b = mm / OHW;
        const int rem = mm - b * OHW;
        const int oy = rem / a.outW;
        const int ox = rem - oy * a.outW;
        xb[u] = b;
        // plain pointwise (stride 1, no padding): the input pixel IS output pixel m
        xiy0[u] = (KS == 1 && STAGE == 0) ? mm : (KS == 1 ? oy : oy * a.stride - a.pad);
        xix0[u] = (KS == 1 && STAGE == 0) ? 0 : (KS == 1 ? ox : ox * a.stride - a.pad);
    }
    const int taps = KS == 1 ? 1 : a.KH * a.KW;
    const int cblocks = a.Cpad / 32;
    const int nk = taps * cblocks;
    const size_t wrow_bytes = (size_t)taps * cblocks * 128;

    constexpr bool PW2 = KS == 1 && STAGE == 0;                       // plain pointwise conv: leaner operand staging (below)
    constexpr bool GATED = PW2 && EPI_K == 0;                         // ... which may carry an SE gate on its input channels
    static_assert(NK == 0 || (KS == 1 && STAGE == 0), "unrolled K loop: plain pointwise convs");
    pf_f32x4 xreg[NK > 1 ? 2 : 1][XUNITS][2];
    // The gate vector of the tile's face sits in LDS when the tile lies inside one face (every gated layer of the Student at
    // 256 x 256); otherwise each unit fetches its gate values when it is split (correct, no look-ahead: small crops only).
    // Multiplying right behind the pixel load put an s_waitcnt vmcnt(0) behind each of a K step's loads.
    // It gets its own 4 KB where two workgroups still fit a CU with it, else the tail of weight stage 0 that no row uses (W_BYTES
    // is rounded up to whole 512-slot DMA passes; load_w skips the slots beyond row BN - 1).
    constexpr int W_USED = BN * 128;
    constexpr bool GATE_SEP = 2 * STAGE_BYTES + 4096 <= 80 * 1024;
    constexpr int GATE_CAP = !GATED ? 0 : (GATE_SEP ? 1024 : (W_BYTES - W_USED) / 4);
    __shared__ __attribute__((aligned(16))) float sgate_sep[GATED && GATE_SEP ? 1024 : 4];
    float* sgate = GATE_SEP ? sgate_sep : reinterpret_cast<float*>(smem + 2 * PLANE_X + W_USED);
    const bool gate_lds = GATED && a.gate != nullptr && (OHW % BM) == 0 && a.Cpad <= GATE_CAP;
    unsigned amax = 0;                                 // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);

    // Weights are pre-split bytes: they go global -> LDS directly (no VGPRs, no ds_write pass, which costs 13
    // LDS-path cycles per 16 bytes against 4 for a read).  LDS slot s (16 B, lane-linear as the DMA requires) is
    // (plane, row, position) with the row's four chunks rotated; the rotation is applied to the SOURCE address.
    // Rows past Npad re-read the last row (their outputs are never stored); slots past the planes land in padding.
    auto load_w = [&](int tap, int cb, int stage) {
        unsigned char* wdst = smem + stage * STAGE_BYTES + 2 * PLANE_X;
#pragma unroll
        for (int c = 0; c < WCHUNKS; ++c) {
            const int sl = t + NTHR * c;
            const int plane = sl >= BN * 4 ? 1 : 0;
            const int r = (sl - plane * BN * 4) >> 2;
            const int row = r < BN ? r : BN - 1;
            const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
            const int n = min(n0 + row, a.Npad - 1);
            const unsigned char* src = wt + (size_t)n * wrow_bytes + ((size_t)tap * cblocks + cb) * 128 + plane * 64 + chunk * 16;
            if constexpr (PW2) { if (sl < BN * 8) pf_glds16(src, wdst + sl * 16); } else pf_glds16(src, wdst + sl * 16);   // (the tail may hold the gate)
        }
    };
    auto load_tile = [&](int tap, int cb, int stage) {
        const int ky = KS == 1 ? 0 : tap / a.KW;
        const int kx = KS == 1 ? 0 : tap - ky * a.KW;
        const int kelem = cb * 32 + xc * 8;
        if constexpr (STAGE == 1) {
#pragma unroll
            for (int u = 0; u < XUNITS; ++u) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = 0.f;
                const int y = xiy0[u], x = xix0[u];           // output pixel (KS == 1: no stride / padding)
                const int H = 2 * a.loH, W = 2 * a.loW;
                if (xvalid[u] && kelem < a.inC) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = a.dw_b[kelem + e];
                    if (kelem < a.C1) {
                        // The bilinear taps of the whole 3x3 depthwise window live in the 3x3 low-res patch around
                        // (y>>1, x>>1) (coordinates clamped), so upsample + depthwise collapse into ONE 3x3 filter
                        // on the low-res map whose weights depend only on the position class of (y, x):
                        // first / last / even / odd row  x  first / last / even / odd column  (16 classes,
                        // precomputed at pack time: E = A_cls^T . Wdw . B_cls, zero padding included).
                        const int my = y >> 1, mx = x >> 1;
                        const int ycls = y == 0 ? 0 : (y == H - 1 ? 1 : 2 + (y & 1));
                        const int xcls = x == 0 ? 0 : (x == W - 1 ? 1 : 2 + (x & 1));
                        const float* we = a.dw_w + (size_t)((ycls * 4 + xcls) * 9) * a.C1 + kelem;
                        const float* lo = a.up_lo + (size_t)xb[u] * a.loH * a.loW * a.loLd + kelem;
#pragma unroll 1
                        for (int j = 0; j < 3; ++j) {   // one patch row at a time keeps the live loads (and VGPRs) bounded
                            const int ry = min(max(my - 1 + j, 0), a.loH - 1);
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const int rx = min(max(mx - 1 + i, 0), a.loW - 1);
                                const float* pp = lo + ((size_t)ry * a.loW + rx) * a.loLd;
                                const float* ww = we + (size_t)(j * 3 + i) * a.C1;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const pf_f32x4 v4 = *reinterpret_cast<const pf_f32x4*>(pp + 4 * h);
                                    const pf_f32x4 w4 = *reinterpret_cast<const pf_f32x4*>(ww + 4 * h);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[4 * h + e] = fmaf(w4[e], v4[e], o[4 * h + e]);
                                }
                            }
                        }
                    } else {
                        const int C2 = a.inC - a.C1;
                        const float* wd = a.dw_w2 + (kelem - a.C1);
                        const float* sk = a.up_skip + (size_t)xb[u] * H * W * a.skipLd + (kelem - a.C1);
#pragma unroll 1
                        for (int k1 = 0; k1 < 3; ++k1) {
                            const int yy = y - 1 + k1;
                            if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
                            for (int k2 = 0; k2 < 3; ++k2) {
                                const int xx = x - 1 + k2;
                                if ((unsigned)xx >= (unsigned)W) continue;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const pf_f32x4 v4 = *reinterpret_cast<const pf_f32x4*>(sk + ((size_t)yy * W + xx) * a.skipLd + 4 * h);
                                    const pf_f32x4 w4 = *reinterpret_cast<const pf_f32x4*>(wd + (size_t)(k1 * 3 + k2) * C2 + 4 * h);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[4 * h + e] = fmaf(w4[e], v4[e], o[4 * h + e]);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) xreg[0][u][e >> 2][e & 3] = o[e];
            }
        } else {
#pragma unroll
        for (int u = 0; u < XUNITS; ++u) {
            const int iy = KS == 1 ? xiy0[u] : xiy0[u] + ky * a.dil;
            const int ix = KS == 1 ? xix0[u] : xix0[u] + kx * a.dil;
            const bool pok = xvalid[u] && (KS == 1 || ((unsigned)iy < (unsigned)a.inH && (unsigned)ix < (unsigned)a.inW));
            const size_t off = KS == 1 ? (size_t)iy * a.inLd + kelem : ((size_t)(xb[u] * a.inH + iy) * a.inW + ix) * a.inLd + kelem;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                if (pok && kelem + 4 * h < a.inC) {
                    v = *reinterpret_cast<const pf_f32x4*>(in + off + 4 * h);
                    if (a.gate) {
                        const float* g = a.gate + (size_t)xb[u] * a.inC + kelem + 4 * h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= g[e];
                    }
                }
                xreg[0][u][h] = v;
            }
        }
        }
        load_w(tap, cb, stage);
    };
    // plain pointwise path: this thread's pixel units of K step cb, requested and nothing else (out-of-range units read element 0
    // and are zeroed when they are split; the SE gate is applied there too: multiplying on the spot put an s_waitcnt vmcnt(0)
    // behind each of a K step's loads)
    auto load_x = [&](int cb, pf_f32x4 (&xr)[XUNITS][2]) {
        const int kelem = cb * 32 + xc * 8;
#pragma unroll
        for (int u = 0; u < XUNITS; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = xvalid[u] && kelem + 4 * h < a.inC;
                xr[u][h] = *reinterpret_cast<const pf_f32x4*>(in + (ok ? (size_t)xiy0[u] * a.inLd + kelem + 4 * h : (size_t)0));
            }
    };
    auto store_tile = [&](int stage, int cb, const pf_f32x4 (&xr)[XUNITS][2]) {
        unsigned char* xh = smem + stage * STAGE_BYTES;
        unsigned char* xl = xh + PLANE_X;
#pragma unroll
        for (int u = 0; u < XUNITS; ++u) {
            pf_f32x4 xv[2] = {xr[u][0], xr[u][1]};
            if constexpr (GATED) {
                // the gate is applied INSIDE each branch: a value loaded in the fall-back branch and used behind the join would make
                // the compiler wait for vmcnt(0) on every path, i.e. for the look-ahead loads too
                const int kelem = cb * 32 + xc * 8;
                if (gate_lds) {
                    xv[0] *= *reinterpret_cast<const pf_f32x4*>(sgate + kelem);
                    xv[1] *= *reinterpret_cast<const pf_f32x4*>(sgate + kelem + 4);
                } else if (a.gate) {
                    const float* g = a.gate + (size_t)xb[u] * a.inC + kelem;
                    if (xvalid[u] && kelem < a.inC) xv[0] *= *reinterpret_cast<const pf_f32x4*>(g);
                    if (xvalid[u] && kelem + 4 < a.inC) xv[1] *= *reinterpret_cast<const pf_f32x4*>(g + 4);
                }
            }
            pf_half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = xv[e >> 2][e & 3];
                if constexpr (PW2) { if (!(xvalid[u] && cb * 32 + xc * 8 + (e & 4) < a.inC)) v = 0.f; }
                const pf_half hv = (pf_half)v;
                hi[e] = hv;
                lo[e] = pf_split_lo(v, hv);
                amax = pf_amax(amax, v);
            }
            const int off = pf_lds_chunk_off(xrow0 + XROWSTEP * u, xc);
            *reinterpret_cast<pf_half8*>(xh + off) = hi;
            *reinterpret_cast<pf_half8*>(xl + off) = lo;
        }
    };

    pf_f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    auto mma_stage = [&](int cur) {
        const unsigned char* xh = smem + cur * STAGE_BYTES;
        const unsigned char* xl = xh + PLANE_X;
        const unsigned char* wh = xl + PLANE_X;
        const unsigned char* wl = wh + PLANE_W;
        if constexpr (MT == 2 && NT >= 4) {
            // wide-N tiles: the two pixel fragments stay live and the weight fragments come one 16-channel tile at a time -- 24
            // fragment registers instead of 8 NT + 8 (same products in the same order per accumulator)
            if (!(pf_dbg(a) & 16)) {
                pf_half8 xhf[MT], xlf[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int off = pf_lds_chunk_off(wm * WM + i * 16 + frow, fchunk);
                    xhf[i] = *reinterpret_cast<const pf_half8*>(xh + off);
                    xlf[i] = *reinterpret_cast<const pf_half8*>(xl + off);
                }
                // weight fragments of tile j + 1 are requested in front of tile j's MFMAs and no further ahead (the compiler fence):
                // left alone, the scheduler of the unrolled instances hoists every tile's reads and spills
                pf_half8 wq[2][2];
                {
                    const int off = pf_lds_chunk_off(wn * WN + frow, fchunk);
                    wq[0][0] = *reinterpret_cast<const pf_half8*>(wh + off);
                    wq[0][1] = *reinterpret_cast<const pf_half8*>(wl + off);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (j + 1 < NT) {
                        const int off = pf_lds_chunk_off(wn * WN + (j + 1) * 16 + frow, fchunk);
                        wq[(j + 1) & 1][0] = *reinterpret_cast<const pf_half8*>(wh + off);
                        wq[(j + 1) & 1][1] = *reinterpret_cast<const pf_half8*>(wl + off);
                    }
                    const pf_half8 whj = wq[j & 1][0], wlj = wq[j & 1][1];
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(wlj, xhf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whj, xlf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whj, xhf[i], acc[j][i]);
                    asm volatile("" ::: "memory");
                }
            }
        } else {
        pf_half8 whf[NT], wlf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int off = pf_lds_chunk_off(wn * WN + j * 16 + frow, fchunk);
            whf[j] = *reinterpret_cast<const pf_half8*>(wh + off);
            wlf[j] = *reinterpret_cast<const pf_half8*>(wl + off);
        }
        if (!(pf_dbg(a) & 16))
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int off = pf_lds_chunk_off(wm * WM + i * 16 + frow, fchunk);
            const pf_half8 xhf = *reinterpret_cast<const pf_half8*>(xh + off);
            const pf_half8 xlf = *reinterpret_cast<const pf_half8*>(xl + off);
            // small terms first, the dominant hi*hi term last
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(wlf[j], xhf, acc[j][i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(whf[j], xlf, acc[j][i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(whf[j], xhf, acc[j][i]);
        }
        }
    };
    if constexpr (PW2) {
        // Plain pointwise convs.  Same schedule as the general loop below (operands of step kt + 1 requested before step kt's MFMAs);
        // the pixel loads are unconditional (masked when split) and the SE gate comes from LDS, so nothing waits between a step's
        // requests.  Two steps of look-ahead were tried this round (second register set): asm-issued loads are unsafe -- the
        // register allocator copies / reuses destination registers of loads it cannot see -- and compiler-visible ones get an
        // s_waitcnt vmcnt(0) at the loop head because the in-flight set is loop-carried.
        load_w(0, 0, 0);
        load_x(0, xreg[0]);
        if constexpr (GATED) {
            if (gate_lds) {                                         // behind the first operand requests: its round trip overlaps theirs
                const float* g = a.gate + (size_t)(m0 / OHW) * a.inC;
                for (int i = t; i < a.Cpad; i += NTHR) sgate[i] = i < a.inC ? g[i] : 0.f;
                __syncthreads();
            }
        }
        if constexpr (NK > 0) {
            // One base pointer per pixel unit / weight slot for ALL steps, the step in the instruction's immediate offset: computed per
            // step, the unrolled loop's 30 x 5 addresses are hoisted to its head and spill.  Rows / channels outside the tensor read
            // inside it or at most 124 bytes behind it (the arena carries that slack) and are zeroed when they are split.
            const float* xbase[XUNITS];
#pragma unroll
            for (int u = 0; u < XUNITS; ++u) xbase[u] = in + (xvalid[u] ? (size_t)xiy0[u] * a.inLd : (size_t)0) + xc * 8;
            const unsigned char* wsrc[WCHUNKS];
#pragma unroll
            for (int c = 0; c < WCHUNKS; ++c) {
                const int sl = t + NTHR * c;
                const int plane = sl >= BN * 4 ? 1 : 0;
                const int r = (sl - plane * BN * 4) >> 2;
                const int row = r < BN ? r : BN - 1;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                wsrc[c] = wt + (size_t)min(n0 + row, a.Npad - 1) * wrow_bytes + plane * 64 + chunk * 16;
            }
            auto load_x_at = [&](auto cb_tag, pf_f32x4 (&xr)[XUNITS][2]) {
                constexpr int cb = decltype(cb_tag)::value;
#pragma unroll
                for (int u = 0; u < XUNITS; ++u) {
                    xr[u][0] = *reinterpret_cast<const pf_f32x4*>(xbase[u] + cb * 32);
                    xr[u][1] = *reinterpret_cast<const pf_f32x4*>(xbase[u] + cb * 32 + 4);
                }
            };
            auto load_w_at = [&](auto cb_tag, int stage) {
                constexpr int cb = decltype(cb_tag)::value;
                unsigned char* wdst = smem + stage * STAGE_BYTES + 2 * PLANE_X;
#pragma unroll
                for (int c = 0; c < WCHUNKS; ++c)
                    if (t + NTHR * c < BN * 8) pf_glds16_raw_off<cb * 128>(wsrc[c], wdst + (t + NTHR * c) * 16);
            };
            // Unrolled: the pixel operands of steps kt + 1 AND kt + 2 are in flight (two register sets, no loop-carried value, so the
            // compiler's own vmcnt counting is exact) while step kt's MFMAs run; the weights of step kt + 1 by asm-issued LDS-DMA,
            // requested BEFORE the newest pixels: vmcnt retires in order, so "all but the 2 XUNITS youngest" at the barrier = the
            // weights of the next step have landed, the newest pixels have not.  With one step of look-ahead and __syncthreads()
            // (which drains vmcnt) a K step of conv1x1 960 -> 160 was 8.3 k cycles against 1.9 k of MFMA issue.
            if constexpr (NK > 1) load_x_at(std::integral_constant<int, 1>{}, xreg[1]);
            store_tile(0, 0, xreg[0]);
            if constexpr (NK > 1) pf_wait_vm_barrier<2 * XUNITS>(); else pf_wait_vm_barrier<0>();
            pf_sched_fence();
            pf_static_for<NK>([&](auto kt_tag) {
                constexpr int kt = decltype(kt_tag)::value;
                constexpr int cur = kt & 1;
                if constexpr (kt + 1 < NK) { if (!(pf_dbg(a) & 256)) load_w_at(std::integral_constant<int, kt + 1>{}, cur ^ 1); }
                if constexpr (kt + 2 < NK) { if (!(pf_dbg(a) & 256)) load_x_at(std::integral_constant<int, kt + 2>{}, xreg[cur]); }
                mma_stage(cur);
                if constexpr (kt + 1 < NK) { if (!(pf_dbg(a) & 512)) store_tile(cur ^ 1, kt + 1, xreg[cur ^ 1]); }
                pf_pin(amax);       // the range guard's running maximum is due NOW: left alone, the compiler keeps every step's eight
                                    // values (in scratch) and folds them at the end of the unrolled loop
                if constexpr (kt + 2 < NK) pf_wait_vm_barrier<2 * XUNITS>(); else pf_wait_vm_barrier<0>();
                pf_sched_fence();
            });
        } else {
        store_tile(0, 0, xreg[0]);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            if (more && !(pf_dbg(a) & 256)) { load_x(kt + 1, xreg[0]); load_w(0, kt + 1, cur ^ 1); }
            mma_stage(cur);
            if (more && !(pf_dbg(a) & 512)) store_tile(cur ^ 1, kt + 1, xreg[0]);
            __syncthreads();
        }
        }
    } else {
    int tap = 0, cb = 0;
    load_tile(tap, cb, 0);
    store_tile(0, 0, xreg[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            if (++tap == taps) { tap = 0; ++cb; }
            if (!(pf_dbg(a) & 256)) load_tile(tap, cb, cur ^ 1);
        }
        mma_stage(cur);
        if (more && !(pf_dbg(a) & 512)) store_tile(cur ^ 1, cb, xreg[0]);
        __syncthreads();
    }
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
    if constexpr (EPI_K < 0) {
        conv_gemm_argmax_epilogue<BM, BN, WARPS_M, WARPS_N>(a, acc, m0, n0, wm, wn, lane, OHW, a.acc_scale);
    } else if constexpr (EPI_K != 0) {
        static_assert(KS == 1 && STAGE == 0, "fused depthwise epilogue: pointwise expand only");
        static_assert(2 * STAGE_BYTES >= (BM * (BN + 4) + (NTHR / BN) * 4 * BN) * 4, "E tile must fit the staging LDS");
        expdw_epilogue<BM, BN, WARPS_M, WARPS_N, EPI_K, EPI_DIL, EPI_W>(a, acc, smem, m0, n0, wm, wn, t, M);   // loop ended on a barrier
    } else {
        conv_gemm_epilogue<float, BM, BN, WARPS_M, WARPS_N>(a, acc, m0, n0, wm, wn, lane, M, OHW, a.acc_scale);
    }
}

// ---- expand 1x1 -> depthwise k x k (+ SE squeeze) on 32 x 32 maps: one workgroup = one image x 16 expanded channels ----
// The 16 x 16 variant above (expdw_epilogue) owns whole images inside a 256-pixel GEMM tile; a 32 x 32 image is 1024
// pixels, too many rows for the GEMM's LDS staging.  With <= 64 input channels (stage 2 of the Student: 40 -> 120) the
// expand GEMM is tiny, so it skips LDS altogether: every wave loads the pixel fragments of its 128 pixels straight from
// global memory (the image's 160 KB of input is re-read by the 8 channel tiles out of L2), splits them and runs
// 3 MFMAs per 16 x 16 tile; the activated 32 x 32 x 16 tile (67 KB, rows padded so four rows land in different banks)
// lives in LDS and the depthwise conv reads it there, thread = (channel, image row), as in the 16 x 16 kernel.
// ACT >= 0: the activation at compile time (as in k_mbconv.h: the run-time switch put two scalar branches behind every 16-pixel tile of
// the expand loop and the tiles could not overlap).
template <int K, int DIL, int ACT = -1>
__global__ __launch_bounds__(512, 4) void expdw_image_kernel(ConvGemmArgs a) {
    unsigned amax = 0;                                 // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    constexpr int HW = 32, CB = 16;
    constexpr int RS = HW * CB + 16;            // floats per image row in LDS
    constexpr int PAD = DIL * (K - 1) / 2;
    constexpr int MAXKS = 2;                    // input channels <= 64
    __shared__ __attribute__((aligned(16))) float es[HW * RS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x, n0 = blockIdx.y * CB;
    const bool track = blockIdx.y == 0;
    const int frow = lane & 15, kg = lane >> 4;
    const int ksteps = a.Cpad / 32;
    const float* __restrict__ in = static_cast<const float*>(a.in) + (size_t)b * HW * HW * a.inLd;
    const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);
    // weight fragments of this channel tile (rows n0 + frow), all K steps
    pf_half8 whf[MAXKS], wlf[MAXKS];
    {
        const int row = min(n0 + frow, a.Npad - 1);
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            whf[ks] = pf_half8{0, 0, 0, 0, 0, 0, 0, 0};
            wlf[ks] = whf[ks];
            if (ks < ksteps) {
                const unsigned char* p = wt + ((size_t)row * ksteps + ks) * 128 + kg * 16;
                whf[ks] = *reinterpret_cast<const pf_half8*>(p);
                wlf[ks] = *reinterpret_cast<const pf_half8*>(p + 64);
            }
        }
    }
    const int cch = 4 * kg;                     // accumulator layout: channels cch..cch+3 of pixel frow
    pf_f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (n0 + cch + r < a.Npad) ? a.bias[n0 + cch + r] : 0.f;
    // ---- expand: 8 waves x 8 tiles of 16 pixels ----------------------------------------------------------------
#pragma unroll 2
    for (int mt = 0; mt < 8; ++mt) {
        const int p = wave * 128 + mt * 16 + frow;
        const float* px = in + (size_t)p * a.inLd;
        pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < ksteps) {
                const int c = ks * 32 + kg * 8;
                pf_f32x4 v0 = pf_f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (c < a.inC) {                // inC % 8 == 0
                    v0 = *reinterpret_cast<const pf_f32x4*>(px + c);
                    v1 = *reinterpret_cast<const pf_f32x4*>(px + c + 4);
                }
                pf_half8 xh, xl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? v0[e & 3] : v1[e & 3];
                    const pf_half hv = (pf_half)v;
                    xh[e] = hv;
                    xl[e] = pf_split_lo(v, hv);
                }
                if (track) {                    // (wave-uniform) the eight channel tiles of an image split the SAME input: one of them reports its range
#pragma unroll
                    for (int e = 0; e < 8; ++e) amax = pf_amax(amax, e < 4 ? v0[e & 3] : v1[e & 3]);
                }
                acc = pf_mfma_16x16x32_f16(wlf[ks], xh, acc);
                acc = pf_mfma_16x16x32_f16(whf[ks], xl, acc);
                acc = pf_mfma_16x16x32_f16(whf[ks], xh, acc);
            }
        }
        pf_f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[r], a.acc_scale, bv[r]);
        if constexpr (ACT >= 0) {
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) v[q_] = pf_act_c<ACT>(v[q_]);
        } else pf_act_rh<4>(v, a.act);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n0 + cch + r >= a.N) v[r] = 0.f;
        *reinterpret_cast<pf_f32x4*>(es + (p >> 5) * RS + (p & 31) * CB + cch) = v;
    }
    // depthwise filters of this thread's channel (requested before the barrier)
    const int c = t & 15, y = t >> 4;
    const int n = n0 + c;
    const bool cok = n < a.N;
    float wk[K * K];
#pragma unroll
    for (int k = 0; k < K * K; ++k) wk[k] = cok ? a.dw_w2[(size_t)k * a.N + n] : 0.f;
    const float bd = cok ? a.dw_b[n] : 0.f;
    __syncthreads();
    // ---- depthwise: thread = (channel c, image row y) -------------------------------------------------------------
    float o[HW];
#pragma unroll
    for (int x = 0; x < HW; ++x) o[x] = bd;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int yy = y + ky * DIL - PAD;
        if ((unsigned)yy >= (unsigned)HW) continue;
        const float* erow = es + yy * RS + c;
        float iv[HW];
#pragma unroll
        for (int x = 0; x < HW; ++x) iv[x] = erow[x * CB];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float w = wk[ky * K + kx];
#pragma unroll
            for (int x = 0; x < HW; ++x) {
                const int xx = x + kx * DIL - PAD;           // compile-time register index
                if (xx >= 0 && xx < HW) o[x] = fmaf(w, iv[xx], o[x]);
            }
        }
    }
    if constexpr (ACT >= 0) {
#pragma unroll
        for (int q_ = 0; q_ < HW; ++q_) o[q_] = pf_act_c<ACT>(o[q_]);
    } else pf_act_rh<HW>(o, a.act);
    float rs = 0.f;
    if (cok) {
        float* out = static_cast<float*>(a.out) + ((size_t)b * HW * HW + (size_t)y * HW) * a.outLd + n;
#pragma unroll
        for (int x = 0; x < HW; ++x) {
            out[(size_t)x * a.outLd] = o[x];
            rs += o[x];
        }
    }
    if (track) pf_amax_commit(a.range_slot, amax, amax_seen);
    if (a.gap_out) {
        __syncthreads();                        // E is dead: its LDS becomes the row-sum scratch
        es[y * CB + c] = rs;
        __syncthreads();
        if (t < CB && n0 + t < a.N) {
            float tot = 0.f;
#pragma unroll
            for (int r = 0; r < HW; ++r) tot += es[r * CB + t];
            a.gap_out[(size_t)b * a.N + n0 + t] = tot / (float)(HW * HW);
        }
    }
}

// ---- same for the stride-2 block that enters stage 2 (64 x 64 x 24 -> expand 72 -> depthwise 5x5 / 2 -> 32 x 32) -------------
// One workgroup = one image x 16 expanded channels, looping over the four 16 x 16 output quadrants: per quadrant the
// expand conv is evaluated on the 35 x 35 input pixels the quadrant's windows cover (1.2x recompute, input from L2,
// pixels outside the image forced to 0 = the depthwise conv's zero padding), parked in LDS (78 KB) and consumed by the
// strided depthwise conv, thread = (channel, output row, half row).  The SE squeeze is complete per workgroup because it
// visits all four quadrants.  Input channels <= 32 (one K step).
template <int K, int ACT = -1>
__global__ __launch_bounds__(512, 4) void expdw_image_s2_kernel(ConvGemmArgs a) {
    unsigned amax = 0;                                 // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    constexpr int IN = 64, OUT = 32, Q = 16, CB = 16;    // input / output size, quadrant size, channels per workgroup
    constexpr int PAD = (K - 1) / 2;
    constexpr int R = (Q - 1) * 2 + K;                   // 35: input rows / columns a quadrant needs
    constexpr int RS = R * CB + 8;                       // floats per region row: 2 * RS = 48 (mod 64) -> 4 output rows, 4 bank groups
    constexpr int NPX = R * R, MTILES = (NPX + 15) / 16;
    constexpr int OX = Q / 2, SPAN = (OX - 1) * 2 + K;   // outputs per thread along x, input span they need
    __shared__ __attribute__((aligned(16))) float es[R * RS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x, n0 = blockIdx.y * CB;
    const bool track = blockIdx.y == 0;
    const int frow = lane & 15, kg = lane >> 4;
    const float* __restrict__ in = static_cast<const float*>(a.in) + (size_t)b * IN * IN * a.inLd;
    const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);
    const int wrow = min(n0 + frow, a.Npad - 1);
    const pf_half8 whf = *reinterpret_cast<const pf_half8*>(wt + (size_t)wrow * 128 + kg * 16);
    const pf_half8 wlf = *reinterpret_cast<const pf_half8*>(wt + (size_t)wrow * 128 + 64 + kg * 16);
    const int cch = 4 * kg;
    pf_f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (n0 + cch + r < a.Npad) ? a.bias[n0 + cch + r] : 0.f;
    const int c = t & 15, y = (t >> 4) & 15, xh = t >> 8;
    const int n = n0 + c;
    const bool cok = n < a.N;
    float wk[K * K];
#pragma unroll
    for (int k = 0; k < K * K; ++k) wk[k] = cok ? a.dw_w2[(size_t)k * a.N + n] : 0.f;
    const float bd = cok ? a.dw_b[n] : 0.f;
    float rs = 0.f;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int oy0 = (q >> 1) * Q, ox0 = (q & 1) * Q;
        const int iy0 = oy0 * 2 - PAD, ix0 = ox0 * 2 - PAD;
        // ---- expand on the quadrant's input region ---------------------------------------------------------------
        for (int mt = wave; mt < MTILES; mt += 8) {
            const int p = mt * 16 + frow;
            const int ry = p / R, rx = p - ry * R;
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool ok = p < NPX && (unsigned)iy < (unsigned)IN && (unsigned)ix < (unsigned)IN;
            pf_f32x4 v0 = pf_f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (ok && kg * 8 < a.inC) {
                const float* px = in + ((size_t)iy * IN + ix) * a.inLd + kg * 8;
                v0 = *reinterpret_cast<const pf_f32x4*>(px);
                v1 = *reinterpret_cast<const pf_f32x4*>(px + 4);
            }
            pf_half8 xhf, xlf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = e < 4 ? v0[e & 3] : v1[e & 3];
                const pf_half hv = (pf_half)v;
                xhf[e] = hv;
                xlf[e] = pf_split_lo(v, hv);
            }
            if (track) {                        // (wave-uniform) one channel tile per image reports the range of the shared input
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = pf_amax(amax, e < 4 ? v0[e & 3] : v1[e & 3]);
            }
            pf_f32x4 acc = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            acc = pf_mfma_16x16x32_f16(wlf, xhf, acc);
            acc = pf_mfma_16x16x32_f16(whf, xlf, acc);
            acc = pf_mfma_16x16x32_f16(whf, xhf, acc);
            pf_f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[r], a.acc_scale, bv[r]);
            if constexpr (ACT >= 0) {
#pragma unroll
                for (int q_ = 0; q_ < 4; ++q_) v[q_] = pf_act_c<ACT>(v[q_]);
            } else pf_act_rh<4>(v, a.act);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!ok || n0 + cch + r >= a.N) v[r] = 0.f;        // zero padding of the EXPANDED map / padding channels
            if (p < NPX) *reinterpret_cast<pf_f32x4*>(es + ry * RS + rx * CB + cch) = v;
        }
        __syncthreads();
        // ---- depthwise K x K / 2: thread = (channel, output row y, half row xh) ----------------------------------------
        float o[OX];
#pragma unroll
        for (int x = 0; x < OX; ++x) o[x] = bd;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const float* erow = es + (2 * y + ky) * RS + (2 * xh * OX) * CB + c;
            float iv[SPAN];
#pragma unroll
            for (int i = 0; i < SPAN; ++i) iv[i] = erow[i * CB];
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int x = 0; x < OX; ++x) o[x] = fmaf(wk[ky * K + kx], iv[2 * x + kx], o[x]);
        }
        if constexpr (ACT >= 0) {
#pragma unroll
            for (int q_ = 0; q_ < OX; ++q_) o[q_] = pf_act_c<ACT>(o[q_]);
        } else pf_act_rh<OX>(o, a.act);
        if (cok) {
            float* out = static_cast<float*>(a.out) + ((size_t)b * OUT * OUT + (size_t)(oy0 + y) * OUT + ox0 + xh * OX) * a.outLd + n;
#pragma unroll
            for (int x = 0; x < OX; ++x) {
                out[(size_t)x * a.outLd] = o[x];
                rs += o[x];
            }
        }
        __syncthreads();                        // the next quadrant overwrites the region
    }
    if (track) pf_amax_commit(a.range_slot, amax, amax_seen);
    if (a.gap_out) {
        es[(t >> 4) * CB + c] = rs;             // 32 partial sums per channel
        __syncthreads();
        if (t < CB && n0 + t < a.N) {
            float tot = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) tot += es[r * CB + t];
            a.gap_out[(size_t)b * a.N + n0 + t] = tot / (float)(OUT * OUT);
        }
    }
}

// ---- 3x3 stride-1 convolution with the input tile (plus halo) resident in LDS -------------------------------
// The generic kernel above is an im2col pipeline: every tap re-fetches, re-splits and re-writes the same 128
// pixels x 32 channels (9x per channel chunk).  Here a workgroup's 128 output pixels are BM / W whole image rows;
// for each 32-channel chunk the (rows + 2) x (W + 2) input patch is fetched ONCE (register-prefetched one chunk
// ahead), split to hi/lo and parked in LDS, zero padding included; the nine taps then read their pixel fragments
// at shifted row offsets (conflict-free for any shift with the same chunk rotation) and only the weights stream
// per tap (LDS-DMA, two stages).  Activation fetches, conversions and LDS writes drop ~4x (halo overhead 2.06x
// at W = 64); the matrix-core work and the epilogue are unchanged.  Host guarantees: pad = dil = stride = 1,
// W in {16, 32, 64}, (H * W) % 128 == 0, no input gate.
template <int BN, int WARPS_M, int WARPS_N, int BM = 128>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64, WARPS_M * WARPS_N / 2) void conv3x3_halo_split_kernel(ConvGemmArgs a) {
    constexpr int NTHR = WARPS_M * WARPS_N * 64;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    // halo pixels: BM = 128: (2 + 2) x (64 + 2) = 264 at W = 64 (204 / 180 at 32 / 16); BM = 256 (the narrow HRNet variants: twice
    // the MFMAs per barrier, halo overhead 1.55x instead of 2.06x at W = 64): 6 x 66 = 396 (340 / 324 at 32 / 16)
    constexpr int MAXHP = BM == 128 ? 272 : 400;
    static_assert(BM == 128 || BM == 256, "tile rows");
    constexpr int XU = (MAXHP * 4 + NTHR - 1) / NTHR;    // (pixel, 8-float unit) pairs per thread
    constexpr int PLANE_X = MAXHP * 64;
    constexpr int WCHUNKS = (BN * 8 + NTHR - 1) / NTHR;
    constexpr int W_BYTES = WCHUNKS * NTHR * 16;
    static_assert(NTHR == 512 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PLANE_X + 2 * W_BYTES];
    unsigned char* xh = smem;
    unsigned char* xl = smem + PLANE_X;
    unsigned char* wbase = smem + 2 * PLANE_X;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    int mtile = blockIdx.x;
    if ((gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware tile order
    const int m0 = mtile * BM;
    const int n0 = blockIdx.y * BN;
    const int W = a.outW, H = a.outH, OHW = H * W;
    const int M = a.B * OHW;
    const int HW2 = W + 2;
    const int TR = BM / W;
    const int HP = (TR + 2) * HW2;
    const int face = m0 / OHW;
    const int y0 = (m0 - face * OHW) / W;
    const float* __restrict__ in = static_cast<const float*>(a.in) + (size_t)face * OHW * a.inLd;
    const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);
    const int cblocks = a.Cpad / 32;
    const size_t wrow_bytes = (size_t)9 * cblocks * 128;

    // this thread's halo units
    int xoff[XU], xhp[XU];
    const int xc = t & 3;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int hp = (t >> 2) + (NTHR / 4) * u;
        xhp[u] = hp < HP ? hp : -1;
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = hp < HP && m0 < M && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        xoff[u] = ok ? (iy * W + ix) * a.inLd + xc * 8 : -1;
    }
    pf_f32x4 xreg[XU][2];
    unsigned amax = 0;                                 // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    auto load_x = [&](int cb) {
#pragma unroll
        for (int u = 0; u < XU; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                if (xoff[u] >= 0 && cb * 32 + xc * 8 + 4 * h < a.inC) v = *reinterpret_cast<const pf_f32x4*>(in + xoff[u] + cb * 32 + 4 * h);
                xreg[u][h] = v;
            }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (xhp[u] < 0) continue;
            pf_half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = xreg[u][e >> 2][e & 3];
                const pf_half hv = (pf_half)v;
                hi[e] = hv;
                lo[e] = pf_split_lo(v, hv);
                amax = pf_amax(amax, v);
            }
            const int off = pf_lds_chunk_off(xhp[u], xc);
            *reinterpret_cast<pf_half8*>(xh + off) = hi;
            *reinterpret_cast<pf_half8*>(xl + off) = lo;
        }
    };
    auto load_w = [&](int tap, int cb, int stage) {
        unsigned char* wdst = wbase + stage * W_BYTES;
#pragma unroll
        for (int c = 0; c < WCHUNKS; ++c) {
            const int sl = t + NTHR * c;
            const int plane = sl >= BN * 4 ? 1 : 0;
            const int r = (sl - plane * BN * 4) >> 2;
            const int row = r < BN ? r : BN - 1;
            const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
            const int n = min(n0 + row, a.Npad - 1);
            pf_glds16(wt + (size_t)n * wrow_bytes + ((size_t)tap * cblocks + cb) * 128 + plane * 64 + chunk * 16, wdst + sl * 16);
        }
    };

    pf_f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;
    int hp0[MT];                                         // halo row of this lane's pixel at tap (0, 0)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int p = wm * WM + i * 16 + frow;
        const int ty = p / W, tx = p - ty * W;
        hp0[i] = ty * HW2 + tx;
    }

    load_x(0);
    load_w(0, 0, 0);
    store_x();
    __syncthreads();
    const int nk = 9 * cblocks;
    int tap = 0, cb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        const bool last_tap = tap == 8;
        if (more && !(pf_dbg(a) & 1)) load_w(last_tap ? 0 : tap + 1, last_tap ? cb + 1 : cb, cur ^ 1);
        if (tap == 0 && cb + 1 < cblocks && !(pf_dbg(a) & 64)) load_x(cb + 1);          // next chunk's patch: nine taps of latency cover
        const unsigned char* wh = wbase + cur * W_BYTES;
        const unsigned char* wl = wh + BN * 64;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int shift = ky * HW2 + kx;
        if (!(pf_dbg(a) & 16)) {
            if constexpr (MT == 2) {
                // pixel fragments of the (two) 16-pixel sub-tiles stay live, weight fragments come one 16-channel tile at a
                // time: 24 fragment registers instead of 40, which is what keeps this kernel out of scratch at 128 VGPRs
                pf_half8 xhf[MT], xlf[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int off = pf_lds_chunk_off(hp0[i] + shift, fchunk);
                    xhf[i] = *reinterpret_cast<const pf_half8*>(xh + off);
                    xlf[i] = *reinterpret_cast<const pf_half8*>(xl + off);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int off = pf_lds_chunk_off(wn * WN + j * 16 + frow, fchunk);
                    const pf_half8 whf = *reinterpret_cast<const pf_half8*>(wh + off);
                    const pf_half8 wlf = *reinterpret_cast<const pf_half8*>(wl + off);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(wlf, xhf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xlf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xhf[i], acc[j][i]);
                }
            } else {
                pf_half8 whf[NT], wlf[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int off = pf_lds_chunk_off(wn * WN + j * 16 + frow, fchunk);
                    whf[j] = *reinterpret_cast<const pf_half8*>(wh + off);
                    wlf[j] = *reinterpret_cast<const pf_half8*>(wl + off);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int off = pf_lds_chunk_off(hp0[i] + shift, fchunk);
                    const pf_half8 xhf = *reinterpret_cast<const pf_half8*>(xh + off);
                    const pf_half8 xlf = *reinterpret_cast<const pf_half8*>(xl + off);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(wlf[j], xhf, acc[j][i]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(whf[j], xlf, acc[j][i]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(whf[j], xhf, acc[j][i]);
                }
            }
        }
        if (last_tap && more && !(pf_dbg(a) & 64)) {
            __syncthreads();                 // every wave is done with this chunk's patch
            store_x();
        }
        if (!(pf_dbg(a) & 128)) __syncthreads();
        if (last_tap) { tap = 0; ++cb; } else ++tap;
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
    conv_gemm_epilogue<float, BM, BN, WARPS_M, WARPS_N>(a, acc, m0, n0, wm, wn, lane, M, OHW, a.acc_scale);
}

// ---- fused DecoderBlock front end with the low-res patch and its filters resident in LDS ------------------------
// Same operator as conv_gemm_split_kernel<..., STAGE = 1> (bilinear x2 upsample + concat + depthwise 3x3 + BN as
// the producer of a pointwise split-precision GEMM).  There every (pixel, 8-channel) unit issues 36 16-byte
// global loads per K step (9 taps x input + position-class filter) and the launch is bound by the L1 / texture
// address path.  Here, per 32-channel chunk of the upsampled half, the workgroup parks the low-res rows it needs
// ((rows/2 + 2) x (W/2 + 2) pixels, border replication materialised) and the 16 x 9 x 32 class filters in LDS
// (register-prefetched / LDS-DMA one chunk ahead) and the producer reads them there; the skip-connection chunks
// (the last one or two) keep the global-load path.  One x stage, one weight stage, two barriers per K step:
//   phase 1: weights of this step by LDS-DMA || produce + split the pixel operand -> LDS
//   phase 2: next chunk's patch / filters -> LDS || MFMAs
// Host guarantees: W in {16, 32, 64} (= 2 x low-res width), (H * W) % 128 == 0, C1 % 32 == 0.
template <int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64, BN >= 256 ? WARPS_M * WARPS_N / 4 : WARPS_M * WARPS_N / 2) void sepup_patch_kernel(ConvGemmArgs a) {
    constexpr int BM = 128;
    constexpr int NTHR = WARPS_M * WARPS_N * 64;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int MAXPP = 102;                           // 3 x 34 low-res pixels at W = 64 (4 x 18 at 32, 6 x 10 at 16)
    constexpr int PU = (MAXPP * 8 + NTHR - 1) / NTHR;    // 16-byte patch units per thread
    constexpr int P_BYTES = MAXPP * 128;
    constexpr int PW_SLOTS = 16 * 9 * 8;                 // 16-byte slots of one chunk's class filters
    constexpr int PW_BYTES = PW_SLOTS * 16;
    constexpr int PLANE_X = BM * 64;
    constexpr int WCHUNKS = BN * 8 / NTHR;
    constexpr int W_BYTES = BN * 128;
    static_assert(NTHR == 512 && (BN * 8) % NTHR == 0 && PW_SLOTS == 2 * NTHR + 128, "tile shape");
    constexpr int MAXC = 640;                            // depthwise biases of every K step (global loads at the head of each step
                                                         // were a full L2 round trip per step: 0.36 ms of "empty" skeleton)
    __shared__ __attribute__((aligned(16))) unsigned char smem[P_BYTES + PW_BYTES + 2 * PLANE_X + W_BYTES + MAXC * 4];
    float* sdwb = reinterpret_cast<float*>(smem + P_BYTES + PW_BYTES + 2 * PLANE_X + W_BYTES);
    float* pl = reinterpret_cast<float*>(smem);                      // [patch pixel][32 ch]
    float* pw = reinterpret_cast<float*>(smem + P_BYTES);            // [class * 9 + tap][32 ch]
    unsigned char* xh = smem + P_BYTES + PW_BYTES;
    unsigned char* xl = xh + PLANE_X;
    unsigned char* wh = xl + PLANE_X;
    unsigned char* wl = wh + BN * 64;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int W = a.outW, H = a.outH, OHW = H * W;
    const int M = a.B * OHW;
    const int TR = BM / W;
    const int PC = a.loW + 2;
    const int PP = (TR / 2 + 2) * PC;
    const int face = m0 / OHW;
    const int y0 = (m0 - face * OHW) / W;
    const int rmin = (y0 >> 1) - 1;                      // low-res row held in patch row 0 (before clamping)
    const float* __restrict__ lo = a.up_lo + (size_t)face * a.loH * a.loW * a.loLd;
    const float* __restrict__ sk = a.up_skip + (size_t)face * OHW * a.skipLd;
    const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);
    const int cblocks = a.Cpad / 32;
    const size_t wrow_bytes = (size_t)cblocks * 128;
    const int lo_chunks = a.C1 / 32;

    // ---- patch units of this thread (border replication applied to the SOURCE coordinates) ---------------
    int poff[PU], pdst[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int q = t + NTHR * u;
        const int pp = q >> 3, c4 = q & 7;
        const bool ok = pp < PP && m0 < M;
        const int pr = pp / PC, pc = pp - pr * PC;
        const int ry = min(max(rmin + pr, 0), a.loH - 1), rx = min(max(pc - 1, 0), a.loW - 1);
        poff[u] = ok ? (ry * a.loW + rx) * a.loLd + c4 * 4 : -1;
        pdst[u] = pp * 32 + c4 * 4;
    }
    pf_f32x4 preg[PU];
    auto load_patch = [&](int cb) {
#pragma unroll
        for (int u = 0; u < PU; ++u) preg[u] = poff[u] >= 0 ? *reinterpret_cast<const pf_f32x4*>(lo + poff[u] + cb * 32) : pf_f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (poff[u] >= 0) *reinterpret_cast<pf_f32x4*>(pl + pdst[u]) = preg[u];
    };
    auto dma_filters = [&](int cb) {                     // [class*9+tap][C1] rows -> [class*9+tap][32] in LDS
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int sl = t + NTHR * c;
            pf_glds16(a.dw_w + (size_t)(sl >> 3) * a.C1 + cb * 32 + (sl & 7) * 4, reinterpret_cast<unsigned char*>(pw) + sl * 16);
        }
        if (t < 128) {                                   // waves 0 and 1: the last 128 slots
            const int sl = 2 * NTHR + t;
            pf_glds16(a.dw_w + (size_t)(sl >> 3) * a.C1 + cb * 32 + (sl & 7) * 4, reinterpret_cast<unsigned char*>(pw) + sl * 16);
        }
    };
    auto dma_weights = [&](int cb) {
#pragma unroll
        for (int c = 0; c < WCHUNKS; ++c) {
            const int sl = t + NTHR * c;
            const int plane = sl >= BN * 4 ? 1 : 0;
            const int row = (sl - plane * BN * 4) >> 2;
            const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
            const int n = min(n0 + row, a.Npad - 1);
            pf_glds16(wt + (size_t)n * wrow_bytes + (size_t)cb * 128 + plane * 64 + chunk * 16, wh + sl * 16);
        }
    };

    // ---- this thread's producer unit: pixel (y, x), channels [cb*32 + xc*8, +8) --------------------------------
    const int xc = t & 3;
    const int prow = t >> 2;
    const bool pvalid = m0 + prow < M;
    const int py = y0 + prow / W, px = prow % W;
    const int ycls = py == 0 ? 0 : (py == H - 1 ? 1 : 2 + (py & 1));
    const int xcls = px == 0 ? 0 : (px == W - 1 ? 1 : 2 + (px & 1));
    const float* fcls = pw + (ycls * 4 + xcls) * 9 * 32 + xc * 8;
    const float* ppix = pl + (((py >> 1) - (y0 >> 1)) * PC + (px >> 1)) * 32 + xc * 8;   // patch pixel of tap (0, 0)
    const int xrow_off = pf_lds_chunk_off(prow, xc);

    pf_f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;

    unsigned amax = 0;                                 // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    for (int i = t; i < MAXC; i += NTHR) sdwb[i] = i < a.inC ? a.dw_b[i] : 0.f;
    if (lo_chunks > 0) {
        load_patch(0);
        dma_filters(0);
        store_patch();
    }
    __syncthreads();
    if (lo_chunks > 1) load_patch(1);
    for (int cb = 0; cb < cblocks; ++cb) {
        // ---- phase 1: weights of this step || produce the pixel operand ----------------------------------------
        if (!(pf_dbg(a) & 1) || cb == 0) dma_weights(cb);
        float o[8];
        const int kelem = cb * 32 + xc * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        if (pvalid && kelem < a.inC) {
            const pf_f32x4 b0 = *reinterpret_cast<const pf_f32x4*>(sdwb + kelem), b1 = *reinterpret_cast<const pf_f32x4*>(sdwb + kelem + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = b0[e]; o[4 + e] = b1[e]; }
            if (cb < lo_chunks) {
#pragma unroll 1
                for (int j = 0; j < ((pf_dbg(a) & 256) ? 1 : 3); ++j)   // one patch row at a time keeps the live LDS reads (and VGPRs) bounded
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float* pp = ppix + (j * PC + i) * 32;
                        const float* ww = fcls + (j * 3 + i) * 32;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const pf_f32x4 v4 = *reinterpret_cast<const pf_f32x4*>(pp + 4 * h);
                            const pf_f32x4 w4 = *reinterpret_cast<const pf_f32x4*>(ww + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[4 * h + e] = fmaf(w4[e], v4[e], o[4 * h + e]);
                        }
                    }
            } else {
                const int C2 = a.inC - a.C1;
                const float* wd = a.dw_w2 + (kelem - a.C1);
                const float* sp = sk + (kelem - a.C1);
#pragma unroll 1
                for (int k1 = 0; k1 < 3; ++k1) {
                    const int yy = py - 1 + k1;
                    if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
                    for (int k2 = 0; k2 < 3; ++k2) {
                        const int xx = px - 1 + k2;
                        if ((unsigned)xx >= (unsigned)W) continue;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const pf_f32x4 v4 = *reinterpret_cast<const pf_f32x4*>(sp + ((size_t)yy * W + xx) * a.skipLd + 4 * h);
                            const pf_f32x4 w4 = *reinterpret_cast<const pf_f32x4*>(wd + (size_t)(k1 * 3 + k2) * C2 + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[4 * h + e] = fmaf(w4[e], v4[e], o[4 * h + e]);
                        }
                    }
                }
            }
        }
        {
            pf_half8 hi, lo8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const pf_half hv = (pf_half)o[e];
                hi[e] = hv;
                lo8[e] = pf_split_lo(o[e], hv);
                amax = pf_amax(amax, o[e]);
            }
            *reinterpret_cast<pf_half8*>(xh + xrow_off) = hi;
            *reinterpret_cast<pf_half8*>(xl + xrow_off) = lo8;
        }
        __syncthreads();
        // ---- phase 2: next chunk's patch and filters || MFMAs -------------------------------------------------------
        if (cb + 1 < lo_chunks && !(pf_dbg(a) & 64)) {
            store_patch();
            dma_filters(cb + 1);
            if (cb + 2 < lo_chunks) load_patch(cb + 2);
        }
        pf_half8 whf[NT], wlf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int off = pf_lds_chunk_off(wn * WN + j * 16 + frow, fchunk);
            whf[j] = *reinterpret_cast<const pf_half8*>(wh + off);
            wlf[j] = *reinterpret_cast<const pf_half8*>(wl + off);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int off = pf_lds_chunk_off(wm * WM + i * 16 + frow, fchunk);
            const pf_half8 xhf = *reinterpret_cast<const pf_half8*>(xh + off);
            const pf_half8 xlf = *reinterpret_cast<const pf_half8*>(xl + off);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(wlf[j], xhf, acc[j][i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(whf[j], xlf, acc[j][i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = pf_mfma_16x16x32_f16(whf[j], xhf, acc[j][i]);
        }
        __syncthreads();
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
    conv_gemm_epilogue<float, BM, BN, WARPS_M, WARPS_N>(a, acc, m0, n0, wm, wn, lane, M, OHW, a.acc_scale);
}
