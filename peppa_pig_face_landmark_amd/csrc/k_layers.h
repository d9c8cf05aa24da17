// Bandwidth-bound layer kernels of the landmark regressor / detector (NHWC, 16-byte vectors).
//
//   stem_conv_kernel      3x3 stride-2 conv on the 3-channel image (u8 NHWC or f32 NCHW input)
//                         -- timm conv_stem (model.py:252-258) / yolov5-face StemBlock.stem_1
//   dw_conv_kernel        depthwise kxk conv + bias + activation (MobileNetV3 blocks,
//                         SeparableConv2d.conv_dw model.py:21-27)
//   upsample_concat_kernel  F.interpolate(x2, bilinear, align_corners=False) + torch.cat
//                         (DecoderBlock.forward model.py:184-189)
//   gap_kernel            global average pool -> f32 [B][C]   (SE, cSE, ASPPPooling model.py:49)
//   fc_kernel             tiny dense layers on pooled vectors (SE / cSE / ASPP pooling branch)
//   scse_kernel           x*cSE + x*sSE                        (SCSEModule.forward model.py:129-130)
//   hm_decode_kernel      final arg-max reduction + offset gather + landmark un-normalisation
//                         (COTRAIN.postp model.py:511-554, face_landmark.py:112-113)
#pragma once
#include "pf_common.h"

// --------------------------------------------------------------------------------------------
struct StemArgs {
    const void* in;       // u8 [B][H][W][3]  or  f32 [B][3][H][W]
    const float* wt;      // [27][CO] f32, index ((ky*3+kx)*3+ci)*CO + co ; 1/255 folded for u8 input
    const float* bias;    // [CO]
    void* out;            // T [B][outH][outW][CO]   (CO = 16 * gridDim.y: one 16-channel group per blockIdx.y)
    int in_f32_nchw;
    int B, inH, inW, outH, outW, outLd, act, CO;
};

template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemArgs a) {
    constexpr int CO = 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = a.B * a.outH * a.outW;
    if (idx >= total) return;
    const int b = idx / (a.outH * a.outW);
    const int rem = idx - b * a.outH * a.outW;
    const int oy = rem / a.outW, ox = rem - oy * a.outW;
    const int g0 = blockIdx.y * CO;   // first output channel of this workgroup's 16-channel group (wave-uniform)
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = a.bias[g0 + c];
    const unsigned char* in8 = static_cast<const unsigned char*>(a.in);
    const float* inf = static_cast<const float*>(a.in);
    const size_t plane = (size_t)a.inH * a.inW;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if ((unsigned)iy >= (unsigned)a.inH) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if ((unsigned)ix >= (unsigned)a.inW) continue;
            float px[3];
            if (a.in_f32_nchw) {
                const size_t o = (size_t)b * 3 * plane + (size_t)iy * a.inW + ix;
                px[0] = inf[o]; px[1] = inf[o + plane]; px[2] = inf[o + 2 * plane];
            } else {
                const size_t o = ((size_t)b * plane + (size_t)iy * a.inW + ix) * 3;
                px[0] = (float)in8[o]; px[1] = (float)in8[o + 1]; px[2] = (float)in8[o + 2];
            }
            const float* w = a.wt + ((ky * 3 + kx) * 3) * a.CO + g0;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int c = 0; c < CO; ++c) acc[c] = fmaf(px[ci], w[ci * a.CO + c], acc[c]);
        }
    }
    T* o = static_cast<T*>(a.out) + (size_t)idx * a.outLd + g0;
    constexpr int VE = PfVec<T>::N;
    mb_act<CO>(acc, a.act);     // stems use hard-swish (Student), SiLU (detector) or ReLU (Teacher)
#pragma unroll
    for (int v = 0; v < CO / VE; ++v) {
        typename PfVec<T>::type pk;
#pragma unroll
        for (int e = 0; e < VE; ++e) pk[e] = (T)acc[v * VE + e];
        pf_stv<T>(o + v * VE, pk);
    }
}

// --------------------------------------------------------------------------------------------
struct DwArgs {
    const void* in;
    const void* wt;     // T [K*K][C]
    const float* bias;  // [C]
    void* out;
    int B, inH, inW, C, inLd, outH, outW, outLd, K, stride, pad, dil, act;
};

template <typename T>
__global__ __launch_bounds__(256) void dw_conv_kernel(DwArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int CV = a.C / VE;
    const long long total = (long long)a.B * a.outH * a.outW * CV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    const long long pix = idx / CV;
    const int ox = (int)(pix % a.outW);
    const long long t2 = pix / a.outW;
    const int oy = (int)(t2 % a.outH);
    const int b = (int)(t2 / a.outH);
    const T* in = static_cast<const T*>(a.in);
    const T* wt = static_cast<const T*>(a.wt);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = a.bias[cv * VE + e];
    for (int ky = 0; ky < a.K; ++ky) {
        const int iy = oy * a.stride - a.pad + ky * a.dil;
        if ((unsigned)iy >= (unsigned)a.inH) continue;
        for (int kx = 0; kx < a.K; ++kx) {
            const int ix = ox * a.stride - a.pad + kx * a.dil;
            if ((unsigned)ix >= (unsigned)a.inW) continue;
            const vec_t x = pf_ldv<T>(in + ((size_t)(b * a.inH + iy) * a.inW + ix) * a.inLd + cv * VE);
            const vec_t w = pf_ldv<T>(wt + (size_t)(ky * a.K + kx) * a.C + cv * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] = fmaf((float)x[e], (float)w[e], acc[e]);
        }
    }
    vec_t o;
    pf_act_n<VE>(acc, a.act);
#pragma unroll
    for (int e = 0; e < VE; ++e) o[e] = (T)acc[e];
    pf_stv<T>(static_cast<T*>(a.out) + (size_t)pix * a.outLd + cv * VE, o);
}

// --------------------------------------------------------------------------------------------
struct UpcatArgs {
    const void* lo;    // T [B][loH][loW][C1]  (upsampled x2, bilinear, half-pixel centres)
    const void* skip;  // T [B][2loH][2loW][C2]
    void* out;         // T [B][2loH][2loW][C1+C2]
    int B, loH, loW, C1, loLd, C2, skipLd, outLd;
};

template <typename T>
__global__ __launch_bounds__(256) void upsample_concat_kernel(UpcatArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int H = 2 * a.loH, W = 2 * a.loW;
    const int CV = (a.C1 + a.C2) / VE;
    const long long total = (long long)a.B * H * W * CV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    const long long pix = idx / CV;
    const int ox = (int)(pix % W);
    const long long t2 = pix / W;
    const int oy = (int)(t2 % H);
    const int b = (int)(t2 / H);
    const int c = cv * VE;
    T* out = static_cast<T*>(a.out) + (size_t)pix * a.outLd + c;
    if (c >= a.C1) {
        pf_stv<T>(out, pf_ldv<T>(static_cast<const T*>(a.skip) + (size_t)pix * a.skipLd + (c - a.C1)));
        return;
    }
    // source index = (dst + 0.5) / 2 - 0.5, clamped at 0 (torch area_pixel_compute_source_index)
    float sy = (oy + 0.5f) * 0.5f - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = (ox + 0.5f) * 0.5f - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < a.loH - 1 ? 1 : 0), x1 = x0 + (x0 < a.loW - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* lo = static_cast<const T*>(a.lo) + (size_t)b * a.loH * a.loW * a.loLd + c;
    const vec_t p00 = pf_ldv<T>(lo + ((size_t)y0 * a.loW + x0) * a.loLd);
    const vec_t p01 = pf_ldv<T>(lo + ((size_t)y0 * a.loW + x1) * a.loLd);
    const vec_t p10 = pf_ldv<T>(lo + ((size_t)y1 * a.loW + x0) * a.loLd);
    const vec_t p11 = pf_ldv<T>(lo + ((size_t)y1 * a.loW + x1) * a.loLd);
    vec_t o;
#pragma unroll
    for (int e = 0; e < VE; ++e)
        o[e] = (T)(hy * (hx * (float)p00[e] + lx * (float)p01[e]) + ly * (hx * (float)p10[e] + lx * (float)p11[e]));
    pf_stv<T>(out, o);
}

// --------------------------------------------------------------------------------------------
struct GapArgs {
    const void* in;  // T [B][HW][ld]
    float* out;      // [B][C] mean over HW
    int B, HW, C, ld;
};

template <typename T>
__global__ __launch_bounds__(256) void gap_kernel(GapArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    __shared__ float red[32][8][VE];
    const int t = threadIdx.x;
    const int cvi = t & 7, prow = t >> 3;
    const int cv = blockIdx.x * 8 + cvi;
    const int b = blockIdx.y;
    const int CV = a.C / VE;
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    if (cv < CV) {
        const T* in = static_cast<const T*>(a.in) + (size_t)b * a.HW * a.ld + cv * VE;
        for (int p = prow; p < a.HW; p += 32) {
            const vec_t x = pf_ldv<T>(in + (size_t)p * a.ld);
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] += (float)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) red[prow][cvi][e] = acc[e];
    __syncthreads();
    if (prow == 0 && cv < CV) {
        const float inv = 1.0f / (float)a.HW;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[r][cvi][e];
            a.out[(size_t)b * a.C + cv * VE + e] = s * inv;
        }
    }
}

// --------------------------------------------------------------------------------------------
struct FcArgs {
    const float* x;      // [B][K]
    const float* wt;     // [K][N]  (transposed so that consecutive threads read consecutive n)
    const float* bias;   // [N] or nullptr
    const float* scale2; // optional second affine: y = act2(scale2*y + shift2)
    const float* shift2;
    float* y;            // [B][N]
    int B, K, N, act, act2;
};

// Block = 64 outputs x 8 batch items; the 4 waves split every 128-wide K chunk four ways (32 k
// each) so four times as many weight loads are in flight per CU; every weight is read once per 8
// items (consecutive lanes -> consecutive n, coalesced); x is staged through LDS and read as float4.
#define PF_FC_BN 64
#define PF_FC_BB 8
#define PF_FC_MAXK 1024   // largest K staged whole (64 KB of LDS)
#define PF_FC_KT 128
// WHOLE_K: the 8 input vectors are staged in LDS once for all of K (K <= PF_FC_MAXK), so the k loop has
// no barrier and the weight loads pipeline freely -- the SE bottleneck FCs (K up to 960, 64-240 blocks in
// flight) were pure load-latency chains with the tile-by-tile version.  Same k -> thread assignment and
// summation order in both variants.
template <bool WHOLE_K>
__global__ __launch_bounds__(256) void fc_kernel(FcArgs a) {
    constexpr int XK = WHOLE_K ? PF_FC_MAXK : PF_FC_KT;
    __shared__ __attribute__((aligned(16))) float smem[PF_FC_BB * XK + 4 * PF_FC_BB * PF_FC_BN];
    float(*xs)[XK] = reinterpret_cast<float(*)[XK]>(smem);
    float* red = smem + PF_FC_BB * XK;   // [4 k-slices][PF_FC_BB items][64 outputs]
    const int t = threadIdx.x;
    const int nl = t & (PF_FC_BN - 1);
    const int ks = t >> 6;
    const int n = blockIdx.x * PF_FC_BN + nl;
    const int b0 = blockIdx.y * PF_FC_BB;
    float acc[PF_FC_BB];
#pragma unroll
    for (int i = 0; i < PF_FC_BB; ++i) acc[i] = 0.f;
    const int kround = (a.K + PF_FC_KT - 1) / PF_FC_KT * PF_FC_KT;
    // one k round of this thread: 32 weights (rows past K: the clamped row's finite weight meets a zero of xs; columns past N: the
    // clamped column's result is dropped at the store) x the 8 staged vectors
    const int nc = min(n, a.N - 1);
    auto load_w = [&](int k0, float (&wv)[32]) {
#pragma unroll
        for (int j = 0; j < 32; ++j) wv[j] = a.wt[(size_t)min(k0 + ks * 32 + j, a.K - 1) * a.N + nc];
    };
    auto mac = [&](int xk0, const float (&wv)[32]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int i = 0; i < PF_FC_BB; ++i) {
                const pf_f32x4 xv = *reinterpret_cast<const pf_f32x4*>(&xs[i][xk0 + ks * 32 + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i] = fmaf(wv[4 * q + j], xv[j], acc[i]);
            }
        }
    };
    if constexpr (WHOLE_K) {
        // The kernel is a chain of load round trips (64-480 workgroups of a few hundred loads): the weights of FOUR k rounds (512 of K)
        // are requested in one go, the first four before the input vectors are even staged, so an SE bottleneck FC (K up to 960) is
        // two round trips instead of nine.  Same k -> thread assignment and order of additions as round by round.
        constexpr int NR = 4;
        float wv[NR][32];
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r * PF_FC_KT < a.K) load_w(r * PF_FC_KT, wv[r]);
        if ((a.K & 3) == 0) {
            // 16-byte loads, ALL of the thread's (up to 8) requested before the first is stored: loads from clamped addresses and a
            // select, no branch around a load -- with one the compiler waits for every load before the next (7 serialized round
            // trips for the K = 960 squeeze vectors, most of the kernel's time)
            const int k4 = kround / 4;
            constexpr int NV = PF_FC_BB * PF_FC_MAXK / 4 / 256;
            pf_f32x4 v[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = t + 256 * j;
                const int bb = i / k4, kk = (i - bb * k4) * 4;
                v[j] = *reinterpret_cast<const pf_f32x4*>(a.x + (size_t)min(b0 + bb, a.B - 1) * a.K + min(kk, a.K - 4));
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = t + 256 * j;
                const int bb = i / k4, kk = (i - bb * k4) * 4;
                if (i < PF_FC_BB * k4)
                    *reinterpret_cast<pf_f32x4*>(&xs[bb][kk]) = (b0 + bb < a.B && kk < a.K) ? v[j] : pf_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            for (int i = t; i < PF_FC_BB * kround; i += 256) {
                const int bb = i / kround, kk = i - bb * kround;
                xs[bb][kk] = (b0 + bb < a.B && kk < a.K) ? a.x[(size_t)(b0 + bb) * a.K + kk] : 0.f;
            }
        }
        __syncthreads();
        for (int kb = 0; kb < a.K; kb += NR * PF_FC_KT) {
            if (kb) {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (kb + r * PF_FC_KT < a.K) load_w(kb + r * PF_FC_KT, wv[r]);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (kb + r * PF_FC_KT < a.K) mac(kb + r * PF_FC_KT, wv[r]);
        }
    } else {
        for (int k0 = 0; k0 < a.K; k0 += PF_FC_KT) {
            for (int i = t; i < PF_FC_BB * PF_FC_KT; i += 256) {
                const int bb = i / PF_FC_KT, kk = i - bb * PF_FC_KT;
                xs[bb][kk] = (b0 + bb < a.B && k0 + kk < a.K) ? a.x[(size_t)(b0 + bb) * a.K + k0 + kk] : 0.f;
            }
            __syncthreads();
            float wv[32];
            load_w(k0, wv);
            mac(0, wv);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < PF_FC_BB; ++i) red[(ks * PF_FC_BB + i) * PF_FC_BN + nl] = acc[i];
    __syncthreads();
    for (int o = t; o < PF_FC_BB * PF_FC_BN; o += 256) {
        const int i = o / PF_FC_BN, nn = o - i * PF_FC_BN;
        const int b = b0 + i, ng = blockIdx.x * PF_FC_BN + nn;
        if (b >= a.B || ng >= a.N) continue;
        float v = (a.bias ? a.bias[ng] : 0.f) + ((red[(0 * PF_FC_BB + i) * PF_FC_BN + nn] + red[(1 * PF_FC_BB + i) * PF_FC_BN + nn]) +
                                                 (red[(2 * PF_FC_BB + i) * PF_FC_BN + nn] + red[(3 * PF_FC_BB + i) * PF_FC_BN + nn]));
        v = pf_act(v, a.act);
        if (a.scale2) v = pf_act(a.scale2[ng] * v + a.shift2[ng], a.act2);
        a.y[(size_t)b * a.N + ng] = v;
    }
}

// --------------------------------------------------------------------------------------------
// Two dependent FCs on pooled vectors in ONE launch (round 6): y = act2(W2 . h + b2), h = act1b(s2 * act1(W1 . x + b1) + t2) -- a
// squeeze-excite gate (timm SE: reduce -> ReLU -> expand -> hard-sigmoid), the cSE gate of SCSE (model.py:117-130) or the pooled
// branch of the ASPP (model.py:46-61, 85-93).  The Student's forward had 20 fc_kernel launches of 5-17 us each, every one a chain of
// launch latency + dependent loads (three times the faces cost 1.25 x: profiles/r06_run1_kernel_table_1lane_f96.json); the hidden
// vector (24 ... 240 floats per face) never needs HBM.
// A workgroup of 1 024 threads owns PF_FC2_FB faces and streams BOTH weight matrices once (16-byte loads, [K][N] rows): thread =
// (4 output columns, k slice); the slices' partial sums meet in LDS and are added in a fixed order (deterministic).  With 4 faces
// per workgroup a launch of 256 faces is 64 workgroups x 16 waves -- the parallelism of the FC launches it replaces -- and reads each
// matrix 64 times from L2 (118 MB for the 960 -> 240 -> 960 pair).
#define PF_FC2_FB 4
#define PF_FC2_MAXK 1088          // staged input / hidden floats per face (K, R <= 960 + chunk padding)
struct Fc2Args {
    const float* x;       // [B][K]
    const float* w1;      // [K][R]
    const float* b1;      // [R] or nullptr
    const float* scale2;  // optional affine + activation behind act1 (ASPP: the post-concat BN of the pooled branch)
    const float* shift2;
    const float* w2;      // [R][N]
    const float* b2;      // [N] or nullptr
    float* y;             // [B][N]
    int B, K, R, N, act1, act1b, act2;
    int nparts;           // x is [B][nparts][K]: partial sums to be added (in order) and scaled by xscale -- the per-tile channel sums the decoder
    float xscale;         // front end leaves behind (k_sepup.h gap_part) instead of a pooled vector; 1, 1.0 otherwise
};

// one FC phase: out[f][n] (LDS, [FB][NOUT]) = sum_k xs[f][k] * wt[k][n] for the workgroup's faces; xs rows are PF_FC2_MAXK floats, zero beyond K
__device__ __forceinline__ void pf_fc2_phase(const float* __restrict__ wt, int K, int NOUT, const float (*xs)[PF_FC2_MAXK], float* red, float (*out)[PF_FC2_MAXK]) {
    const int t = threadIdx.x;
    const int n4 = (NOUT + 3) >> 2;
    int cg = 8;
    while (cg < n4) cg <<= 1;                             // column groups: power of two >= NOUT / 4 (8 ... 256)
    const int ks = min(1024 / cg, 16);                    // k slices: at most 16 (the partial sums are added serially; spare threads idle)
    const int chunk = ((K + ks - 1) / ks + 3) & ~3;       // k per slice, whole groups of four
    const int c = t & (cg - 1), s = t / cg;
    const int col = min(c, n4 - 1) * 4;                   // (columns past NOUT recompute the last group; dropped at the reduction)
    float acc[PF_FC2_FB][4];
#pragma unroll
    for (int f = 0; f < PF_FC2_FB; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[f][e] = 0.f;
    const int k0 = s * chunk;
    if (s < ks && k0 < K) {
        // eight weight rows requested at a time (the loop is a chain of L2 round trips otherwise); rows past K: a finite weight of the
        // last row meets a staged zero
        for (int kk = 0; kk < chunk; kk += 8) {
            pf_f32x4 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w[j] = *reinterpret_cast<const pf_f32x4*>(wt + (size_t)min(k0 + kk + j, K - 1) * NOUT + col);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (kk + 4 * h >= chunk) break;
#pragma unroll
                for (int f = 0; f < PF_FC2_FB; ++f) {
                    const pf_f32x4 xv = *reinterpret_cast<const pf_f32x4*>(&xs[f][k0 + kk + 4 * h]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[f][e] = fmaf(w[4 * h + j][e], xv[j], acc[f][e]);
                }
            }
        }
    }
    // red[slice][column group][face][4]
    if (s < ks) {
#pragma unroll
        for (int f = 0; f < PF_FC2_FB; ++f)
            *reinterpret_cast<pf_f32x4*>(red + ((size_t)(s * cg + c) * PF_FC2_FB + f) * 4) = pf_f32x4{acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
    }
    __syncthreads();
    for (int o = t; o < PF_FC2_FB * NOUT; o += 1024) {
        const int f = o / NOUT, n = o - f * NOUT;
        float v = 0.f;
        for (int q = 0; q < ks; ++q) v += red[((size_t)(q * cg + (n >> 2)) * PF_FC2_FB + f) * 4 + (n & 3)];
        out[f][n] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void fc2_kernel(Fc2Args a) {
    __shared__ __attribute__((aligned(16))) float xs[PF_FC2_FB][PF_FC2_MAXK];
    __shared__ __attribute__((aligned(16))) float hs[PF_FC2_FB][PF_FC2_MAXK];
    __shared__ __attribute__((aligned(16))) float red[1024 * PF_FC2_FB * 4];
    const int t = threadIdx.x;
    const int b0 = blockIdx.x * PF_FC2_FB;
    {
        // the faces' input vectors: every load requested before the first is stored (clamped addresses + a select, no branch around a
        // load); the padding of both staging arrays zeroed alongside
        float v[PF_FC2_FB];
#pragma unroll
        for (int f = 0; f < PF_FC2_FB; ++f) v[f] = a.x[(size_t)min(b0 + f, a.B - 1) * a.nparts * a.K + min(t, a.K - 1)];
        if (a.nparts > 1) {
            for (int p = 1; p < a.nparts; ++p)
#pragma unroll
                for (int f = 0; f < PF_FC2_FB; ++f) v[f] += a.x[((size_t)min(b0 + f, a.B - 1) * a.nparts + p) * a.K + min(t, a.K - 1)];
#pragma unroll
            for (int f = 0; f < PF_FC2_FB; ++f) v[f] *= a.xscale;
        }
#pragma unroll
        for (int f = 0; f < PF_FC2_FB; ++f) {
            xs[f][t] = (b0 + f < a.B && t < a.K) ? v[f] : 0.f;
            hs[f][t] = 0.f;
            if (t < PF_FC2_MAXK - 1024) { xs[f][1024 + t] = 0.f; hs[f][1024 + t] = 0.f; }
        }
    }
    __syncthreads();
    pf_fc2_phase(a.w1, a.K, a.R, xs, red, hs);
    for (int o = t; o < PF_FC2_FB * a.R; o += 1024) {
        const int f = o / a.R, r = o - f * a.R;
        float v = pf_act(hs[f][r] + (a.b1 ? a.b1[r] : 0.f), a.act1);
        if (a.scale2) v = pf_act(a.scale2[r] * v + a.shift2[r], a.act1b);
        hs[f][r] = v;
    }
    __syncthreads();
    pf_fc2_phase(a.w2, a.R, a.N, hs, red, xs);
    for (int o = t; o < PF_FC2_FB * a.N; o += 1024) {
        const int f = o / a.N, n = o - f * a.N;
        if (b0 + f < a.B) a.y[(size_t)(b0 + f) * a.N + n] = pf_act(xs[f][n] + (a.b2 ? a.b2[n] : 0.f), a.act2);
    }
}

// --------------------------------------------------------------------------------------------
struct ScseArgs {
    const void* in;      // T [B][HW][ld]
    void* out;           // T [B][HW][outLd]
    const float* cse;    // [B][C] channel gate (already sigmoid-ed)
    const float* sse_w;  // [C] 1x1 conv to one channel
    float sse_b;
    int B, HW, C, ld, outLd;
};

template <typename T>
__global__ __launch_bounds__(256) void scse_kernel(ScseArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int LPP = a.C / VE;  // lanes per pixel: power of two <= 64 (host checks)
    const int ppb = 256 / LPP;
    const int t = threadIdx.x;
    const int cl = t % LPP;
    const long long pix = (long long)blockIdx.x * ppb + t / LPP;
    const long long total = (long long)a.B * a.HW;
    const bool ok = pix < total;
    const long long pp = ok ? pix : 0;
    const int b = (int)(pp / a.HW);
    const vec_t x = pf_ldv<T>(static_cast<const T*>(a.in) + (size_t)pp * a.ld + cl * VE);
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) dot = fmaf((float)x[e], a.sse_w[cl * VE + e], dot);
    for (int mask = 1; mask < LPP; mask <<= 1) dot += pf_shfl_xor_f32(dot, mask);
    const float sse = 1.f / (1.f + expf(-(dot + a.sse_b)));
    vec_t o;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const float xv = (float)x[e];
        o[e] = (T)(xv * a.cse[(size_t)b * a.C + cl * VE + e] + xv * sse);
    }
    if (ok) pf_stv<T>(static_cast<T*>(a.out) + (size_t)pp * a.outLd + cl * VE, o);
}

// --------------------------------------------------------------------------------------------
struct HmDecodeArgs {
    const float* amax_val;  // [B][P][nslots]
    const int* amax_idx;
    const void* feat;       // T [B][H*W][featLd] : input of the hm 1x1 conv (decx4)
    const float* off_wt;    // [2P][C] f32 : rows P..3P of the hm conv (offset-x then offset-y)
    const float* off_bias;  // [2P]
    const float* crop;      // optional [B][5] = {w_crop, h_crop, x0, y0, add} (face_landmark.py:104)
    float* loc;             // [B][2P] normalised (x0,y0,x1,y1,...)  == ONNX output 0
    float* score;           // [B][P]                                == ONNX output 1
    float* kps;             // optional [B][P][2] frame coordinates (face_landmark.py:112-113)
    int B, P, nslots, H, W, C, featLd;
};

template <typename T>
__global__ __launch_bounds__(256) void hm_decode_kernel(HmDecodeArgs a) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, p) pair handled by this wave
    const bool ok = item < a.B * a.P;
    const int it = ok ? item : 0;
    const int b = it / a.P, p = it - b * a.P;
    float bv = -3.0e38f;
    int bi = 0x7fffffff;
    for (int s = lane; s < a.nslots; s += 64) {
        const float v = a.amax_val[(size_t)it * a.nslots + s];
        const int i = a.amax_idx[(size_t)it * a.nslots + s];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    for (int mask = 1; mask < 64; mask <<= 1) {
        const float ov = pf_shfl_xor_f32(bv, mask);
        const int oi = pf_shfl_xor_i32(bi, mask);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    // A heat-map with no finite maximum (all NaN, or everything below the -3e38 sentinel: f16 overflow, bad weights)
    // never wins a comparison and leaves the sentinel index: report NaN like torch.max (model.py:522) instead of
    // reading the feature row of pixel 0x7fffffff.
    const bool no_max = (unsigned)bi >= (unsigned)(a.H * a.W);
    if (no_max) { bi = 0; bv = __builtin_nanf(""); }
    const T* f = static_cast<const T*>(a.feat) + ((size_t)b * a.H * a.W + bi) * a.featLd;
    float sx = 0.f, sy = 0.f;
    for (int k = lane; k < a.C; k += 64) {
        const float xv = (float)f[k];
        sx = fmaf(a.off_wt[(size_t)p * a.C + k], xv, sx);
        sy = fmaf(a.off_wt[(size_t)(a.P + p) * a.C + k], xv, sy);
    }
    for (int mask = 1; mask < 64; mask <<= 1) {
        sx += pf_shfl_xor_f32(sx, mask);
        sy += pf_shfl_xor_f32(sy, mask);
    }
    if (ok && lane == 0) {
        const float ox = sx + a.off_bias[p], oy = sy + a.off_bias[a.P + p];
        const float lx = no_max ? bv : ((float)(bi % a.W) + ox) / (float)a.W;
        const float ly = no_max ? bv : ((float)(bi / a.W) + oy) / (float)a.H;
        a.loc[(size_t)b * 2 * a.P + 2 * p] = lx;
        a.loc[(size_t)b * 2 * a.P + 2 * p + 1] = ly;
        a.score[(size_t)b * a.P + p] = bv;
        if (a.kps && a.crop) {
            const float* c = a.crop + (size_t)b * 5;
            a.kps[((size_t)b * a.P + p) * 2] = (lx * c[0] + c[2]) - c[4];
            a.kps[((size_t)b * a.P + p) * 2 + 1] = (ly * c[1] + c[3]) - c[4];
        }
    }
}

// --------------------------------------------------------------------------------------------
// Detector-only helpers (yolov5-face: StemBlock max-pool, nearest upsample / channel shuffle copies)
struct PoolArgs {
    const void* in; void* out;
    int B, inH, inW, C, inLd, outH, outW, outLd;
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(PoolArgs a) {  // 2x2 stride 2, ceil_mode
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int CV = a.C / VE;
    const long long total = (long long)a.B * a.outH * a.outW * CV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    const long long pix = idx / CV;
    const int ox = (int)(pix % a.outW);
    const long long t2 = pix / a.outW;
    const int oy = (int)(t2 % a.outH);
    const int b = (int)(t2 / a.outH);
    const T* in = static_cast<const T*>(a.in) + (size_t)b * a.inH * a.inW * a.inLd + cv * VE;
    float m[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) m[e] = -3.0e38f;
    for (int dy = 0; dy < 2; ++dy) {
        const int iy = oy * 2 + dy;
        if (iy >= a.inH) continue;
        for (int dx = 0; dx < 2; ++dx) {
            const int ix = ox * 2 + dx;
            if (ix >= a.inW) continue;
            const vec_t x = pf_ldv<T>(in + ((size_t)iy * a.inW + ix) * a.inLd);
#pragma unroll
            for (int e = 0; e < VE; ++e) m[e] = fmaxf(m[e], (float)x[e]);
        }
    }
    vec_t o;
#pragma unroll
    for (int e = 0; e < VE; ++e) o[e] = (T)m[e];
    pf_stv<T>(static_cast<T*>(a.out) + (size_t)pix * a.outLd + cv * VE, o);
}

struct CopyArgs {
    const void* in; void* out;
    int B, inH, inW, C, inLd, outLd, outCs, up;  // out is (inH*up) x (inW*up); channel c -> c*outCs
};

template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(CopyArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int CV = a.C / VE;
    const int H = a.inH * a.up, W = a.inW * a.up;
    const long long total = (long long)a.B * H * W * CV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    const long long pix = idx / CV;
    const int ox = (int)(pix % W);
    const long long t2 = pix / W;
    const int oy = (int)(t2 % H);
    const int b = (int)(t2 / H);
    const vec_t x = pf_ldv<T>(static_cast<const T*>(a.in) +
                              ((size_t)(b * a.inH + oy / a.up) * a.inW + ox / a.up) * a.inLd + cv * VE);
    T* o = static_cast<T*>(a.out) + (size_t)pix * a.outLd;
    if (a.outCs == 1) {
        pf_stv<T>(o + cv * VE, x);
    } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) o[(size_t)(cv * VE + e) * a.outCs] = x[e];
    }
}

// yolov5-face Detect decode (models/yolo.py Detect.forward, export_cat branch -- third-party, restated):
// sigmoid on cols 0:5 and 15, xy = (2s-0.5+grid)*stride, wh = (2s)^2*anchor,
// landmark cols 5:15 = raw*anchor + grid*stride.  Rows ordered (anchor, y, x) per level.
struct DetDecArgs {
    const void* in;      // T [B][ny][nx][ld], channel = anchor*16 + o
    float* rows;         // [B][nrows_total][16]
    const float* anchors;  // [3][2] (w,h) in pixels
    int B, ny, nx, ld, row0, nrows_total;
    float stride;
};

template <typename T>
__global__ __launch_bounds__(256) void detect_decode_kernel(DetDecArgs a) {
    const long long total = (long long)a.B * 3 * a.ny * a.nx;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % a.nx);
    long long t2 = idx / a.nx;
    const int y = (int)(t2 % a.ny);
    t2 /= a.ny;
    const int an = (int)(t2 % 3);
    const int b = (int)(t2 / 3);
    const T* in = static_cast<const T*>(a.in) + ((size_t)(b * a.ny + y) * a.nx + x) * a.ld + an * 16;
    float v[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) v[o] = (float)in[o];
    const float aw = a.anchors[an * 2], ah = a.anchors[an * 2 + 1];
    const float gx = (float)x, gy = (float)y;
    float r[16];
    const float s0 = pf_act(v[0], PF_ACT_SIGMOID), s1 = pf_act(v[1], PF_ACT_SIGMOID);
    const float s2 = pf_act(v[2], PF_ACT_SIGMOID), s3 = pf_act(v[3], PF_ACT_SIGMOID);
    r[0] = (s0 * 2.f - 0.5f + gx) * a.stride;
    r[1] = (s1 * 2.f - 0.5f + gy) * a.stride;
    r[2] = (s2 * 2.f) * (s2 * 2.f) * aw;
    r[3] = (s3 * 2.f) * (s3 * 2.f) * ah;
    r[4] = pf_act(v[4], PF_ACT_SIGMOID);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        r[5 + 2 * k] = v[5 + 2 * k] * aw + gx * a.stride;
        r[6 + 2 * k] = v[6 + 2 * k] * ah + gy * a.stride;
    }
    r[15] = pf_act(v[15], PF_ACT_SIGMOID);
    float* out = a.rows + ((size_t)b * a.nrows_total + a.row0 + ((size_t)an * a.ny + y) * a.nx + x) * 16;
#pragma unroll
    for (int o = 0; o < 16; o += 4) *reinterpret_cast<pf_f32x4*>(out + o) = pf_f32x4{r[o], r[o + 1], r[o + 2], r[o + 3]};
}

// --------------------------------------------------------------------------------------------
// Depthwise conv, TX outputs per thread along x: input taps are loaded once per row and reused across
// the TX outputs and the K horizontal taps (the per-output variant above issues K*K 16-byte loads per
// output vector; this one (TX-1)*S+(K-1)*DIL+1 per row).
template <typename T, int K, int S, int DIL, int TX>
__global__ __launch_bounds__(256) void dw_conv_tiled_kernel(DwArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int CV = a.C / VE;
    const int xtiles = (a.outW + TX - 1) / TX;
    const long long total = (long long)a.B * a.outH * xtiles * CV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    long long t2 = idx / CV;
    const int xt = (int)(t2 % xtiles);
    t2 /= xtiles;
    const int oy = (int)(t2 % a.outH);
    const int b = (int)(t2 / a.outH);
    const int ox0 = xt * TX;
    const T* in = static_cast<const T*>(a.in) + (size_t)b * a.inH * a.inW * a.inLd + cv * VE;
    const T* wt = static_cast<const T*>(a.wt) + cv * VE;
    float acc[TX][VE];
#pragma unroll
    for (int j = 0; j < TX; ++j)
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[j][e] = a.bias[cv * VE + e];
    const int ix_base = ox0 * S - a.pad;
    constexpr int SPAN = (TX - 1) * S + (K - 1) * DIL + 1;
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - a.pad + ky * DIL;
        if ((unsigned)iy >= (unsigned)a.inH) continue;
        const T* row = in + (size_t)iy * a.inW * a.inLd;
        vec_t w[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) w[kx] = pf_ldv<T>(wt + (size_t)(ky * K + kx) * a.C);
        vec_t x[SPAN];
#pragma unroll
        for (int i = 0; i < SPAN; ++i) {
            const int ix = ix_base + i;
            x[i] = (unsigned)ix < (unsigned)a.inW ? pf_ldv<T>(row + (size_t)ix * a.inLd) : pf_zero_vec<T>();
        }
#pragma unroll
        for (int j = 0; j < TX; ++j)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[j][e] = fmaf((float)x[j * S + kx * DIL][e], (float)w[kx][e], acc[j][e]);
    }
    T* out = static_cast<T*>(a.out) + ((size_t)(b * a.outH + oy) * a.outW) * a.outLd + cv * VE;
#pragma unroll
    for (int j = 0; j < TX; ++j) pf_act_n<VE>(acc[j], a.act);
#pragma unroll
    for (int j = 0; j < TX; ++j) {
        if (ox0 + j < a.outW) {
            vec_t o;
#pragma unroll
            for (int e = 0; e < VE; ++e) o[e] = (T)acc[j][e];
            pf_stv<T>(out + (size_t)(ox0 + j) * a.outLd, o);
        }
    }
}

// --------------------------------------------------------------------------------------------
// HRNet fuse layers (timm HighResolutionModule.forward, Teacher encoder model.py:306-311):
// out = act(a + nearest_upsample(b, 2^shift)), 16-byte channel vectors.
struct AddUpArgs {
    const void* a;   // T [B][H][W][aLd]
    const void* b;   // T [B][H>>shift][W>>shift][bLd]
    void* out;       // T [B][H][W][outLd]
    int B, H, W, C, aLd, bLd, outLd, shift, act;
};

template <typename T>
__global__ __launch_bounds__(256) void add_upsample_kernel(AddUpArgs a) {
    typedef typename PfVec<T>::type vec_t;
    constexpr int VE = PfVec<T>::N;
    const int CV = a.C / VE;
    const long long total = (long long)a.B * a.H * a.W * CV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    const long long pix = idx / CV;
    const int x = (int)(pix % a.W);
    const long long t2 = pix / a.W;
    const int y = (int)(t2 % a.H);
    const int b = (int)(t2 / a.H);
    const int lh = a.H >> a.shift, lw = a.W >> a.shift;
    const vec_t va = pf_ldv<T>(static_cast<const T*>(a.a) + (size_t)pix * a.aLd + cv * VE);
    const vec_t vb = pf_ldv<T>(static_cast<const T*>(a.b) + ((size_t)(b * lh + (y >> a.shift)) * lw + (x >> a.shift)) * a.bLd + cv * VE);
    float sum[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) sum[e] = (float)va[e] + (float)vb[e];
    pf_act_n<VE>(sum, a.act);
    vec_t o;
#pragma unroll
    for (int e = 0; e < VE; ++e) o[e] = (T)sum[e];
    pf_stv<T>(static_cast<T*>(a.out) + (size_t)pix * a.outLd + cv * VE, o);
}

// --------------------------------------------------------------------------------------------
// The whole fuse sum of a HighResolutionModule towards one of its higher-resolution branches in ONE launch (round 4):
//   out = act(y + sum over the lower branches s of nearest_upsample(bn(conv1x1_s(x_s)), 2^shift_s))
// (timm hrnet.py HighResolutionModule.forward / fuse_layers[i][j > i]; Teacher encoder, model.py:306-311).  As separate launches
// the 64 x 64 x 18 branch was read and written once per term (three add_upsample launches of 34 us for 168 MB each, plus three
// small 1x1 convs): here a workgroup owns a 16 x 16 tile of the output, computes the 1x1 convs of the (8 x 8, 4 x 4, 2 x 2) low-
// resolution pixels under it in plain f32 FMAs from LDS (84 pixels x 18 channels x <= 144 deep: nothing next to the tile's
// 40 KB of traffic), and adds them to y on the way through.  y may already hold the module's strided-conv terms (residual
// epilogues of those convs).  Terms are added in branch order, like the launches this replaces.
struct FuseUpArgs {
    const float* y;      // [B][H][W][yLd]
    float* out;          // [B][H][W][outLd]
    int B, H, W, C, Cs, yLd, outLd, act, nsrc;     // C real channels, Cs = C rounded up to 4 (vector padding, written as zeros)
    const float* src[3]; // [B][H >> shift][W >> shift][srcLd]
    const float* wt[3];  // [srcC][Cs]
    const float* bias[3];// [Cs]
    int srcLd[3], srcC[3], shift[3];
};

template <int CAP>       // floats of LDS: weights + source patches + low-resolution results of one tile
__global__ __launch_bounds__(256) void fuse_up_kernel(FuseUpArgs a) {
    constexpr int T = 16, NTHR = 256;
    __shared__ __attribute__((aligned(16))) float smem[CAP];
    const int t = threadIdx.x;
    const int tilesX = (a.W + T - 1) / T, tilesY = (a.H + T - 1) / T;
    const int ntiles = a.B * tilesX * tilesY;
    const int Q = a.Cs / 4;
    // LDS layout: [weights srcC x Cs of every source][per source: patch R x R x srcC, results R x R x Cs] with R = 16 >> shift
    int woff[3], xoff[3], loff[3];
    int off = 0;
    for (int s = 0; s < a.nsrc; ++s) { woff[s] = off; off += a.srcC[s] * a.Cs; }
    for (int s = 0; s < a.nsrc; ++s) {
        const int R = max(1, T >> a.shift[s]);
        xoff[s] = off; off += R * R * a.srcC[s];
        loff[s] = off; off += R * R * a.Cs;
    }
    // the weights once per workgroup: it walks tiles blockIdx.x, + gridDim.x, ... (one per tile, the 20-41 KB of weights doubled the
    // launch's traffic: 132 us per launch against 72 us for the launches this kernel replaces)
    for (int s = 0; s < a.nsrc; ++s) {
        const int nw4 = a.srcC[s] * a.Cs / 4;
        for (int i = t; i < nw4; i += NTHR) *reinterpret_cast<pf_f32x4*>(smem + woff[s] + 4 * i) = *reinterpret_cast<const pf_f32x4*>(a.wt[s] + 4 * i);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (tilesX * tilesY), tt = tile - b * tilesX * tilesY;
        const int ty0 = (tt / tilesX) * T, tx0 = (tt % tilesX) * T;
        const int th = min(T, a.H - ty0), tw = min(T, a.W - tx0);
        int rh[3], rw[3];
        for (int s = 0; s < a.nsrc; ++s) {
            const int sh = a.shift[s], K = a.srcC[s];
            rh[s] = (th + (1 << sh) - 1) >> sh; rw[s] = (tw + (1 << sh) - 1) >> sh;
            const int lh = a.H >> sh, lw = a.W >> sh, k4 = K / 4;
            const float* src = a.src[s] + (size_t)b * lh * lw * a.srcLd[s];
            for (int i = t; i < rh[s] * rw[s] * k4; i += NTHR) {
                const int px = i / k4, kq = i - px * k4;
                const int py = px / rw[s], pxx = px - py * rw[s];
                *reinterpret_cast<pf_f32x4*>(smem + xoff[s] + px * K + 4 * kq) =
                    *reinterpret_cast<const pf_f32x4*>(src + ((size_t)((ty0 >> sh) + py) * lw + (tx0 >> sh) + pxx) * a.srcLd[s] + 4 * kq);
            }
        }
        __syncthreads();                                           // patches (and, first tile, weights) in place
        // 1x1 convs of the low-resolution pixels under the tile: thread = (pixel, four output channels), k ascending
        for (int s = 0; s < a.nsrc; ++s) {
            const int K = a.srcC[s], n = rh[s] * rw[s] * Q;
            for (int i = t; i < n; i += NTHR) {
                const int px = i / Q, q = i - px * Q;
                pf_f32x4 acc = *reinterpret_cast<const pf_f32x4*>(a.bias[s] + 4 * q);
                const float* xv = smem + xoff[s] + px * K;
                const float* wv = smem + woff[s] + 4 * q;
#pragma unroll 4
                for (int k = 0; k < K; ++k) {
                    const float xk = xv[k];
                    const pf_f32x4 w4 = *reinterpret_cast<const pf_f32x4*>(wv + k * a.Cs);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(xk, w4[e], acc[e]);
                }
                *reinterpret_cast<pf_f32x4*>(smem + loff[s] + px * a.Cs + 4 * q) = acc;     // channels >= C: zero weights and bias
            }
        }
        __syncthreads();
        const float* y = a.y + (size_t)b * a.H * a.W * a.yLd;
        float* out = a.out + (size_t)b * a.H * a.W * a.outLd;
        for (int i = t; i < th * tw * Q; i += NTHR) {
            const int p = i / Q, q = i - p * Q;
            const int py = p / tw, px = p - py * tw;
            const size_t pix = (size_t)(ty0 + py) * a.W + tx0 + px;
            const pf_f32x4 v = *reinterpret_cast<const pf_f32x4*>(y + pix * a.yLd + 4 * q);
            float sum[4] = {v[0], v[1], v[2], v[3]};
            for (int s = 0; s < a.nsrc; ++s) {
                const int sh = a.shift[s];
                const pf_f32x4 l = *reinterpret_cast<const pf_f32x4*>(smem + loff[s] + ((py >> sh) * rw[s] + (px >> sh)) * a.Cs + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += l[e];
            }
            pf_act_n<4>(sum, a.act);
            *reinterpret_cast<pf_f32x4*>(out + pix * a.outLd + 4 * q) = pf_f32x4{sum[0], sum[1], sum[2], sum[3]};
        }
        __syncthreads();                                           // the next tile's patches overwrite what this one still reads
    }
}

// --------------------------------------------------------------------------------------------
// Range guard of the split-precision (f32s) convolutions, ALWAYS ON.  Those kernels write every f32 activation as hi + lo
// with hi = f16(v): |v| >= 65504 overflows to inf, and a tensor whose LARGEST magnitude is below ~1e-3 loses its low halves to
// the f16 subnormal range.  Weights are pre-scaled per layer at pack time; activations depend on the data, so every kernel that
// splits keeps the maximum of |v| over what it splits and commits it to its op's slot (pf_amax / pf_amax_commit,
// pf_common.h); at the end of EVERY forward this kernel compares the slots against [2^-10, 6e4], poisons the outputs with NaN
// and records (op, value) on a violation -- no silent inf, no silent garbage, on any call -- and clears the slots for the next one.
struct RangeVerdictArgs {
    unsigned* slots;         // [n_ops][PF_RANGE_SUBSLOTS] words PF_RANGE_STRIDE apart: raw bits of max |v|, 0 = not measured; cleared here.  Behind them: an 8-byte
                             // verdict key (all ones = no violation) and a 4-byte arrival ticket (pf_load_program initialises both)
    int n_ops;
    float lo, hi;            // accepted range of a tensor's max |x|
    int* status;             // [4] = code (0 ok, 1 overflow, 2 underflow), op index, value bits, program slot (host-mapped)
    int prog_slot;
    float* poison0; long long n0;   // outputs overwritten with NaN on violation (may be null)
    float* poison1; long long n1;
    float* poison2; long long n2;
};
#define PF_RANGE_TAIL_WORDS 4

// One wave per op (round 4; one 256-thread workgroup used to walk all ~75 ops of the Student one after the other: 41-47 us at the end
// of every forward, twice per pipeline call).  Each wave reduces and clears its op's words and, on a violation, lowers the
// verdict key -- overflow (or NaN) before underflow, then program order, exactly the old kernel's choice -- with one 64-bit
// atomicMin; the last wave to take a ticket reads the key, reports and poisons.
__global__ __launch_bounds__(64) void range_verdict_kernel(RangeVerdictArgs a) {
    static_assert(PF_RANGE_SUBSLOTS == 256, "four words per lane");
    const int i = blockIdx.x, t = threadIdx.x;
    unsigned* s = a.slots + (size_t)i * PF_RANGE_OP_WORDS;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned v = s[(t + 64 * k) * PF_RANGE_STRIDE];
        s[(t + 64 * k) * PF_RANGE_STRIDE] = 0;                       // cleared for the next forward
        m = v > m ? v : m;
    }
    for (int mask = 1; mask < 64; mask <<= 1) {
        const unsigned o = (unsigned)pf_shfl_xor_i32((int)m, mask);
        m = o > m ? o : m;
    }
    unsigned long long* key = reinterpret_cast<unsigned long long*>(a.slots + (size_t)a.n_ops * PF_RANGE_OP_WORDS);
    unsigned* ticket = reinterpret_cast<unsigned*>(key + 1);
    int last = 0;
    if (t == 0) {
        if (m != 0) {                                                // 0: op not measured / all-zero tensor
            const float v = __uint_as_float(m);
            const int code = !(v <= a.hi) ? 1 : (v < a.lo ? 2 : 0);  // NaN bits land in the first case
            if (code) atomicMin(key, ((unsigned long long)(code == 1 ? 0 : 1) << 63) | ((unsigned long long)i << 32) | m);
        }
        __threadfence();
        last = atomicAdd(ticket, 1u) == (unsigned)(a.n_ops - 1) ? 1 : 0;
    }
    last = pf_shfl_i32(last, 0);
    if (!last) return;
    __threadfence();
    unsigned long long k = 0;
    if (t == 0) {
        k = atomicMin(key, ~0ull);                                   // reads the key (device scope) without changing it
        if (k != ~0ull) atomicMax(key, ~0ull);                       // ... and re-arms it for the next forward
        atomicExch(ticket, 0u);
    }
    const unsigned klo = (unsigned)pf_shfl_i32((int)(unsigned)k, 0), khi = (unsigned)pf_shfl_i32((int)(unsigned)(k >> 32), 0);
    if (klo == 0xffffffffu && khi == 0xffffffffu) return;
    if (t == 0 && a.status[0] == 0) {
        a.status[1] = (int)(khi & 0x7fffffffu); a.status[2] = (int)klo; a.status[3] = a.prog_slot;
        a.status[0] = (khi >> 31) ? 2 : 1;
    }
    const float nan = __builtin_nanf("");
    for (long long j = t; j < a.n0; j += 64) a.poison0[j] = nan;
    for (long long j = t; j < a.n1; j += 64) a.poison1[j] = nan;
    for (long long j = t; j < a.n2; j += 64) a.poison2[j] = nan;
}
