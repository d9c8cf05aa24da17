// Chains of HRNet BasicBlocks with one face's whole feature map resident in LDS (f32s mode).
//
// A BasicBlock is relu(bn2(conv2(relu(bn1(conv1(x))))) + x) with two 3x3 / stride 1 / pad 1 convs of equal width (timm
// hrnet.py BasicBlock; the Teacher's encoder, TRAIN/face_landmark/lib/core/base_trainer/model.py:306-311), and every
// HighResolutionModule runs FOUR of them back to back on each branch.  On the two low-resolution branches (72 channels
// at 16 x 16, 144 channels at 8 x 8 for a 256 x 256 crop) a conv launch is a few tens of microseconds of mostly latency:
// the map is tiny, the weights (187 / 746 KB per conv, split) are the bigger operand.  Here ONE workgroup owns ONE face
// for the whole chain (2 x n_blocks convs):
//   * the map sits in LDS as the split-precision pixel operand -- per 32-channel chunk a hi and a lo plane of
//     (H + 2) x (W + 2) halo pixels x 64 bytes, zero ring included, in the chunk-rotated layout of
//     conv3x3_halo_split_kernel, so the nine taps are shifted reads of the same planes;
//   * each wave keeps the f32 values of ITS output tile in registers for the whole chain: the accumulators of the conv in
//     flight plus the block input it will need for the residual add.  Between two convs the wave splits its tile to hi / lo
//     and overwrites its part of the planes (after a barrier: every wave has finished reading the previous tensor);
//   * weights stream per (tap, 32-channel chunk) by LDS-DMA into a ring of NSTG stages, one barrier per stage, running ahead
//     across the conv boundaries.  The requests are issued from inline asm (pf_glds16_raw) and retired with partial vmcnt
//     waits: with compiler-visible LDS-DMA every fragment read waits for ALL pending requests and the ring degenerates to
//     one round trip per stage (0.88 us per stage measured, with or without the MFMAs);
//   * 16 (72 channels) / 12 (144 channels) waves per workgroup: three to four waves per SIMD overlap one wave's fragment reads
//     with another's MFMAs between the per-stage barriers (profiles/r02_run14_teacher_chain_variants.md).
// HBM traffic per chain: the map in, the map out, the weights once per workgroup (L2 resident).  The arithmetic is the one
// conv_gemm_split_kernel does on the same f32 tensors: activations split at the same point (hi = f16(v), lo = f16(v - hi)),
// three v_mfma_f32_16x16x32_f16 per product, f32 accumulate, f32 bias / residual / relu.
// Host guarantees: N == Cin == C, maps of HW x HW pixels, weights packed by ir.pack_conv_weight(force_split=True).
#pragma once
#include "pf_common.h"
#include "k_conv_gemm.h"
#include <type_traits>

#define PF_CHAIN_MAX_CONVS 8

struct ChainArgs {
    const float* in;      // [B][HW][HW][inLd]
    float* out;           // [B][HW][HW][outLd]
    int B, inLd, outLd;
    int n_convs;          // 2 x blocks
    const void* wt[PF_CHAIN_MAX_CONVS];     // [NTILES*16][9][CBLK][hi 32 x f16 | lo 32 x f16]
    const float* bias[PF_CHAIN_MAX_CONVS];  // [NTILES*16]
    float acc_scale[PF_CHAIN_MAX_CONVS];
    unsigned* range_slot; // range guard: max |v| over every tensor this launch splits (float bits, atomicMax), or nullptr
    int dbg;
};

template <int C, int HW, int WARPS_M, int WARPS_N, int NT, int NSTG>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64, WARPS_M * WARPS_N / 4) void basic_chain_kernel(ChainArgs a) {
    constexpr int NTHR = WARPS_M * WARPS_N * 64;
    constexpr int CBLK = (C + 31) / 32;
    constexpr int NTILES = (C + 15) / 16;
    constexpr int BN = NTILES * 16;
    constexpr int HW2 = HW + 2;
    constexpr int HP = HW2 * HW2;
    constexpr int M = HW * HW;
    constexpr int WM = M / WARPS_M;
    constexpr int MT = WM / 16;
    constexpr int PLANE = HP * 64;                       // one (chunk, hi | lo) plane
    constexpr int ACT_BYTES = CBLK * 2 * PLANE;
    constexpr int WSLOTS = BN * 8;                       // 16-byte slots of one weight stage (hi plane, then lo plane)
    constexpr int WCHUNKS = (WSLOTS + NTHR - 1) / NTHR;      // LDS-DMA requests of a full wave per stage ...
    constexpr int WFULL = (WSLOTS % NTHR) / 64;              // ... the last round is issued by the first WFULL waves only
    constexpr int W_BYTES = WSLOTS * 16;
    constexpr int AHEAD = NSTG - 1;                          // stages in flight
    static_assert(NTHR <= 1024 && (WARPS_M * WARPS_N) % 4 == 0 && WM % 16 == 0 && WARPS_N * NT >= NTILES && C % 4 == 0, "tile shape");
    static_assert(WSLOTS % 64 == 0 && WFULL > 0 && NSTG >= 2 && (AHEAD - 1) * WCHUNKS < 64, "weight ring");
    constexpr int BIAS_BYTES = PF_CHAIN_MAX_CONVS * BN * 4;  // every conv's bias: no VMEM besides the weight ring inside the chain
    static_assert(ACT_BYTES + NSTG * W_BYTES + BIAS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ACT_BYTES + NSTG * W_BYTES + BIAS_BYTES];
    unsigned char* wbase = smem + ACT_BYTES;
    float* sbias = reinterpret_cast<float*>(smem + ACT_BYTES + NSTG * W_BYTES);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    const int face = blockIdx.x;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int crow = fchunk * 4;
    const float* __restrict__ in = a.in + (size_t)face * M * a.inLd;
    float* __restrict__ out = a.out + (size_t)face * M * a.outLd;

    // ---- zero the planes (halo ring and padding channels stay zero for the whole chain) ------------------------------
    for (int o = t * 16; o < ACT_BYTES; o += NTHR * 16) *reinterpret_cast<pf_f32x4*>(smem + o) = pf_f32x4{0.f, 0.f, 0.f, 0.f};

    // weight stage `s` of the chain (s = conv * NK + tap * CBLK + cb) -> ring slot s % NSTG
    constexpr int NK = 9 * CBLK;
    const int n_stages = a.n_convs * NK;
    auto load_w = [&](int s) {
        if (s >= n_stages) return;
        const int conv = s / NK, kt = s - conv * NK;
        const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt[conv]);
        unsigned char* wdst = wbase + (s % NSTG) * W_BYTES;
#pragma unroll
        for (int c = 0; c < WCHUNKS; ++c) {
            const int sl = t + NTHR * c;
            if (sl < WSLOTS) {
                const int plane = sl >= BN * 4 ? 1 : 0;
                const int row = (sl - plane * BN * 4) >> 2;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                pf_glds16_raw(wt + ((size_t)row * NK + kt) * 128 + plane * 64 + chunk * 16, wdst + sl * 16);
            }
        }
    };
    // "stage s has landed": everything this wave issued except the AHEAD - 1 younger stages is complete
    auto wait_stage = [&](int s) {
        if (s + AHEAD - 1 < n_stages) {
            if (wave < WFULL) pf_wait_vm_barrier<(AHEAD - 1) * WCHUNKS>();
            else pf_wait_vm_barrier<(AHEAD - 1) * (WCHUNKS - 1)>();
        } else {
            pf_wait_vm_barrier<0>();                       // the chain's last stages: drain
        }
    };

    // this lane's pixels: halo-plane row at tap (0, 0) and map index
    int hp0[MT], pix[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int p = wm * WM + i * 16 + frow;
        const int py = p / HW, px = p - py * HW;
        pix[i] = p;
        hp0[i] = py * HW2 + px;
    }
    // this wave's channel tiles (wave-uniform count)
    const int jt0 = wn * NT;
    const int njt = min(NT, NTILES - jt0);

    float vmax = 0.f;
    bool vbad = false;
    // split this lane's 4 channels x MT pixels of tile j and park them in the planes
    auto park = [&](const pf_f32x4 (&v)[NT][MT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j >= njt) break;
            const int jt = jt0 + j;
            const int cb = jt >> 1;
            const int unit = ((jt & 1) << 1) + (fchunk >> 1);
            unsigned char* ph = smem + (size_t)(cb * 2) * PLANE;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                pf_half4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = v[j][i][r];
                    const pf_half hv = (pf_half)x;
                    hi[r] = hv;
                    lo[r] = pf_split_lo(x, hv);
                    const float ax = fabsf(x);
                    vbad |= !(ax == ax);
                    vmax = fmaxf(vmax, ax);
                }
                const int hp = hp0[i] + HW2 + 1;
                const int off = pf_lds_chunk_off(hp, unit) + (fchunk & 1) * 8;
                *reinterpret_cast<pf_half4*>(ph + off) = hi;
                *reinterpret_cast<pf_half4*>(ph + PLANE + off) = lo;
            }
        }
    };

    // ---- the block input: global -> registers (accumulator layout) -> planes --------------------------------------------
    pf_f32x4 resid[NT][MT], acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int n = (jt0 + j) * 16 + crow;
            pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < njt && n < C) v = *reinterpret_cast<const pf_f32x4*>(in + (size_t)pix[i] * a.inLd + n);
            resid[j][i] = v;
        }
    for (int i = t; i < a.n_convs * BN; i += NTHR) sbias[i] = a.bias[i / BN][i % BN];
    __syncthreads();                                     // zero fill complete before anybody parks values (drains the x loads too)
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) load_w(s);
    park(resid);

    int stage = 0;
    for (int conv = 0; conv < a.n_convs; ++conv) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        int tap = 0, cb = 0;
        for (int kt = 0; kt < NK; ++kt, ++stage) {
            const bool last_cb = cb == CBLK - 1;
            wait_stage(stage);                           // this stage's weights landed; everybody left the previous stage (and parked)
            load_w(stage + AHEAD);                       // into the slot the previous stage just released
            const unsigned char* wh = wbase + (stage % NSTG) * W_BYTES;
            const unsigned char* wl = wh + BN * 64;
            const unsigned char* xh = smem + (size_t)(cb * 2) * PLANE;
            const unsigned char* xl = xh + PLANE;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int shift = ky * HW2 + kx;
            if (!(pf_dbg(a) & 16)) {
                pf_half8 xhf[MT], xlf[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int off = pf_lds_chunk_off(hp0[i] + shift, fchunk);
                    xhf[i] = *reinterpret_cast<const pf_half8*>(xh + off);
                    xlf[i] = *reinterpret_cast<const pf_half8*>(xl + off);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (j >= njt) break;
                    const int off = pf_lds_chunk_off((jt0 + j) * 16 + frow, fchunk);
                    const pf_half8 whf = *reinterpret_cast<const pf_half8*>(wh + off);
                    const pf_half8 wlf = *reinterpret_cast<const pf_half8*>(wl + off);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(wlf, xhf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xlf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xhf[i], acc[j][i]);
                }
            }
            if (last_cb) { cb = 0; ++tap; } else ++cb;
        }
        // ---- epilogue in registers: bias, (residual), relu; the result is the next conv's operand ---------------------------
        const float sc = a.acc_scale[conv];
        const float* bias = sbias + conv * BN;
        const bool second = (conv & 1) != 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j >= njt) break;
            const int n = (jt0 + j) * 16 + crow;
            const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[j][i][r] * sc + bv[r];
                    if (second) v += resid[j][i][r];
                    acc[j][i][r] = pf_act_c<PF_ACT_RELU>(v);
                }
                if (second) resid[j][i] = acc[j][i];
            }
        }
        if (conv + 1 < a.n_convs) {
            pf_wait_vm_barrier<63>();                    // everybody is done reading the planes (weights stay in flight)
            park(acc);                                   // made visible by the next stage's barrier
        }
    }

    // ---- the chain's output ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j >= njt) break;
        const int n = (jt0 + j) * 16 + crow;
        if (n < C && !(pf_dbg(a) & 32)) {
#pragma unroll
            for (int i = 0; i < MT; ++i) *reinterpret_cast<pf_f32x4*>(out + (size_t)pix[i] * a.outLd + n) = resid[j][i];
        }
    }
    if (a.range_slot) {
        if (vbad) vmax = __builtin_inff();
        for (int mask = 1; mask < 64; mask <<= 1) vmax = fmaxf(vmax, pf_shfl_xor_f32(vmax, mask));
        if (lane == 0 && vmax != 0.f) { unsigned* w = pf_amax_word(a.range_slot); if (__float_as_uint(vmax) > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, __float_as_uint(vmax)); }
    }
}

// ---- one BasicBlock per launch on the two high-resolution HRNet branches (18 channels @ 64 x 64, 36 @ 32 x 32) ---------------
// Those maps do not fit in LDS, and as single convs they are neither HBM- nor matrix-core bound: a workgroup loads its patch,
// waits, runs nine short tap steps with a barrier each, stores, and the 18 (36) channels are padded to a 32 (64) deep K step
// and a 32 (48) wide N tile.  Here a workgroup computes TR output rows of one face through BOTH convs:
//   * x patch (TR + 4 rows, zero ring) -> split hi / lo planes in LDS, ONE load of the tensor per block instead of two plus
//     the intermediate's round trip;
//   * K is the flattened (tap, 8-channel group) axis: pixel rows in the planes are CG x 16 bytes (CG = ceil(C / 8): 48 / 80
//     bytes), a 32-deep MFMA step takes four consecutive (tap, group) pairs, each
//     lane reading ITS pair's shifted pixel -- 7 steps instead of 9 at 18 channels, 12 instead of 18 at 36;
//   * conv1 runs on TR + 2 rows (one-row halo recompute), its relu output is split in registers and parked over the x planes
//     (rows outside the image as zeros: they are conv2's padding), conv2 runs on the TR rows, and the f32 residual -- read
//     from global at the start -- is added in the epilogue;
//   * weights: [Npad][NCH][hi 32 | lo 32] with k = tap * 8 CG + c (ir.pack_flatk_weight); G chunks per LDS stage, either the
//     whole conv at once (NBUF = 1: no barrier inside a conv) or double-buffered per chunk (NBUF = 2).
template <int C, int W, int TR, int G, int NBUF>
struct BlockCfg {
    static constexpr int CG = (C + 7) / 8;               // 8-channel groups per pixel
    static constexpr int ROWB = CG * 16;                 // bytes per pixel in one plane
    static constexpr int KG = 9 * CG;                    // (tap, group) pairs
    static constexpr int NCH = (KG + 3) / 4;             // 32-deep K steps
    static constexpr int NTILES = (C + 15) / 16;
    static constexpr int BN = NTILES * 16;
    static constexpr int W2 = W + 2;
    static constexpr int XROWS = TR + 4;
    static constexpr int PLANE = XROWS * W2 * ROWB;      // hi (or lo) plane of the x patch; y1 reuses the first TR + 2 rows
    static constexpr int M1T = (TR + 2) * W / 16, M2T = TR * W / 16;
    static constexpr int MT1 = (M1T + 7) / 8, MT2 = M2T / 8;
    static constexpr int NG = (NCH + G - 1) / G;         // weight stages per conv
    static constexpr int WSLOTS = G * BN * 8;            // 16-byte slots of one weight stage
    static constexpr int WROUNDS = (WSLOTS + 511) / 512;
    static constexpr int W_BYTES = WSLOTS * 16;
    static constexpr int LDS = 2 * PLANE + NBUF * W_BYTES;
};

struct BlockArgs {
    const float* in;      // [B][H][W][inLd], channels [C, Cs) zero
    float* out;           // [B][H][W][outLd]
    int B, H, inLd, outLd, Cs;
    const void* wt[2];    // flat-K split weights of conv1 / conv2
    const float* bias[2]; // [BN]
    float acc_scale[2];
    unsigned* range_slot;
    int dbg;
};

template <int C, int W, int TR, int G, int NBUF>
__global__ __launch_bounds__(512, 4) void basic_block_kernel(BlockArgs a) {
    using K = BlockCfg<C, W, TR, G, NBUF>;
    constexpr int CG = K::CG, ROWB = K::ROWB, NCH = K::NCH, NTILES = K::NTILES, BN = K::BN, W2 = K::W2, PLANE = K::PLANE;
    constexpr int MT1 = K::MT1, MT2 = K::MT2, NT = NTILES;
    static_assert(W % 16 == 0 && K::M2T % 8 == 0 && (NBUF == 1 ? K::NG == 1 : G == 1), "tile shape");
    static_assert(K::LDS <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[K::LDS];
    unsigned char* xh = smem;
    unsigned char* xl = smem + PLANE;
    unsigned char* wbase = smem + 2 * PLANE;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int crow = fchunk * 4;
    const int tiles_per_face = a.H / TR;
    const int face = blockIdx.x / tiles_per_face;
    const int y0 = (blockIdx.x - face * tiles_per_face) * TR;
    const float* __restrict__ in = a.in + (size_t)face * a.H * W * a.inLd;
    float* __restrict__ out = a.out + (size_t)face * a.H * W * a.outLd;

    auto load_w = [&](int conv, int g, int buf) {
        const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt[conv]);
        unsigned char* wdst = wbase + buf * K::W_BYTES;
#pragma unroll
        for (int c = 0; c < K::WROUNDS; ++c) {
            const int sl = t + 512 * c;
            const int q = sl / (BN * 8);                 // chunk within the stage
            const int s = sl - q * (BN * 8);
            if (sl < K::WSLOTS && g * G + q < NCH) {
                const int plane = s >= BN * 4 ? 1 : 0;
                const int row = (s - plane * BN * 4) >> 2;
                const int chunk = ((s & 3) - 2 * (row >> 2)) & 3;
                pf_glds16(wt + ((size_t)row * NCH + g * G + q) * 128 + plane * 64 + chunk * 16, wdst + sl * 16);
            }
        }
    };

    // ---- x patch: zero fill, then the rows that exist, split to hi / lo --------------------------------------------------------
    load_w(0, 0, 0);
    for (int o = t * 16; o < 2 * PLANE; o += 512 * 16) *reinterpret_cast<pf_f32x4*>(smem + o) = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int XUNITS = K::XROWS * W * CG;
    constexpr int XU = (XUNITS + 511) / 512;
    pf_f32x4 xreg[XU][2];
    int xdst[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int id = t + 512 * u;
        const int px = id / CG, cg = id - px * CG;
        const int xr = px / W, xc = px - xr * W;
        const int iy = y0 - 2 + xr;
        const bool ok = id < XUNITS && (unsigned)iy < (unsigned)a.H;
        xdst[u] = ok ? (xr * W2 + xc + 1) * ROWB + cg * 16 : -1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok && cg * 8 + 4 * h < a.Cs) v = *reinterpret_cast<const pf_f32x4*>(in + (size_t)(iy * W + xc) * a.inLd + cg * 8 + 4 * h);
            xreg[u][h] = v;
        }
    }
    // this wave's output tiles: conv1 tile i covers pixels (tile * 16 ..) of the (TR + 2) x W region, conv2 of the TR x W region
    int hp1[MT1], hp2[MT2], opix[MT2];
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        const int p = (i * 8 + wave) * 16 + frow;
        const int r = p / W, c = p - r * W;
        hp1[i] = (r * W2 + c) * ROWB;
    }
#pragma unroll
    for (int i = 0; i < MT2; ++i) {
        const int p = (i * 8 + wave) * 16 + frow;
        const int r = p / W, c = p - r * W;
        hp2[i] = (r * W2 + c) * ROWB;
        opix[i] = (y0 + r) * W + c;
    }
    // residual: x at this lane's output pixels / channels (accumulator layout), in flight while conv1 runs
    pf_f32x4 resid[NT][MT2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT2; ++i) {
            const int n = j * 16 + crow;
            pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < a.Cs) v = *reinterpret_cast<const pf_f32x4*>(in + (size_t)opix[i] * a.inLd + n);
            resid[j][i] = v;
        }
    // per K step q, this lane's (tap, channel group) pair kg = 4 q + fchunk -> byte offset inside a plane
    auto koff = [&](int q) {
        int kg = q * 4 + fchunk;
        if (kg >= K::KG) kg = 0;                          // zero weights there; any finite operand will do
        const int tap = kg / CG, cg = kg - tap * CG;
        const int ky = tap / 3, kx = tap - ky * 3;
        return (ky * W2 + kx) * ROWB + cg * 16;
    };
    float vmax = 0.f;
    bool vbad = false;
    __syncthreads();                                      // zero fill done
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        if (xdst[u] < 0) continue;
        pf_half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = xreg[u][e >> 2][e & 3];
            const pf_half hv = (pf_half)v;
            hi[e] = hv;
            lo[e] = pf_split_lo(v, hv);
            const float av = fabsf(v);
            vbad |= !(av == av);
            vmax = fmaxf(vmax, av);
        }
        *reinterpret_cast<pf_half8*>(xh + xdst[u]) = hi;
        *reinterpret_cast<pf_half8*>(xl + xdst[u]) = lo;
    }
    __syncthreads();                                      // planes and the first weight stage are in place

    // one conv over this wave's MT tiles; weights stage `buf` holds chunk group g
    auto mma_group = [&](auto& acc, const int* hp, auto mt_tag, int ntl, int g, int buf) {
        constexpr int MT = decltype(mt_tag)::value;
#pragma unroll
        for (int qq = 0; qq < G; ++qq) {
            const int q = g * G + qq;
            if (q >= NCH) break;
            const unsigned char* wh = wbase + buf * K::W_BYTES + qq * (BN * 128);
            const unsigned char* wl = wh + BN * 64;
            pf_half8 xhf[MT], xlf[MT];
            const int ko = koff(q);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i >= ntl) break;
                xhf[i] = *reinterpret_cast<const pf_half8*>(xh + hp[i] + ko);
                xlf[i] = *reinterpret_cast<const pf_half8*>(xl + hp[i] + ko);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int off = pf_lds_chunk_off(j * 16 + frow, fchunk);
                const pf_half8 whf = *reinterpret_cast<const pf_half8*>(wh + off);
                const pf_half8 wlf = *reinterpret_cast<const pf_half8*>(wl + off);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (i >= ntl) break;
                    acc[j][i] = pf_mfma_16x16x32_f16(wlf, xhf[i], acc[j][i]);
                    acc[j][i] = pf_mfma_16x16x32_f16(whf, xlf[i], acc[j][i]);
                    acc[j][i] = pf_mfma_16x16x32_f16(whf, xhf[i], acc[j][i]);
                }
            }
        }
    };
    using Tag1 = std::integral_constant<int, MT1>;
    using Tag2 = std::integral_constant<int, MT2>;
    // tiles of conv1 this wave owns (the last round may be ragged)
    const int nt1 = (K::M1T - wave + 7) / 8;

    // ---- conv1 -----------------------------------------------------------------------------------------------------------------------
    pf_f32x4 acc1[NT][MT1];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT1; ++i) acc1[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    if constexpr (NBUF == 1) {
        if (!(pf_dbg(a) & 16)) mma_group(acc1, hp1, Tag1{}, nt1, 0, 0);
        __syncthreads();                                  // everybody is done with x and with conv1's weights
        load_w(1, 0, 0);
    } else {
        for (int g = 0; g < K::NG; ++g) {
            if (g + 1 < K::NG) load_w(0, g + 1, buf ^ 1); else load_w(1, 0, buf ^ 1);
            if (!(pf_dbg(a) & 16)) mma_group(acc1, hp1, Tag1{}, nt1, g, buf);
            __syncthreads();
            buf ^= 1;
        }
    }
    // ---- relu(conv1 + b1) -> split -> over the x planes (pixel (r, c) of the y1 region at plane pixel (r, c + 1)) --------------------
    {
        const float sc = a.acc_scale[0];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 16 + crow;
            if (n >= CG * 8) continue;                    // beyond the channel groups a plane row holds
            const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(a.bias[0] + n);
#pragma unroll
            for (int i = 0; i < MT1; ++i) {
                if (i >= nt1) break;
                const int p = (i * 8 + wave) * 16 + frow;
                const int r = p / W;
                const bool inside = (unsigned)(y0 - 1 + r) < (unsigned)a.H;
                pf_half4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = pf_act_c<PF_ACT_RELU>(acc1[j][i][e] * sc + bv[e]);
                    if (!inside) v = 0.f;
                    const pf_half hv = (pf_half)v;
                    hi[e] = hv;
                    lo[e] = pf_split_lo(v, hv);
                    const float av = fabsf(v);
                    vbad |= !(av == av);
                    vmax = fmaxf(vmax, av);
                }
                const int off = hp1[i] + ROWB + (n >> 3) * 16 + (n & 4) * 2;
                *reinterpret_cast<pf_half4*>(xh + off) = hi;
                *reinterpret_cast<pf_half4*>(xl + off) = lo;
            }
        }
    }
    __syncthreads();                                      // y1 parked (and, NBUF == 1, conv2's weights landed)

    // ---- conv2 -----------------------------------------------------------------------------------------------------------------------
    pf_f32x4 acc2[NT][MT2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc2[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NBUF == 1) {
        if (!(pf_dbg(a) & 16)) mma_group(acc2, hp2, Tag2{}, MT2, 0, 0);
    } else {
        for (int g = 0; g < K::NG; ++g) {
            if (g + 1 < K::NG) load_w(1, g + 1, buf ^ 1);
            if (!(pf_dbg(a) & 16)) mma_group(acc2, hp2, Tag2{}, MT2, g, buf);
            if (g + 1 < K::NG) __syncthreads();
            buf ^= 1;
        }
    }
    {
        const float sc = a.acc_scale[1];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 16 + crow;
            if (n >= a.Cs || (pf_dbg(a) & 32)) continue;
            const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(a.bias[1] + n);
#pragma unroll
            for (int i = 0; i < MT2; ++i) {
                pf_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pf_act_c<PF_ACT_RELU>(acc2[j][i][e] * sc + bv[e] + resid[j][i][e]);
                *reinterpret_cast<pf_f32x4*>(out + (size_t)opix[i] * a.outLd + n) = v;
            }
        }
    }
    if (a.range_slot) {
        if (vbad) vmax = __builtin_inff();
        for (int mask = 1; mask < 64; mask <<= 1) vmax = fmaxf(vmax, pf_shfl_xor_f32(vmax, mask));
        if (lane == 0 && vmax != 0.f) { unsigned* w = pf_amax_word(a.range_slot); if (__float_as_uint(vmax) > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, __float_as_uint(vmax)); }
    }
}
