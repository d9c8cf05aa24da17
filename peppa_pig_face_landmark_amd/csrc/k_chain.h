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
//   * weights stream per (tap, 32-channel chunk) by LDS-DMA into two stages, one barrier per stage, prefetched across the
//     conv boundaries.
// HBM traffic per chain: the map in, the map out, the weights once per workgroup (L2 resident).  The arithmetic is the one
// conv_gemm_split_kernel does on the same f32 tensors: activations split at the same point (hi = f16(v), lo = f16(v - hi)),
// three v_mfma_f32_16x16x32_f16 per product, f32 accumulate, f32 bias / residual / relu.
// Host guarantees: N == Cin == C, maps of HW x HW pixels, weights packed by ir.pack_conv_weight(force_split=True).
#pragma once
#include "pf_common.h"
#include "k_conv_gemm.h"

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

template <int C, int HW, int WARPS_M, int WARPS_N, int NT>
__global__ __launch_bounds__(512, 2) void basic_chain_kernel(ChainArgs a) {
    constexpr int NTHR = 512;
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
    constexpr int WCHUNKS = (WSLOTS + NTHR - 1) / NTHR;
    constexpr int W_BYTES = WCHUNKS * NTHR * 16;
    static_assert(WARPS_M * WARPS_N == 8 && WM % 16 == 0 && WARPS_N * NT >= NTILES && C % 4 == 0, "tile shape");
    static_assert(ACT_BYTES + 2 * W_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ACT_BYTES + 2 * W_BYTES];
    unsigned char* wbase = smem + ACT_BYTES;

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

    auto load_w = [&](int conv, int tap, int cb, int stage) {
        const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt[conv]);
        unsigned char* wdst = wbase + stage * W_BYTES;
#pragma unroll
        for (int c = 0; c < WCHUNKS; ++c) {
            const int sl = t + NTHR * c;
            if (sl < WSLOTS) {
                const int plane = sl >= BN * 4 ? 1 : 0;
                const int row = (sl - plane * BN * 4) >> 2;
                const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
                pf_glds16(wt + ((size_t)(row * 9 + tap) * CBLK + cb) * 128 + plane * 64 + chunk * 16, wdst + sl * 16);
            }
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
                    lo[r] = (pf_half)(x - (float)hv);
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
    load_w(0, 0, 0, 0);
    __syncthreads();                                     // zero fill complete before anybody parks values
    park(resid);
    __syncthreads();

    constexpr int NK = 9 * CBLK;
    int stage = 0;
    for (int conv = 0; conv < a.n_convs; ++conv) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        int tap = 0, cb = 0;
        for (int kt = 0; kt < NK; ++kt) {
            const bool last_cb = cb == CBLK - 1;
            if (kt + 1 < NK) load_w(conv, last_cb ? tap + 1 : tap, last_cb ? 0 : cb + 1, stage ^ 1);
            else if (conv + 1 < a.n_convs) load_w(conv + 1, 0, 0, stage ^ 1);
            const unsigned char* wh = wbase + stage * W_BYTES;
            const unsigned char* wl = wh + BN * 64;
            const unsigned char* xh = smem + (size_t)(cb * 2) * PLANE;
            const unsigned char* xl = xh + PLANE;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int shift = ky * HW2 + kx;
            if (!(a.dbg & 16)) {
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
            __syncthreads();
            stage ^= 1;
            if (last_cb) { cb = 0; ++tap; } else ++cb;
        }
        // ---- epilogue in registers: bias, (residual), relu; the result is the next conv's operand ---------------------------
        const float sc = a.acc_scale[conv];
        const float* __restrict__ bias = a.bias[conv];
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
            park(acc);                                   // every wave left the K loop through a barrier: the planes are free
            __syncthreads();
        }
    }

    // ---- the chain's output ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j >= njt) break;
        const int n = (jt0 + j) * 16 + crow;
        if (n < C && !(a.dbg & 32)) {
#pragma unroll
            for (int i = 0; i < MT; ++i) *reinterpret_cast<pf_f32x4*>(out + (size_t)pix[i] * a.outLd + n) = resid[j][i];
        }
    }
    if (a.range_slot) {
        if (vbad) vmax = __builtin_inff();
        for (int mask = 1; mask < 64; mask <<= 1) vmax = fmaxf(vmax, pf_shfl_xor_f32(vmax, mask));
        if (lane == 0) atomicMax(a.range_slot, __float_as_uint(vmax));
    }
}
