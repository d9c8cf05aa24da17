// 3x3 / stride 1 / pad 1 conv, 128 -> 128 channels, on 16-, 32- or 64-pixel-wide maps in f32s mode -- the Student's hero
// layer up2.conv2 (model.py:165-172, 40.7 % of all MACs) -- as ONE 16-wave workgroup per CU on a 256-pixel tile (round 3).
//
// conv3x3_halo_split_kernel (k_conv_gemm.h) keeps the input patch of a 128-pixel tile resident in LDS and streams only the
// weights, one 16 KB (tap, 32-channel chunk) step per barrier, two 8-wave workgroups per CU.  Its timing ablations add up
// almost linearly (profiles/r02_run2_hero_timing_ablations.md: 0.18 skeleton + 0.31 MFMA + 0.11 weight DMA + 0.09 stores +
// 0.06 patch staging + 0.05 barriers = 0.78 ms): a step carries only 24 MFMAs per wave (384 cycles of its SIMD's matrix
// pipe), and every step pays run-time tap / chunk bookkeeping, a barrier turn-around and the LDS latency of its first
// fragment reads.  Three round-3 cuts of a producer / consumer pipelined persistent kernel (the structure of k_sepup.h;
// profiles/r03_hero_pipe_experiments.md) ended at 0.79 ms: with all 16 waves of a CU in one workgroup nothing runs during a
// barrier's turn-around (~400 cycles), only half the waves feed the matrix pipe, and the weight requests of a 3-step
// interval (~2000 cycles from request to landing) are late for an interval that short.  What those cuts taught is applied here:
//   * a 256-pixel tile (4 image rows at W = 64): 16 waves x (32 pixels x 64 channels), four waves per SIMD all feeding the
//     matrix pipe; a step's 16 KB of weights now serves 48 MFMAs per SIMD-wave pair ... i.e. twice the matrix work per byte;
//   * ONE barrier per kernel row (three steps, 72 MFMAs per wave, ~4600 cycles of each SIMD's matrix pipe): the weight ring
//     is two 48 KB halves, the requests of a whole row are issued at the start of the interval before (one asm-issued LDS-DMA
//     per thread and step) and have that interval to land;
//   * the tap loop is unrolled: tap, ring half and fragment shifts are compile-time (the pipelined cut that derived them from
//     a running step counter spent ~900 cycles per step on integer divisions and select chains);
//   * the next chunk's patch is loaded (global loads) during the chunk's first row, split to f16 hi / lo in registers during
//     the second, and written over the single patch buffer in a short interval of its own after the third.
// Arithmetic, operand layouts and the weight format are those of conv3x3_halo_split_kernel (same packed weights, same
// epilogue: acc * acc_scale + bias -> activation).  Host guarantees: KH = KW = 3, stride = pad = dil = 1, Cpad % 32 == 0,
// N == Npad == 128, W in {16, 32, 64}, (H * W) % 256 == 0, no residual / gate / per-face bias / arg-max.
#pragma once
#include "pf_common.h"
#include "k_conv_gemm.h"

template <int W>
__global__ __launch_bounds__(1024, 4) void hero_wide_kernel(ConvGemmArgs a) {
    constexpr int BM = 256, BN = 128;
    constexpr int TR = BM / W;                           // image rows per tile
    constexpr int HW2 = W + 2;
    constexpr int HP = (TR + 2) * HW2;                   // halo pixels of a tile: 396 / 340 / 324
    constexpr int PLANE = ((HP * 64 + 1023) / 1024) * 1024;   // one (hi | lo) plane of a 32-channel chunk
    constexpr int XU = (HP * 4 + 1023) / 1024;           // (halo pixel, 8-float unit) pairs per thread
    constexpr int W_BYTES = BN * 128;                    // weights of one step: hi rows then lo rows = 1024 16-byte slots
    constexpr int H_BYTES = 3 * W_BYTES;                 // one half of the weight ring: a kernel row
    constexpr int WN = BN / 2, NT = WN / 16;
    static_assert(BN * 8 == 1024, "one weight request per thread and step");
    static_assert(2 * PLANE + 2 * H_BYTES + BN * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PLANE + 2 * H_BYTES + BN * 4];
    unsigned char* const xh = smem;
    unsigned char* const xl = smem + PLANE;
    unsigned char* const wbase = smem + 2 * PLANE;
    float* const sbias = reinterpret_cast<float*>(wbase + 2 * H_BYTES);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave & 7, wn = wave >> 3;             // 8 x 2 waves of 32 pixels x 64 channels
    int mtile = blockIdx.x;
    if ((gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware tile order
    const int H = a.outH, OHW = H * W;
    const int m0 = mtile * BM;
    const int face = m0 / OHW;
    const int y0 = (m0 - face * OHW) / W;
    const int cblocks = a.Cpad >> 5;
    const size_t wrow_bytes = (size_t)9 * cblocks * 128;
    const float* __restrict__ in = static_cast<const float*>(a.in) + (size_t)face * OHW * a.inLd;

    if (t < BN) sbias[t] = a.bias[t];

    // weight rows are [n][tap][chunk][128 B]: this thread's request slot (source row / 16-byte piece, LDS slot)
    const unsigned char* wsrc;
    {
        const int plane = t >= BN * 4 ? 1 : 0;
        const int row = (t - plane * BN * 4) >> 2;
        const int chunk = ((t & 3) - 2 * (row >> 2)) & 3;
        wsrc = static_cast<const unsigned char*>(a.wt) + (size_t)row * wrow_bytes + plane * 64 + chunk * 16;
    }
    auto w_issue = [&](int ky, int cb, int half) {       // the three taps of kernel row ky, chunk cb -> ring half
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            pf_glds16_raw(wsrc + ((ky * 3 + kx) * cblocks + cb) * 128, wbase + half * H_BYTES + kx * W_BYTES + t * 16);
    };
    // this thread's halo units: clamped source pixel, "inside the image" bit; exactly 2 * XU loads per thread and chunk
    const int xc = t & 3;
    int xhp[XU];
    const float* xsrc[XU];
    unsigned xok = 0;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int hp = (t >> 2) + 256 * u;
        xhp[u] = hp < HP ? hp : -1;
        const int hq = hp < HP ? hp : 0;
        const int hy = hq / HW2, hx = hq - hy * HW2;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) xok |= 1u << u;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        xsrc[u] = in + (size_t)(cy * W + cx) * a.inLd + xc * 8;
    }
    pf_f32x4 xreg[XU][2];
    pf_half8 xph[XU], xpl[XU];
    unsigned amax = 0;                                   // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    auto load_x = [&](int cb) {
#pragma unroll
        for (int u = 0; u < XU; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) xreg[u][h] = *reinterpret_cast<const pf_f32x4*>(xsrc[u] + cb * 32 + 4 * h);
    };
    auto split_x = [&]() {                               // f32 registers -> packed hi / lo (zero padding included)
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const bool ok = (xok >> u) & 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = ok ? xreg[u][e >> 2][e & 3] : 0.f;
                const pf_half hv = (pf_half)v;
                xph[u][e] = hv;
                xpl[u][e] = (pf_half)(v - (float)hv);
                amax = pf_amax(amax, v);
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (xhp[u] < 0) continue;
            const int off = pf_lds_chunk_off(xhp[u], xc);
            *reinterpret_cast<pf_half8*>(xh + off) = xph[u];
            *reinterpret_cast<pf_half8*>(xl + off) = xpl[u];
        }
    };

    const int frow = lane & 15, fchunk = lane >> 4;
    int hp0[2];                                          // halo row of this lane's two pixels at tap (0, 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wm * 32 + i * 16 + frow;
        const int ty = p / W, tx = p - ty * W;
        hp0[i] = ty * HW2 + tx;
    }
    pf_f32x4 acc[NT][2];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) { acc[jt][0] = pf_f32x4{0.f, 0.f, 0.f, 0.f}; acc[jt][1] = pf_f32x4{0.f, 0.f, 0.f, 0.f}; }

    load_x(0);
    if (!(pf_dbg(a) & 1)) w_issue(0, 0, 0);
    split_x();
    store_x();
    pf_wait_vm_barrier<0>();

    // Interval ii = 3 * chunk + kernel row reads ring half ii & 1 = (chunk + row) & 1; the requests of interval ii + 1 go out at
    // the start of interval ii, into the half read in interval ii - 1, and have landed at its closing barrier.  The patch
    // loads of the next chunk go out BEFORE the first row's weight requests (the compiler guards the registers it reuses with
    // a vmcnt(0) -- it sees none of the asm-issued requests); their first use is pinned right after that row's barrier, where
    // that vmcnt(0) is free.
    for (int cb = 0; cb < cblocks; ++cb) {
        const bool has_next = cb + 1 < cblocks;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const unsigned char* whalf = wbase + ((cb + ky) & 1) * H_BYTES;
            if (!(pf_dbg(a) & 16)) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (kx == 1) {                       // after the first tap: the matrix pipes start right behind the barrier
                        if (ky == 0 && has_next && !(pf_dbg(a) & 128)) load_x(cb + 1);
                        if (!(pf_dbg(a) & 1)) {
                            if (ky < 2) w_issue(ky + 1, cb, (cb + ky + 1) & 1);
                            else if (has_next) w_issue(0, cb + 1, (cb + 3) & 1);
                        }
                    }
                    if (kx == 2 && ky == 1 && has_next && !(pf_dbg(a) & 128)) split_x();
                    const int shift = ky * HW2 + kx;
                    const unsigned char* wh = whalf + kx * W_BYTES;
                    const unsigned char* wlp = wh + BN * 64;
                    pf_half8 xhf[2], xlf[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        int h0 = hp0[i];
                        asm volatile("" : "+v"(h0));     // keep the nine taps' offsets from being hoisted out of the chunk loop (18 VGPRs)
                        const int off = pf_lds_chunk_off(h0 + shift, fchunk);
                        xhf[i] = *reinterpret_cast<const pf_half8*>(xh + off);
                        xlf[i] = *reinterpret_cast<const pf_half8*>(xl + off);
                    }
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        const int off = pf_lds_chunk_off(wn * WN + jt * 16 + frow, fchunk);
                        const pf_half8 whf = *reinterpret_cast<const pf_half8*>(wh + off);
                        const pf_half8 wlf = *reinterpret_cast<const pf_half8*>(wlp + off);
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[jt][i] = pf_mfma_16x16x32_f16(wlf, xhf[i], acc[jt][i]);     // small terms first
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[jt][i] = pf_mfma_16x16x32_f16(whf, xlf[i], acc[jt][i]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[jt][i] = pf_mfma_16x16x32_f16(whf, xhf[i], acc[jt][i]);
                    }
                }
            }
            pf_wait_vm_barrier<0>();
            if (ky == 0 && has_next) {
#pragma unroll
                for (int u = 0; u < XU; ++u) asm volatile("" : "+v"(xreg[u][0]), "+v"(xreg[u][1]));
            }
        }
        if (has_next) {                                  // every wave is done with this chunk's patch
            if (!(pf_dbg(a) & 128)) store_x();
            pf_wait_vm_barrier<0>();
        }
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);

    // epilogue: bias, activation, 16-byte stores (acc[jt][i][r] = channel wn * 64 + 16 jt + 4 (lane >> 4) + r, pixel wm * 32 + 16 i + (lane & 15))
    float* __restrict__ orow = static_cast<float*>(a.out) + (size_t)m0 * a.outLd;
    const int crow = fchunk * 4;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int n = wn * WN + jt * 16 + crow;
        const pf_f32x4 bv = *reinterpret_cast<const pf_f32x4*>(sbias + n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = wm * 32 + i * 16 + frow;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[jt][i][r], a.acc_scale, bv[r]);
            pf_act_n<4>(v, a.act);
            if (!(pf_dbg(a) & 32)) *reinterpret_cast<pf_f32x4*>(orow + (size_t)m * a.outLd + n) = pf_f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}
