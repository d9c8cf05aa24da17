// The Student's hero conv (up2.conv2: 3x3, 128 -> 128 channels on 64 x 64 maps, f32s; DecoderBlock.conv2, model.py:165-172) with the
// weight stream TWO steps ahead (round 4).  Same tiling and arithmetic as conv3x3_halo_split_kernel<128, 4, 2> (k_conv_gemm.h): a
// workgroup of 8 waves owns two image rows (128 pixels) x 128 output channels, the (2 + 2) x 64 input pixels of a 32-channel chunk
// sit in LDS as split hi / lo planes, the nine taps are shifted fragment reads, weights stream per (chunk, tap) by LDS-DMA.  There
// the stream had ONE step of look-ahead and every step ended in __syncthreads() (which drains vmcnt): round 3's ablations priced the
// weight DMA at 0.13 of the kernel's 0.78 ms, and a third 16 KB weight stage did not fit beside the planes (82 944 bytes per
// workgroup against the 81 920 that leave room for two workgroups per CU).  Here
//   * the planes hold the 64 real pixels of a row only (256 rows, 32 KB): the zero padding columns are a per-lane select on the two
//     edge taps (tx == 0 with kx == 0, tx == 63 with kx == 2) instead of 8 stored pixels per plane -- 3 x 16 KB of weight stages
//     now fit: 81 920 bytes exactly;
//   * the 36 steps are unrolled completely; step kt issues the asm LDS-DMA of step kt + 2 (the step in the instruction's immediate
//     offset -- which the hardware adds to the LDS address too, pf_glds16_raw_off), then, at tap 0, the unconditional pixel loads of
//     the next chunk; barriers are raw s_barriers behind "all but the N youngest VMEM operations have completed", N counted per step.
// Host guarantees: W == 64, H * W % 128 == 0, Cpad == 128 (CB == 4), Npad == 128, pad = dil = stride = 1, no gate.
#pragma once
#include "k_conv_gemm.h"

// ONEPROD (round 6, opt-in per conv: ir.py conv(products=1), build_student_program(one_product=...)): ONE f16 product per 32 k -- both operands
// rounded to f16, exact products, f32 accumulation -- instead of the three of the split: no lo planes, no lo weight rows in the stream,
// a third of the MFMAs.  Which layers tolerate it is a property of the network: tools/teacher_precision_study.py --model student
// (profiles/r06_student_precision_study.txt) puts this conv alone at 4.9e-5 of the oracle's landmarks (north star 1e-3).
template <int CB, bool LATE_DMA = true, bool ONEPROD = false>
__global__ __launch_bounds__(512, 4) void conv3x3_hero_kernel(ConvGemmArgs a) {
    constexpr int BN = 128, BM = 128, W = 64, TR = 2, WARPS_M = 4, WARPS_N = 2, NTHR = 512;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N, MT = WM / 16, NT = WN / 16;
    constexpr int XROWS = (TR + 2) * W;                  // 256 pixel rows per plane
    constexpr int PLANE_X = XROWS * 64;
    constexpr int W_BYTES = BN * 128;
    constexpr int NSTG = 3, NKT = 9 * CB;
    constexpr int XU = XROWS * 4 / NTHR;                 // (pixel, 8-channel unit) pairs per thread: 2
    constexpr int WCH = (ONEPROD ? BN * 4 : BN * 8) / NTHR;      // 16-byte weight slots per thread and stage: 2 (hi rows, then lo rows); ONEPROD: the hi rows only
    constexpr int WMID = (NKT / 2) * 128;                // the weight pointers sit in the middle of a row: offsets of +-2304 bytes fit the instruction
    static_assert(MT == 2 && NT == 4 && XU == 2 && WCH == (ONEPROD ? 1 : 2) && 2 * PLANE_X + NSTG * W_BYTES <= 80 * 1024, "tile shape");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PLANE_X + NSTG * W_BYTES];
    unsigned char* xh = smem;
    unsigned char* xl = smem + PLANE_X;
    unsigned char* wbase = smem + 2 * PLANE_X;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave % WARPS_M, wn = wave / WARPS_M;
    int mtile = blockIdx.x;
    if ((gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware tile order
    const int m0 = mtile * BM;
    const int H = a.outH, OHW = H * W, M = a.B * OHW;
    const int face = m0 / OHW, y0 = (m0 - face * OHW) / W;
    const float* __restrict__ in = static_cast<const float*>(a.in) + (size_t)face * OHW * a.inLd;
    const unsigned char* __restrict__ wt = static_cast<const unsigned char*>(a.wt);

    // this thread's pixel units: unconditional loads (rows outside the image read row 0 and are zeroed when they are split)
    const int xc = t & 3;
    unsigned xsrc[XU];                                   // element offsets from `in`
    bool xok[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int hp = (t >> 2) + (NTHR / 4) * u;
        const int iy = y0 - 1 + (hp >> 6);
        xok[u] = (unsigned)iy < (unsigned)H;
        xsrc[u] = (unsigned)(((xok[u] ? iy : 0) * W + (hp & 63)) * a.inLd + xc * 8);
    }
    unsigned wsrc[WCH];                                  // byte offsets from wt (one VGPR each; the base is wave-uniform)
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
        const int sl = t + NTHR * c;
        const int plane = sl >= BN * 4 ? 1 : 0;
        const int row = (sl - plane * BN * 4) >> 2;
        const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
        wsrc[c] = (unsigned)(min(row, a.Npad - 1) * (9 * CB * 128) + plane * 64 + chunk * 16 + WMID);
    }
    pf_f32x4 xreg[XU][2];
    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    auto load_x = [&](auto cb_tag) {
        constexpr int cb = decltype(cb_tag)::value;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            xreg[u][0] = *reinterpret_cast<const pf_f32x4*>(in + xsrc[u] + cb * 32);
            xreg[u][1] = *reinterpret_cast<const pf_f32x4*>(in + xsrc[u] + cb * 32 + 4);
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            pf_half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = xok[u] ? xreg[u][e >> 2][e & 3] : 0.f;
                const pf_half hv = (pf_half)v;
                hi[e] = hv;
                if constexpr (!ONEPROD) lo[e] = pf_split_lo(v, hv);
                amax = pf_amax(amax, v);
            }
            const int off = pf_lds_chunk_off((t >> 2) + (NTHR / 4) * u, xc);
            *reinterpret_cast<pf_half8*>(xh + off) = hi;
            if constexpr (!ONEPROD) *reinterpret_cast<pf_half8*>(xl + off) = lo;
        }
    };
    // weights of step kt (chunk kt / 9, tap kt % 9; K index tap * CB + chunk) -> ring stage kt % NSTG
    auto load_w = [&](auto kt_tag) {
        constexpr int kt = decltype(kt_tag)::value;
        constexpr int kidx = (kt % 9) * CB + kt / 9;
        unsigned char* wdst = wbase + (kt % NSTG) * W_BYTES;
#pragma unroll
        for (int c = 0; c < WCH; ++c) pf_glds16_raw_soff<kidx * 128 - WMID>(wt, wsrc[c], wdst + (t + NTHR * c) * 16);
    };

    pf_f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;
    int row0[MT];                                        // plane row of this lane's pixel at tap (0, 1): (ty, tx)
    bool edge0[MT], edge2[MT];                           // the pixel sits in the first / last column: taps kx == 0 / kx == 2 read padding
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int p = wm * WM + i * 16 + frow;
        row0[i] = p;                                     // ty * 64 + tx
        edge0[i] = (p & 63) == 0;
        edge2[i] = (p & 63) == 63;
    }

    load_w(std::integral_constant<int, 0>{});
    if constexpr (NKT > 1) load_w(std::integral_constant<int, 1>{});
    load_x(std::integral_constant<int, 0>{});
    store_x();                                           // (the compiler waits for the pixel loads: the youngest operations)
    pf_wait_vm_barrier<0>();
    pf_sched_fence();
    pf_static_for<NKT>([&](auto kt_tag) {
        constexpr int kt = decltype(kt_tag)::value;
        constexpr int tap = kt % 9, cb = kt / 9, ky = tap / 3, kx = tap % 3;
        constexpr bool next_x = tap == 0 && cb + 1 < CB;
        constexpr bool restage = tap == 8 && cb + 1 < CB;                           // this step ends with the next chunk's planes being written
        constexpr bool late = restage && LATE_DMA;
        // (the compiler waits for the pixel registers with vmcnt(0) where it splits them: a DMA issued in front of that would be drained
        // on the spot, so the re-staging steps request their weights behind the split)
        if constexpr (kt + 2 < NKT && !late) load_w(std::integral_constant<int, kt + 2>{});
        if constexpr (next_x) load_x(std::integral_constant<int, cb + 1>{});         // AFTER the DMA: younger, may outlive two barriers
        const unsigned char* wh = wbase + (kt % NSTG) * W_BYTES;
        const unsigned char* wl = wh + BN * 64;
        {
            pf_half8 xhf[MT], xlf[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = row0[i] + ky * W + kx - 1;
                const int off = pf_lds_chunk_off(min(max(r, 0), XROWS - 1), fchunk);
                xhf[i] = *reinterpret_cast<const pf_half8*>(xh + off);
                if constexpr (!ONEPROD) xlf[i] = *reinterpret_cast<const pf_half8*>(xl + off);
                if constexpr (kx != 1) {
                    if (kx == 0 ? edge0[i] : edge2[i]) {
                        xhf[i] = pf_half8{0, 0, 0, 0, 0, 0, 0, 0};
                        if constexpr (!ONEPROD) xlf[i] = xhf[i];
                    }
                }
            }
            // weight fragments of tile j + 1 requested in front of tile j's MFMAs and no further ahead (compiler fence): the scheduler of
            // this one huge basic block otherwise hoists every tile's reads and spills
            pf_half8 wq[2][2];
            {
                const int off = pf_lds_chunk_off(wn * WN + frow, fchunk);
                wq[0][0] = *reinterpret_cast<const pf_half8*>(wh + off);
                if constexpr (!ONEPROD) wq[0][1] = *reinterpret_cast<const pf_half8*>(wl + off);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j + 1 < NT) {
                    const int off = pf_lds_chunk_off(wn * WN + (j + 1) * 16 + frow, fchunk);
                    wq[(j + 1) & 1][0] = *reinterpret_cast<const pf_half8*>(wh + off);
                    if constexpr (!ONEPROD) wq[(j + 1) & 1][1] = *reinterpret_cast<const pf_half8*>(wl + off);
                }
                const pf_half8 whf = wq[j & 1][0];
                if constexpr (!ONEPROD) {
                    const pf_half8 wlf = wq[j & 1][1];
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(wlf, xhf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xlf[i], acc[j][i]);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xhf[i], acc[j][i]);
                asm volatile("" ::: "memory");
            }
        }
        // VMEM operations younger than step kt + 1's weights at this point: this step's DMA (WCH), and the next chunk's pixel loads
        // (2 XU) while they are younger than those weights, i.e. in the steps of taps 0 and 1
        constexpr int dma = kt + 2 < NKT ? WCH : 0;
        constexpr int keep = dma + (((tap == 0 || tap == 1) && cb + 1 < CB) ? 2 * XU : 0);
        if constexpr (restage) {
            pf_wait_vm_barrier<late ? 0 : keep>();       // every wave is done with this chunk's planes
            store_x();
            pf_pin(amax);
            if constexpr (kt + 2 < NKT && late) load_w(std::integral_constant<int, kt + 2>{});
        }
        pf_wait_vm_barrier<keep>();
        pf_sched_fence();
    });
    pf_amax_commit(a.range_slot, amax, amax_seen);
    conv_gemm_epilogue<float, BM, BN, WARPS_M, WARPS_N>(a, acc, m0, 0, wm, wn, lane, M, OHW, a.acc_scale);
}
