// Fused DecoderBlock front end (model.py:184-189 + SeparableConv2d :15-43), round-3 kernel: bilinear x2 upsample of the
// low-res map + concat with the skip connection + depthwise 3x3 (+BN) as the PRODUCER of a split-precision pointwise
// GEMM (3 x v_mfma_f32_16x16x32_f16 per product, f32 accumulate) -- the operator of sepup_patch_kernel
// (k_conv_gemm.h), restructured around what that kernel's profile showed (profiles/r02_run2_hero_timing_ablations.md):
// 0.36 of its 0.65 ms was "skeleton", a serial chain per 128-pixel workgroup of prologue, nine two-barrier K steps that
// each waited for an LDS-DMA issued inside the same step, a skip-connection step made of dependent global loads, and an
// epilogue -- overlapped only by the one other workgroup resident on the CU; no unit was above 45 % busy.
//
// Here ONE persistent workgroup of 16 waves owns a CU and the two halves of the work run side by side instead of in turns:
//   * waves {0-3, 8-11} are PRODUCERS (two per SIMD): per K step (32 channels) they evaluate the collapsed
//     upsample (x) depthwise filter -- one 3x3 filter on the low-res patch with position-class weights,
//     ir.py::sepconv_up -- for the tile's 128 pixels out of LDS, split the result into f16 hi / lo and park it in one of
//     two pixel-operand stages;
//   * waves {4-7, 12-15} are CONSUMERS (two per SIMD): they run the 128 x BN x 32 MFMA step on the stage the producers
//     filled one iteration earlier, and the bias / activation / store epilogue after a tile's last step;
//   * ONE barrier per K step; everything that comes from memory is requested one or two steps ahead by LDS-DMA issued from
//     inline asm (pf_glds16_raw) and retired by partial vmcnt waits at that barrier (the recipe of k_chain.h): the
//     low-res patch rows (border replication = clamped SOURCE addresses), the position-class filters of the chunk, and
//     the pointwise weights; the rings run on across tile boundaries, so a tile's prologue hides behind the previous
//     tile's last steps;
//   * the skip-connection channels (24 / 40 of 280 / 296) never touch the producers' global-load path: a small kernel
//     (sepup_skip_kernel) runs their plain depthwise 3x3 once and writes the result pre-split in the exact byte layout of
//     a pixel-operand stage, so those K steps are a DMA plus an LDS-to-LDS copy;
//   * the depthwise bias is folded into the pointwise bias at pack time (the two convs are separated by an affine BN
//     only): no per-step bias traffic;
//   * patch pixels are 128 B apart; the 16-byte slot index is XOR-ed with bit 1 of the pixel index so that the four
//     patch pixels one ds_read_b128 lane group touches land on disjoint banks (the old kernel lost a quarter of its LDS
//     cycles to that conflict, r01_run26_pmc_sq_sepup_patch_kernel.json);
//   * workgroup b serves XCD b & 7 and walks the tiles of faces x, x + 8, ... so one XCD's L2 holds one face's inputs.
// Host guarantees (engine.cpp): W in {16, 32, 64}, (H * W) % 128 == 0, at least two tiles per face, C1 % 32 == 0,
// C2 % 8 == 0, N == Npad in {128, 256}.
#pragma once
#include "pf_common.h"
#include "k_conv_gemm.h"

struct SepupArgs {
    const float* lo;            // [B][H/2][W/2][loLd]    channels [0, C1): upsampled
    const float* skip;          // [B][H][W][skipLd]      channels [C1, C1 + C2)
    float* out;                 // [B][H][W][outLd]
    const float* dw_lo;         // [9][C1]                plain depthwise weights of the upsampled channels (BN folded)
    const float* dw_v;          // [4][9][C1]             VCOL: the same filters with the VERTICAL half of the upsample folded in, per row class
                                //                        (first / last / even / odd): V[cls][j][kx] = sum_ky A_cls[ky][j] w[ky][kx] (ir.py::sepconv_up)
    const float* dw_w2;         // [9][C2]                plain depthwise weights of the skip channels
    const unsigned char* wt;    // [Npad][Cpad/32][hi 32 x f16 | lo 32 x f16]  pointwise weights, pre-split
    const float* bias;          // [Npad]                 pointwise bias with the depthwise bias folded in
    unsigned char* skipx;       // [B * tiles per face][skip chunks][16 KB]  pre-split depthwise output of the skip channels
    int B, H, C1, C2, loLd, skipLd, outLd;
    int N, Cpad, act;
    float acc_scale;
    float* gap_part;            // [B * tiles per face][N] or nullptr: per-tile channel sums of the ACTIVATED output (the squeeze of the SCSE block
                                //                        that follows, model.py:117-130): the consumers have every output of a tile in registers
    unsigned* range_slot;       // f32s range guard: max |v| (raw bits) over the depthwise outputs both kernels split, or nullptr
    unsigned long long* prof;   // dbg & 64: per-role cycle totals {producer: work, barrier wait | consumer: dma issue, mfma, epilogue, barrier wait} + wave counts
    int dbg;                    // timing ablations (1 no weight refresh, 2 no patch/filter refresh, 4 no producer taps, 8 no MFMAs, 16 no stores)
};

// ---- depthwise 3x3 of the skip-connection channels, written as ready-made pixel-operand stages -------------------------
// One workgroup per 128-pixel tile.  The tile's rows plus one halo row above and below (zero outside the image) and a
// zero column either side go through LDS once, as 16-byte units read in address order (an NHWC row of the skip tensor is
// contiguous), so a tile costs ~5 memory instructions per wave instead of the 38 of a per-thread gather; the nine taps and
// the weights are then LDS reads.  Thread = (pixel of the tile, 8 channels).  Stage layout = what the GEMM reads: hi plane
// (128 rows x 64 B) then lo plane, 16-byte chunks rotated per row (pf_lds_chunk_off); one stage per 32-channel chunk.
// Host guarantees: C2 % 8 == 0, C2 <= MAXC (32 or 64: the LDS tile, hence the workgroups per CU, follow the channel count).
template <int W, int MAXC>
__global__ __launch_bounds__(512) void sepup_skip_kernel(SepupArgs a) {
    constexpr int TR = 128 / W;
    __shared__ __attribute__((aligned(16))) float tile[(TR + 2) * (W + 2) * MAXC];
    __shared__ __attribute__((aligned(16))) float wsm[9 * MAXC];
    const int H = a.H, C2 = a.C2;
    const int TPF = H * W / 128;
    const int tb = blockIdx.x;
    const int face = tb / TPF, y0 = (tb - face * TPF) * TR;
    const int t = threadIdx.x;
    const int upp = C2 >> 2;                                         // 16-byte units per pixel
    const int row_units = W * upp;
    const float* __restrict__ sp = a.skip + (size_t)face * H * W * a.skipLd;
    for (int u = t; u < (TR + 2) * row_units; u += 512) {            // interior pixels of the TR + 2 rows
        const int r = u / row_units, q = u - r * row_units;
        const int px = q / upp, part = q - px * upp;
        const int yy = y0 - 1 + r;
        pf_f32x4 v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H) v = *reinterpret_cast<const pf_f32x4*>(sp + ((size_t)yy * W + px) * a.skipLd + part * 4);
        *reinterpret_cast<pf_f32x4*>(tile + ((r * (W + 2) + px + 1) * C2 + part * 4)) = v;
    }
    for (int u = t; u < (TR + 2) * 2 * upp; u += 512) {              // the two zero columns
        const int r = u / (2 * upp), q = u - r * 2 * upp;
        const int side = q / upp, part = q - side * upp;
        *reinterpret_cast<pf_f32x4*>(tile + ((r * (W + 2) + side * (W + 1)) * C2 + part * 4)) = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = t; i < 9 * C2; i += 512) wsm[i] = a.dw_w2[i];
    __syncthreads();
    const int xc = t & 3, prow = t >> 2;
    const int ty = prow / W, px = prow - ty * W;
    const int nskip = (C2 + 31) >> 5;
    unsigned amax = 0;                                               // range guard (pf_common.h)
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    for (int sc = 0; sc < nskip; ++sc) {
        const int c0 = sc * 32 + xc * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        if (c0 < C2) {
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2) {
                    const float* pp = tile + ((ty + k1) * (W + 2) + px + k2) * C2 + c0;
                    const float* ww = wsm + (k1 * 3 + k2) * C2 + c0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const pf_f32x4 v4 = *reinterpret_cast<const pf_f32x4*>(pp + 4 * h);
                        const pf_f32x4 w4 = *reinterpret_cast<const pf_f32x4*>(ww + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[4 * h + e] = fmaf(w4[e], v4[e], o[4 * h + e]);
                    }
                }
        }
        pf_half8 hi, lo8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const pf_half hv = (pf_half)o[e];
            hi[e] = hv;
            lo8[e] = pf_split_lo(o[e], hv);
            amax = pf_amax(amax, o[e]);
        }
        unsigned char* dst = a.skipx + ((size_t)tb * nskip + sc) * 16384 + pf_lds_chunk_off(prow, xc);
        *reinterpret_cast<pf_half8*>(dst) = hi;
        *reinterpret_cast<pf_half8*>(dst + 8192) = lo8;
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
}

// position class of a row / column: 0 first, 1 last, 2 even, 3 odd (ir.py::sepconv_up builds the filters in this order)
__device__ __forceinline__ int pf_pos_class(int v, int n) { return v == 0 ? 0 : (v == n - 1 ? 1 : 2 + (v & 1)); }

// P_BY_CONS: the consumer waves issue the patch / filter requests too (the producers' step is the longer one at BN = 128: 774 cycles
// of request issue + 1 600 of work against 388 + 765 + 754, profiles/r05_run51_sepup_roles.txt).  BIAS_REG: the pointwise bias lives in
// the consumers' registers instead of LDS -- the 512 bytes that keep a fourth ring stage (D = 4) from fitting beside BN = 128.
// VCOL (round 6): the producers' step is bound by its own VALU instruction stream (with requests, patch reads, MFMAs and stores all
// compiled out the launch still takes 64 % of its time: profiles/r06_run5_ub_sepup_ablate.txt), 92 packed operations per thread of which
// 32 only interpolate vertically.  The vertical bilinear weights and the depthwise filter's zero padding above / below the image are
// linear in the filter, so they fold into it at pack time: out[dy][dx] = sum_j sum_kx V[row class of dy][j][kx] * hrow[j][dx + kx] on the
// horizontally interpolated patch rows -- 36 fma as before, no `u` window (60 packed operations), twice the filter words per chunk
// (one 3 x 3 set per image row of the tile, chosen by the row's class when the tile's requests are set up).
template <int BN, int W, int D, bool W_BY_PROD, bool DEFER, bool P_BY_CONS = false, bool BIAS_REG = false, bool VCOL = false>
__global__ __launch_bounds__(1024, 4) void sepup_pipe_kernel(SepupArgs a) {
    constexpr int TR = 128 / W;                          // image rows per tile
    constexpr int PC = W / 2 + 2;                        // patch columns (one replicated column each side)
    constexpr int PP = (TR / 2 + 2) * PC;                // patch pixels
    constexpr int BCOLS = W / 2;                         // 2 x 2 output blocks per tile row pair
    constexpr int PATCH_SLOTS = PP * 8;                  // 16-byte slots
    constexpr int FILT_SLOTS = (VCOL ? TR : 1) * 9 * 8;  // the chunk's depthwise weights, [tap][32 channels] (VCOL: one set per image row of the tile)
    constexpr int PI = 2;                                // patch-stage DMA wave-instructions per producer wave and stage
    constexpr int P_INSTR = 8 * PI;
    constexpr int P_BYTES = P_INSTR * 1024;              // 16 KB: also exactly one ready-made skip-chunk operand
    constexpr int X_BYTES = 16384;                       // pixel-operand stage: hi plane + lo plane of 128 rows x 64 B
    constexpr int W_BYTES = BN * 128;                    // weight stage: hi rows then lo rows
    constexpr int WI = BN * 8 / 512;                     // weight DMA wave-instructions per wave (of the issuing role) and stage
    constexpr int WN = BN / 2, NT = WN / 16;             // consumers: 4 (pixels) x 2 (channels) waves, 32 x WN per wave
    constexpr int NV = NT * 2;                           // output vectors (4 channels of one pixel) per consumer lane and tile
    constexpr int EXP_P = (P_BY_CONS ? 0 : PI) + (W_BY_PROD ? WI : 0);      // requests per iteration of a producer / consumer wave in steady state
    constexpr int EXP_C = (P_BY_CONS ? PI : 0) + (W_BY_PROD ? 0 : WI);
    constexpr int BIAS_BYTES = BIAS_REG ? 0 : BN * 4;
    constexpr bool GAP_OK = !DEFER && W_BY_PROD;         // channel sums of the output (a.gap_part): the instance whose consumers count no vmcnt
    constexpr int GAP_BYTES = GAP_OK ? 4 * BN * 4 : 0;   // [pixel quarter wm][channel]
    static_assert(PATCH_SLOTS + FILT_SLOTS <= P_INSTR * 64 && 1024 <= P_INSTR * 64, "patch/filter stage");
    static_assert(D >= 2 && (D - 2) * (PI + WI) + 1 < 63, "ring depth");
    static_assert(2 * X_BYTES + D * W_BYTES + D * P_BYTES + BIAS_BYTES + GAP_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * X_BYTES + D * W_BYTES + D * P_BYTES + BIAS_BYTES + GAP_BYTES];
    unsigned char* const xbase = smem;
    unsigned char* const wbase = smem + 2 * X_BYTES;
    unsigned char* const pbase = wbase + D * W_BYTES;
    float* const sbias = reinterpret_cast<float*>(pbase + D * P_BYTES);
    float* const gsum = reinterpret_cast<float*>(pbase + D * P_BYTES + BIAS_BYTES);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const bool producer = ((wave >> 2) & 1) == 0;        // waves w, w + 4, w + 8, w + 12 share a SIMD: two of each role there
    const int rw = ((wave >> 3) << 2) | (wave & 3);      // wave index within the role, 0..7
    const int rt = rw * 64 + lane;                       // thread index within the role, 0..511

    // ---- this workgroup's tiles ----------------------------------------------------------------------------------------
    const int H = a.H, loH = H >> 1;
    constexpr int loW = W / 2;
    const int TPF = H * W / 128;
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, L = gridDim.x >> 3;
    const int nfaces = (a.B - xcd + 7) >> 3;             // faces xcd, xcd + 8, ...
    const int ntile_x = nfaces * TPF;
    if (wl >= ntile_x) return;
    const int nt = (ntile_x - wl + L - 1) / L;           // tiles wl, wl + L, ... of this XCD's list
    const int NK = a.Cpad >> 5, lo_chunks = a.C1 >> 5, nskip = NK - lo_chunks;
    const int S = nt * NK;                               // K steps of this workgroup, all tiles
    const size_t wrow_bytes = (size_t)NK * 128;
    const bool prof = (pf_dbg(a) & 64) != 0;

    if constexpr (!BIAS_REG) { for (int i = t; i < BN; i += 1024) sbias[i] = a.bias[i]; }

    // tile j of this workgroup -> (face, first row, global tile id)
    auto tile_of = [&](int j, int& face, int& y0, int& gt) {
        const int q = wl + j * L;
        const int k = q / TPF, s = q - k * TPF;
        face = xcd + 8 * k;
        y0 = s * TR;
        gt = face * TPF + s;
    };
    // ---- the two request streams (LDS-DMA from inline asm; either role can issue them with its role-local thread index) -------
    // patch + position-class filters (or the ready-made skip-chunk operand) of step (dj, dcb) -> P ring stage
    const float* dsrc[PI];                                          // per-lane source of this wave's requests, current DMA tile
    int dadv[PI];                                                   // floats the source advances per channel chunk (0 for padding slots)
    int dj = 0, dcb = 0, dgt = 0;                                   // DMA stream position: tile index, chunk, global tile id
    auto dma_tile = [&](int j) {
        int face, y0, gt;
        tile_of(j, face, y0, gt);
        dgt = gt;
        const int rmin = (y0 >> 1) - 1;
        const float* lo_face = a.lo + (size_t)face * loH * loW * a.loLd;
#pragma unroll
        for (int k = 0; k < PI; ++k) {
            const int s = (rw + 8 * k) * 64 + lane;
            if (s < PATCH_SLOTS) {
                const int pp = s >> 3, sl = s & 7;
                const int pr = pp / PC, pc = pp - pr * PC;
                const int ry = min(max(rmin + pr, 0), loH - 1), rx = min(max(pc - 1, 0), loW - 1);
                dsrc[k] = lo_face + (size_t)(ry * loW + rx) * a.loLd + (sl << 2);
                dadv[k] = 32;
            } else if (s < PATCH_SLOTS + FILT_SLOTS) {
                const int fs = s - PATCH_SLOTS;
                if constexpr (VCOL) {
                    const int idx = fs >> 3, r = idx / 9, jk = idx - r * 9;          // image row r of the tile, filter word (j, kx)
                    dsrc[k] = a.dw_v + (size_t)(pf_pos_class(y0 + r, H) * 9 + jk) * a.C1 + ((fs & 7) << 2);
                } else {
                    dsrc[k] = a.dw_lo + (size_t)(fs >> 3) * a.C1 + ((fs & 7) << 2);
                }
                dadv[k] = 32;
            } else {
                dsrc[k] = a.dw_lo;                                  // padding slots of the stage: any valid address
                dadv[k] = 0;
            }
        }
    };
    auto dma_issue = [&](int stage) {                               // requests the stage of step (dj, dcb); advances the stream
        unsigned char* dst = pbase + stage * P_BYTES;
        const bool lo_step = dcb < lo_chunks;
        const unsigned char* skp = a.skipx + ((size_t)dgt * nskip + (dcb - lo_chunks)) * 16384;
#pragma unroll
        for (int k = 0; k < PI; ++k) {
            const int s = (rw + 8 * k) * 64 + lane;
            const void* src = lo_step ? (const void*)(dsrc[k] + dcb * dadv[k]) : (const void*)(skp + (size_t)s * 16);
            pf_glds16_raw(src, dst + (size_t)s * 16);
        }
        if (++dcb == NK) { dcb = 0; ++dj; if (dj < nt) dma_tile(dj); }
    };
    auto w_issue = [&](int step) {                                  // pointwise weights of K step `step` (chunk = step mod NK)
        const int cb = step % NK;
        unsigned char* dst = wbase + (step % D) * W_BYTES;
#pragma unroll
        for (int c = 0; c < WI; ++c) {
            const int sl = rt + 512 * c;
            const int plane = sl >= BN * 4 ? 1 : 0;
            const int row = (sl - plane * BN * 4) >> 2;
            const int chunk = ((sl & 3) - 2 * (row >> 2)) & 3;
            pf_glds16_raw(a.wt + (size_t)row * wrow_bytes + (size_t)cb * 128 + plane * 64 + chunk * 16, dst + (size_t)sl * 16);
        }
    };
    int issued_p = 0, issued_w = 0;                                 // stages requested so far (meaningful in the issuing role only)

    if (producer) {
        // =================================================================================================================
        // PRODUCERS
        // =================================================================================================================
        // unit of work: a 2 x 2 block of output pixels (one of each parity class) x 2 channels; 32 blocks x 16 channel pairs
        const int blk = rt >> 4, cp = rt & 15;
        const int brow = blk / BCOLS, bn = blk - brow * BCOLS;
        const int pp0 = brow * PC + bn;                              // patch pixel of low-res (row m - 1, column n - 1)
        // pixel-operand rows of the block's four pixels, byte offset of this thread's two channels inside them
        int xoff[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                xoff[dy][dx] = pf_lds_chunk_off((2 * brow + dy) * W + 2 * bn + dx, cp >> 2) + (cp & 3) * 4;
        const float ml = bn == 0 ? 0.f : 1.f, mr = bn == BCOLS - 1 ? 0.f : 1.f;     // upsampled columns -1 / W are padding
        if constexpr (!P_BY_CONS) dma_tile(0);
#pragma unroll
        for (int k = 0; k < D - 1; ++k) {
            if (!P_BY_CONS && issued_p < S) { dma_issue(issued_p % D); ++issued_p; }
            if (W_BY_PROD && issued_w < S) { w_issue(issued_w); ++issued_w; }
        }
        if (EXP_P > 0) pf_wait_vm_barrier<(D - 2) * EXP_P>();       // stage 0 has landed (S >= NK >= D - 1 always holds)
        else pf_wait_vm_barrier<63>();                              // no requests of its own: nothing but the rendezvous

        int pj = 0, pcb = 0;                                        // produce position
        float mt = 1.f, mb = 1.f;                                    // upsampled rows -1 / H are padding
        auto prod_tile = [&](int j) {
            int face, y0, gt;
            tile_of(j, face, y0, gt);
            mt = (y0 + 2 * brow == 0) ? 0.f : 1.f;
            mb = (y0 + 2 * brow + 2 >= H) ? 0.f : 1.f;
        };
        prod_tile(0);
        unsigned amax = 0;                                           // range guard (pf_common.h)
        const unsigned amax_seen = pf_amax_seen(a.range_slot);
        unsigned long long t_dma = 0, t_work = 0, t_wait = 0;
        for (int g = 0; g <= S; ++g) {
            const unsigned long long c0 = prof ? pf_clock() : 0;
            // requests of this iteration: the stage read D - 1 steps from now and (W_BY_PROD) the weights consumed D - 1 iterations from now
            int nreq = 0;
            if (!P_BY_CONS && issued_p < S && !(pf_dbg(a) & 2)) { dma_issue(issued_p % D); ++issued_p; nreq += PI; }
            if (W_BY_PROD && g >= 1 && issued_w < S && !(pf_dbg(a) & 1)) { w_issue(issued_w); ++issued_w; nreq += WI; }
            const unsigned long long c1 = prof ? pf_clock() : 0;
            if (g < S) {
                const unsigned char* pst = pbase + (g % D) * P_BYTES;
                unsigned char* xdst = xbase + (g & 1) * X_BYTES;
                if (pcb < lo_chunks) {
                    // bilinear x2 (align_corners = False: taps 0.25 / 0.75, borders replicated in the patch) of the 3 x 3
                    // low-res neighbourhood -> the 4 x 4 upsampled window around the block, rows / columns outside the
                    // image zeroed (the depthwise conv's padding) -> four depthwise 3 x 3 results; two channels at a time
                    const unsigned char* pa = pst + pp0 * 128 + cp * 8;
                    const unsigned char* wa = pst + PATCH_SLOTS * 16 + cp * 8;
                    pf_f32x2 hrow[3][4];
                    {
                        const float l0 = 0.75f * ml, l1 = 0.25f * ml, r0 = 0.25f * mr, r1 = 0.75f * mr;
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            pf_f32x2 p0 = pf_f32x2{0.f, 0.f}, p1 = p0, p2 = p0;
                            if (!(pf_dbg(a) & 4)) {
                                p0 = *reinterpret_cast<const pf_f32x2*>(pa + (r * PC + 0) * 128);
                                p1 = *reinterpret_cast<const pf_f32x2*>(pa + (r * PC + 1) * 128);
                                p2 = *reinterpret_cast<const pf_f32x2*>(pa + (r * PC + 2) * 128);
                            }
                            hrow[r][0] = p0 * l0 + p1 * l1;
                            hrow[r][1] = p0 * 0.25f + p1 * 0.75f;
                            hrow[r][2] = p1 * 0.75f + p2 * 0.25f;
                            hrow[r][3] = p1 * r0 + p2 * r1;
                        }
                    }
                    pf_f32x2 o[2][2];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) o[dy][dx] = pf_f32x2{0.f, 0.f};
                    if constexpr (VCOL) {
                        // the vertical interpolation (and the zero rows above / below the image) sit in the filter: one set per image row
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) {
                            const unsigned char* wr = wa + (2 * brow + dy) * (9 * 128);
#pragma unroll
                            for (int j = 0; j < 3; ++j)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx) {
                                    const pf_f32x2 wv = *reinterpret_cast<const pf_f32x2*>(wr + (j * 3 + kx) * 128);
#pragma unroll
                                    for (int dx = 0; dx < 2; ++dx) o[dy][dx] += wv * hrow[j][dx + kx];
                                }
                        }
                    } else {
                        pf_f32x2 u[4][4];
                        const float t0 = 0.75f * mt, t1 = 0.25f * mt, b0 = 0.25f * mb, b1 = 0.75f * mb;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            u[0][c] = hrow[0][c] * t0 + hrow[1][c] * t1;
                            u[1][c] = hrow[0][c] * 0.25f + hrow[1][c] * 0.75f;
                            u[2][c] = hrow[1][c] * 0.75f + hrow[2][c] * 0.25f;
                            u[3][c] = hrow[1][c] * b0 + hrow[2][c] * b1;
                        }
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const pf_f32x2 wv = *reinterpret_cast<const pf_f32x2*>(wa + (ky * 3 + kx) * 128);
#pragma unroll
                                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                                    for (int dx = 0; dx < 2; ++dx) o[dy][dx] += wv * u[dy + ky][dx + kx];
                            }
                    }
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            pf_half2 hi, lo2;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const pf_half hv = (pf_half)o[dy][dx][e];
                                hi[e] = hv;
                                lo2[e] = pf_split_lo(o[dy][dx][e], hv);
                                amax = pf_amax(amax, o[dy][dx][e]);
                            }
                            *reinterpret_cast<pf_half2*>(xdst + xoff[dy][dx]) = hi;
                            *reinterpret_cast<pf_half2*>(xdst + 8192 + xoff[dy][dx]) = lo2;
                        }
                } else {                                            // skip-connection chunk: the stage IS the pixel operand
                    const pf_f32x4 c0v = *reinterpret_cast<const pf_f32x4*>(pst + rt * 16);
                    const pf_f32x4 c1v = *reinterpret_cast<const pf_f32x4*>(pst + 8192 + rt * 16);
                    *reinterpret_cast<pf_f32x4*>(xdst + rt * 16) = c0v;
                    *reinterpret_cast<pf_f32x4*>(xdst + 8192 + rt * 16) = c1v;
                }
                if (++pcb == NK) { pcb = 0; ++pj; if (pj < nt) prod_tile(pj); }
            }
            const unsigned long long c2 = prof ? pf_clock() : 0;
            // end of step g: what was requested D - 2 iterations ago and earlier must have landed
            if (EXP_P == 0) pf_wait_vm_barrier<63>();
            else if (nreq == EXP_P) pf_wait_vm_barrier<(D - 2) * EXP_P>();
            else pf_wait_vm_barrier<0>();
            if (prof) { t_dma += c1 - c0; t_work += c2 - c1; t_wait += pf_clock() - c2; }
        }
        pf_amax_commit(a.range_slot, amax, amax_seen);
        if (prof && lane == 0) {
            atomicAdd(a.prof + 0, t_work); atomicAdd(a.prof + 1, t_wait); atomicAdd(a.prof + 2, 1ull);
            atomicAdd(a.prof + 3, (unsigned long long)S); atomicAdd(a.prof + 9, t_dma);
        }
    } else {
        // =================================================================================================================
        // CONSUMERS
        // =================================================================================================================
        const int wm = rw & 3, wn = rw >> 2;
        const int frow = lane & 15, fchunk = lane >> 4;
        const int crow = fchunk * 4;
        // the matrix waves go first whenever they can issue: each of their instructions keeps the matrix pipe busy for 8-16 cycles
        // while the producers' VALU work fills the slots in between -- the other way round the producers starve them of issue slots
        // (0.548 -> 0.528 ms per 256 faces at 64 x 64; producers first: 0.56; profiles/r05_run27_wave_priority_ab3.txt)
        pf_setprio<3>();
        pf_f32x4 breg[BIAS_REG ? NT : 1];                           // BIAS_REG: this lane's NT x 4 bias values, fetched before any request is in flight
        if constexpr (BIAS_REG) {
#pragma unroll
            for (int j = 0; j < NT; ++j) breg[j] = *reinterpret_cast<const pf_f32x4*>(a.bias + wn * WN + j * 16 + crow);
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(breg[j]));        // (landed: the compiler waits for them here, not inside the ring)
        }
        if constexpr (P_BY_CONS) dma_tile(0);
#pragma unroll
        for (int k = 0; k < D - 1; ++k) {
            if (P_BY_CONS && issued_p < S) { dma_issue(issued_p % D); ++issued_p; }
            if (!W_BY_PROD && issued_w < S) { w_issue(issued_w); ++issued_w; }
        }
        if (EXP_C > 0) pf_wait_vm_barrier<(D - 2) * EXP_C>();
        else pf_wait_vm_barrier<63>();                              // no requests of its own: nothing but the rendezvous

        pf_f32x4 acc[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[j][0] = pf_f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = pf_f32x4{0.f, 0.f, 0.f, 0.f}; }
        // DEFER: a finished tile's output vectors wait in registers and leave one per K step of the next tile, so the
        // stores' issue cost (the same texture-path queue the DMA requests go through) never lands in one iteration
        pf_f32x4 pend[DEFER ? NV : 1];
        float* pend_row = nullptr;
        int pend_left = 0;
        auto store_vec = [&](float* orow, int k, pf_f32x4 v) {      // vector k = (channel tile j = k >> 1, pixel half i = k & 1)
            const int n = wn * WN + (k >> 1) * 16 + crow;
            const int m = wm * 32 + (k & 1) * 16 + frow;
            if (n < a.N && !(pf_dbg(a) & 16)) *reinterpret_cast<pf_f32x4*>(orow + (size_t)m * a.outLd + n) = v;
        };
        int cj = 0, ccb = 0;                                        // consume position
        int gap_gt = -1;                                            // GAP_OK: global tile whose partial sums wait in LDS
        const int xoff0 = pf_lds_chunk_off(wm * 32 + frow, fchunk), xoff1 = pf_lds_chunk_off(wm * 32 + 16 + frow, fchunk);
        unsigned long long t_dma = 0, t_mma = 0, t_epi = 0, t_wait = 0;
        for (int g = 0; g <= S; ++g) {
            const unsigned long long c0 = prof ? pf_clock() : 0;
            unsigned long long c1 = c0, c2 = c0;
            int nreq = 0;
            bool stored = false;                                    // did this step issue a pending-vector store behind its weight requests?
            if constexpr (GAP_OK) {
                // the channel sums of the tile finished in the previous iteration: its four pixel quarters were parked in LDS before that
                // iteration's barrier; added here in a fixed order, one channel per consumer thread
                if (gap_gt >= 0) {
                    if (rt < BN) a.gap_part[(size_t)gap_gt * a.N + rt] = (gsum[rt] + gsum[BN + rt]) + (gsum[2 * BN + rt] + gsum[3 * BN + rt]);
                    gap_gt = -1;
                }
            }
            if (P_BY_CONS && issued_p < S && !(pf_dbg(a) & 2)) { dma_issue(issued_p % D); ++issued_p; nreq += PI; }
            if (g >= 1) {
                const int c = g - 1;
                if (!W_BY_PROD && issued_w < S && !(pf_dbg(a) & 1)) { w_issue(issued_w); ++issued_w; nreq += WI; }
                if (prof) c1 = pf_clock();
                const unsigned char* xs = xbase + (c & 1) * X_BYTES;
                const unsigned char* wh = wbase + (c % D) * W_BYTES;
                const unsigned char* wlp = wh + BN * 64;
                pf_half8 xhf[2], xlf[2];
                xhf[0] = *reinterpret_cast<const pf_half8*>(xs + xoff0);
                xlf[0] = *reinterpret_cast<const pf_half8*>(xs + 8192 + xoff0);
                xhf[1] = *reinterpret_cast<const pf_half8*>(xs + xoff1);
                xlf[1] = *reinterpret_cast<const pf_half8*>(xs + 8192 + xoff1);
                if (!(pf_dbg(a) & 8))
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int off = pf_lds_chunk_off(wn * WN + j * 16 + frow, fchunk);
                    const pf_half8 whf = *reinterpret_cast<const pf_half8*>(wh + off);
                    const pf_half8 wlf = *reinterpret_cast<const pf_half8*>(wlp + off);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = pf_mfma_16x16x32_f16(wlf, xhf[i], acc[j][i]);     // small terms first
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xlf[i], acc[j][i]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = pf_mfma_16x16x32_f16(whf, xhf[i], acc[j][i]);
                }
                if (prof) c2 = pf_clock();
                if (DEFER && pend_left > 0) {                       // one pending vector of the previous tile per step
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        if (pend_left == NV - k) store_vec(pend_row, k, pend[DEFER ? k : 0]);
                    --pend_left;
                    stored = true;
                }
                if (++ccb == NK) {                                  // the tile is complete: bias, activation, store (or park)
                    int face, y0, gt;
                    tile_of(cj, face, y0, gt);
                    float* orow = a.out + (size_t)gt * 128 * a.outLd;
                    if (DEFER && pend_left > 0) {                   // (NK < NV never happens: NK >= 9, NV = 8)
#pragma unroll
                        for (int k = 0; k < NV; ++k)
                            if (k >= NV - pend_left) store_vec(pend_row, k, pend[DEFER ? k : 0]);
                        pend_left = 0;
                    }
                    const bool want_gap = GAP_OK && a.gap_part != nullptr;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const pf_f32x4 bv = BIAS_REG ? breg[BIAS_REG ? j : 0] : *reinterpret_cast<const pf_f32x4*>(sbias + wn * WN + j * 16 + crow);
                        pf_f32x4 gs = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[j][i][r], a.acc_scale, bv[r]);
                            pf_act_n<4>(v, a.act);
                            const pf_f32x4 ov = pf_f32x4{v[0], v[1], v[2], v[3]};
                            if (DEFER) pend[DEFER ? j * 2 + i : 0] = ov;
                            else store_vec(orow, j * 2 + i, ov);
                            gs += ov;
                            acc[j][i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        if constexpr (GAP_OK) {
                            if (want_gap) {         // this wave's 32 pixels of channels n .. n + 3: the 16 lanes of a row hold 16 pixel pairs
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float q = gs[r];
                                    q += pf_row_xchg_f32<0>(q); q += pf_row_xchg_f32<1>(q); q += pf_row_xchg_f32<2>(q); q += pf_row_xchg_f32<3>(q);
                                    gs[r] = q;
                                }
                                if (frow == 0) *reinterpret_cast<pf_f32x4*>(gsum + wm * BN + wn * WN + j * 16 + crow) = gs;
                            }
                        }
                    }
                    if constexpr (GAP_OK) { if (want_gap) gap_gt = gt; }
                    if (DEFER) { pend_row = orow; pend_left = NV; }
                    ccb = 0;
                    ++cj;
                }
            }
            const unsigned long long c3 = prof ? pf_clock() : 0;
            if (EXP_C == 0) pf_wait_vm_barrier<63>();
            // The wait must retire everything OLDER than the (D - 2) steps of weight requests that may stay in flight.  A step that
            // also issued a pending-vector store (younger than its requests) may leave one more operation outstanding; a step that
            // did not -- every step of a workgroup's first tile, steps past the eighth of later tiles -- must not, or the count
            // would cover the previous step's second request, whose lo-plane weights are read right after this barrier.
            else if (nreq == EXP_C && stored) pf_wait_vm_barrier<(D - 2) * EXP_C + 1>();
            else if (nreq == EXP_C) pf_wait_vm_barrier<(D - 2) * EXP_C>();
            else pf_wait_vm_barrier<0>();
            if (prof) { t_dma += c1 - c0; t_mma += c2 - c1; t_epi += c3 - c2; t_wait += pf_clock() - c3; }
        }
        if (DEFER && pend_left > 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (k >= NV - pend_left) store_vec(pend_row, k, pend[DEFER ? k : 0]);
        }
        if constexpr (GAP_OK) {                                     // the last tile's channel sums (parked before the loop's final barrier)
            if (gap_gt >= 0 && rt < BN) a.gap_part[(size_t)gap_gt * a.N + rt] = (gsum[rt] + gsum[BN + rt]) + (gsum[2 * BN + rt] + gsum[3 * BN + rt]);
        }
        if (prof && lane == 0) { atomicAdd(a.prof + 4, t_dma); atomicAdd(a.prof + 5, t_mma); atomicAdd(a.prof + 6, t_epi); atomicAdd(a.prof + 7, t_wait); atomicAdd(a.prof + 8, 1ull); }
    }
}

