// HRNet Bottleneck (timm hrnet.py Bottleneck; TeacherNet encoder layer1, model.py:306-311; oracle/teacher_net.py::_bottleneck) in
// ONE launch (round 4):   out = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + shortcut(x))
// with mid = 64 channels, 256 out; shortcut = x (256 channels) or a 1x1 conv + bn of x (the first block: 64 channels in).
// Layer by layer the four blocks of layer1 were 13 launches and 5.5 of the Teacher's 23.9 ms per 256 crops, bandwidth-bound: the
// 256-channel 64 x 64 map (4 MB per face) crossed HBM three times per block.  Here it is read once (its halo rows twice) and
// written once per block.  A workgroup (16 waves) owns a TR x TW tile of one face's map (8 x 16 at 64 x 64: 1.41 x halo):
//   conv1   K loop over 32-channel chunks of x: the chunks of the (TR + 2) x W pixels for steps c + 1 and c + 2 are in flight
//           (registers) while the MFMAs of step c run on the planes in LDS (two buffers); accumulators (<= 5 tiles per wave) in
//           registers; relu, zero outside the image (conv2 pads ITS input) -> mid1 planes.  The first block's shortcut conv
//           (64 input channels = both buffers resident) runs right behind the loop, its accumulators wait in registers
//   conv2   3x3 as nine shifted fragment reads of the mid1 planes; the tap's weights come from a ring of four 16 KB slots in the
//           x buffers (free by now), filled four taps ahead with one 16-byte piece per thread; result parked over mid1
//   conv3   wave = one 16-channel tile of the 256 outputs, walks the tile's pixel sub-tiles; the identity shortcut is re-read
//           from L2 (the K loop has just streamed it), four sub-tiles per round trip
// Barriers retire LDS traffic only (pf_wait_vm_barrier<63>): __syncthreads() would drain the look-ahead loads at every step.
// Round-4 history (per workgroup, cycles, profiles/r04_run13/17_*): conv1 47 k -> 34 k (now at ~4 TB/s of x, halo rows included),
// conv2 44 k -> 20 k, conv3 14 k; 1.50 -> 1.0 ms per block of 256 faces.
// Split precision (3 x v_mfma_f32_16x16x32_f16 per 32 k), range guard like every other splitting kernel.
#pragma once
#include "k_det.h"

struct HrbArgs {
    const float* x;       // [B][H][W][xLd], CIN channels
    float* out;           // [B][H][W][outLd], 256 channels
    const pf_half* w1; const float* b1;     // [64][CIN/32][64]
    const pf_half* w2; const float* b2;     // [64][9][2][64]
    const pf_half* w3; const float* b3;     // [256][2][64]
    const pf_half* wd; const float* bd;     // downsample [256][2][64] (CIN == 64 blocks only) or nullptr
    float s1, s2, s3, sd;
    int B, H, W, xLd, outLd, TR, TW, tiles_x, tpf;      // tile = TR rows x TW columns; tpf = tiles per face
    unsigned* range_slot;
    unsigned long long* prof;   // ablation build only (PEPPA_DBG & 4096): cycles of conv1 / conv2 / conv3 [3], workgroups
};

template <int CIN, bool DS, int MAXR, int MAXP, int ITX>
__global__ __launch_bounds__(1024) void hr_bottleneck_kernel(HrbArgs a) {
    constexpr int NTHR = 1024, NW = 16, KSA = CIN / 32;
    constexpr int PL = 2 * MAXR * 64;                               // one 32-channel chunk: hi + lo planes
    static_assert(!DS || CIN == 64, "the shortcut conv reads both resident chunks of a 64-channel input");
    __shared__ __attribute__((aligned(16))) unsigned char s_x[2 * PL];      // x chunk planes, two buffers
    __shared__ __attribute__((aligned(16))) unsigned char s_m[2 * PL];      // mid1 planes (region rows); later mid2 (tile rows)
    __shared__ unsigned char s_in[MAXR];
    PF_EMU_POISON(s_x); PF_EMU_POISON(s_m); PF_EMU_POISON(s_in);

    unsigned amax = 0;
    const unsigned amax_seen = pf_amax_seen(a.range_slot);
    const bool prof = PF_ABLATE != 0 && a.prof != nullptr;
    const unsigned long long t0 = prof ? pf_clock() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (workgroup i runs on XCD i % 8): XCD x walks a contiguous run of (face, row-tile) pairs, so the two halo
    // rows a tile shares with its neighbours are in THAT XCD's L2 when the neighbour asks for them -- in dispatch order every tile's
    // neighbours sit on other XCDs and the 2 x (TR + 2) / TR read amplification of the input goes all the way to HBM
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    const int b = tile / a.tpf, tt = tile - b * a.tpf;
    const int oy0 = (tt / a.tiles_x) * a.TR, ox0 = (tt % a.tiles_x) * a.TW;
    const int RW = a.TW + 2, R = (a.TR + 2) * RW, MR = (R + 15) & ~15;
    const int P = a.TR * a.TW, MRD = (P + 15) & ~15;
    const int g = lane >> 4, g4 = g * 4;
    const float* x = a.x + (size_t)b * a.H * a.W * a.xLd;

    // ---- conv1: K loop over the chunks of x ------------------------------------------------------------------------------------------
    // One (region pixel, 8-channel unit) item per thread and chunk (host: (TR + 2) x (TW + 2) x 4 <= ITX x 1024); the rows beyond the
    // region are zeroed once in both buffers.  Chunks c + 1 AND c + 2 are in flight while the MFMAs of chunk c run: with one step of
    // look-ahead the ~1 k cycles of a step's MFMAs did not cover a loaded-memory round trip (a step took 5.9 k cycles), and
    // __syncthreads() would drain vmcnt at every step -- the barriers below only retire the LDS writes.
    int xo[ITX], xr[ITX];                                           // element offset (-1: pixel outside the image), region row (-1: no item)
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
        const int i = tid + it * NTHR;
        const int r = i >> 2;
        const int ry = r / RW, rx = r - ry * RW;
        const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
        xr[it] = r < R ? r : -1;
        xo[it] = (r < R && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? (iy * a.W + ix) * a.xLd + (i & 3) * 8 : -1;
    }
    auto load_x = [&](int c, pf_f32x4 (&s)[ITX][2]) {
#pragma unroll
        for (int it = 0; it < ITX; ++it) {
            s[it][0] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            s[it][1] = s[it][0];
            if (xo[it] >= 0) {
                s[it][0] = *reinterpret_cast<const pf_f32x4*>(x + xo[it] + c * 32);
                s[it][1] = *reinterpret_cast<const pf_f32x4*>(x + xo[it] + c * 32 + 4);
            }
        }
    };
    auto park_x = [&](int buf, const pf_f32x4 (&s)[ITX][2]) {
#pragma unroll
        for (int it = 0; it < ITX; ++it)
            if (xr[it] >= 0) det_park8(s_x + buf * PL, MR, xr[it], (tid + it * NTHR) & 3, s[it][0], s[it][1], amax);
    };
    const int ntA = wave & 3, mgA = wave >> 2;                      // conv1 / conv2: 4 channel tiles x 4 groups of pixel tiles
    constexpr int MAXTA = (MAXR / 16 + 3) / 4;
    pf_f32x4 accA[MAXTA];
#pragma unroll
    for (int j = 0; j < MAXTA; ++j) accA[j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int DEPTH = 3;                                        // register sets: chunk c (being parked / read), c + 1, c + 2
    pf_f32x4 st[DEPTH][ITX][2];
    pf_half8 wq[DEPTH][2];                                          // conv1 weight fragments (hi, lo) of the same chunks
    const pf_half* w1p = a.w1 + ((size_t)(ntA * 16 + (lane & 15)) * KSA) * 64 + g * 8;
#pragma unroll
    for (int c = 0; c < 2 && c < KSA; ++c) {
        load_x(c, st[c]);
        wq[c][0] = *reinterpret_cast<const pf_half8*>(w1p + c * 64);
        wq[c][1] = *reinterpret_cast<const pf_half8*>(w1p + c * 64 + 32);
    }
    for (int i = tid; i < MR * 4; i += NTHR) {                      // rows beyond the region in both buffers, and the in-image flags
        const int r = i >> 2;
        const int ry = r / RW, rx = r - ry * RW;
        if (r >= R) {
            const pf_f32x4 z = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            det_park8(s_x, MR, r, i & 3, z, z, amax);
            det_park8(s_x + PL, MR, r, i & 3, z, z, amax);
        }
        if ((i & 3) == 0) s_in[r] = (r < R && (unsigned)(oy0 - 1 + ry) < (unsigned)a.H && (unsigned)(ox0 - 1 + rx) < (unsigned)a.W) ? 1 : 0;
    }
    park_x(0, st[0]);
    pf_wait_vm_barrier<63>();
#pragma unroll
    for (int c = 0; c < KSA; ++c) {
        if (c + 2 < KSA) {
            load_x(c + 2, st[(c + 2) % DEPTH]);
            wq[(c + 2) % DEPTH][0] = *reinterpret_cast<const pf_half8*>(w1p + (c + 2) * 64);
            wq[(c + 2) % DEPTH][1] = *reinterpret_cast<const pf_half8*>(w1p + (c + 2) * 64 + 32);
        }
        const unsigned char* xb = s_x + (c & 1) * PL;
#pragma unroll
        for (int j = 0; j < MAXTA; ++j) {
            const int mt = mgA + 4 * j;
            if (mt < MR / 16) {
                pf_half8 xh, xl;
                det_frag(xb, MR, 0, mt * 16 + (lane & 15), g, xh, xl);
                accA[j] = pf_mfma_16x16x32_f16(wq[c % DEPTH][1], xh, accA[j]);
                accA[j] = pf_mfma_16x16x32_f16(wq[c % DEPTH][0], xl, accA[j]);
                accA[j] = pf_mfma_16x16x32_f16(wq[c % DEPTH][0], xh, accA[j]);
            }
        }
        if (c + 1 < KSA) {
            park_x((c + 1) & 1, st[(c + 1) % DEPTH]);               // the other buffer: nobody reads it during this step
            pf_wait_vm_barrier<63>();
        }
    }
    // The shortcut conv of the first block (64 input channels = both x buffers resident): run NOW, wave = one 16-channel tile of
    // the 256 outputs x every pixel tile, accumulators held in registers until conv3's epilogue -- so that the x buffers are free for
    // conv2's weight slots in both variants.
    const int ntC = wave;
    constexpr int MAXTC = MAXP / 16;
    pf_f32x4 accd[DS ? MAXTC : 1];
    if constexpr (DS) {
        pf_half8 wdh[2], wdl[2];
        det_wfrag<2>(a.wd, ntC, lane, wdh, wdl);
#pragma unroll
        for (int mt = 0; mt < MAXTC; ++mt) {
            accd[mt] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            const int p = mt * 16 + (lane & 15);
            const int pc = p < P ? p : 0;
            const int py = pc / a.TW, px = pc - py * a.TW;
            const int row = (py + 1) * RW + px + 1;                 // the pixel's row in the x planes
            if (mt < MRD / 16) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    pf_half8 xh, xl;
                    det_frag(s_x + ks * PL, MR, 0, row, g, xh, xl);
                    accd[mt] = pf_mfma_16x16x32_f16(wdl[ks], xh, accd[mt]);
                    accd[mt] = pf_mfma_16x16x32_f16(wdh[ks], xl, accd[mt]);
                    accd[mt] = pf_mfma_16x16x32_f16(wdh[ks], xh, accd[mt]);
                }
            }
        }
    }
    // conv2's weights go through LDS (the x buffers are free once conv1 and the shortcut conv are done): four 16 KB tap slots
    // [ks][hi | lo][64 rows][64 B] in s_x, every thread moves one 16-byte piece per tap, requested four taps ahead.  Per-wave fragment
    // loads from L2 with one tap of look-ahead cost a loaded L2 round trip per tap (44 k cycles for 7 k cycles of MFMAs; 19.7 k now).
    static_assert(2 * PL >= 4 * 16384, "four tap slots in the x buffers");
    const pf_half* w2src = a.w2 + ((size_t)(tid >> 4) * 9 * 2 + ((tid >> 3) & 1)) * 64 + ((tid >> 2) & 1) * 32 + (tid & 3) * 8;   // + tap * 128
    const int w2dst = (((tid >> 3) & 1) * 2 + ((tid >> 2) & 1)) * 4096 + pf_lds_chunk_off(tid >> 4, tid & 3);                       // + slot * 16384
    pf_half8 w2p[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w2p[t] = *reinterpret_cast<const pf_half8*>(w2src + t * 128);
    {
        const pf_f32x4 b1v = *reinterpret_cast<const pf_f32x4*>(a.b1 + ntA * 16 + g4);
#pragma unroll
        for (int j = 0; j < MAXTA; ++j) {
            const int mt = mgA + 4 * j;
            if (mt < MR / 16) {
                const int r = mt * 16 + (lane & 15);
                pf_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = fmaf(accA[j][e], a.s1, b1v[e]); v[e] = v[e] > 0.f ? v[e] : 0.f; }
                if (!s_in[r]) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                det_park4(s_m, MR, r, ntA * 4 + g, v, amax);
            }
        }
    }
    pf_wait_vm_barrier<63>();                                       // mid1 is complete; every wave is done reading the x buffers
    const unsigned long long t1 = prof ? pf_clock() : 0;

    // ---- conv2: 3x3 on the mid1 planes -----------------------------------------------------------------------------------------------
    constexpr int MAXTB = (MAXP / 16 + 3) / 4;                      // tile pixels: TR * W <= MAXP
    pf_f32x4 accB[MAXTB];
    int rowB[MAXTB];
#pragma unroll
    for (int j = 0; j < MAXTB; ++j) {
        accB[j] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        const int p = (mgA + 4 * j) * 16 + (lane & 15);
        const int pc = p < P ? p : 0;
        const int py = pc / a.TW, px = pc - py * a.TW;
        rowB[j] = py * RW + px;                                     // region row of tap (0, 0)
    }
    {
        pf_half8 w2q[2];                                            // pieces of taps t + 3 and t + 4, by tap parity
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<pf_half8*>(s_x + t * 16384 + w2dst) = w2p[t];
        w2q[1] = w2p[3];
        pf_wait_vm_barrier<63>();
        const int wrow = pf_lds_chunk_off(ntA * 16 + (lane & 15), g);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 4 < 9) w2q[tap & 1] = *reinterpret_cast<const pf_half8*>(w2src + (tap + 4) * 128);
            const unsigned char* wb = s_x + (tap & 3) * 16384 + wrow;
            pf_half8 wch[2], wcl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                wch[ks] = *reinterpret_cast<const pf_half8*>(wb + (ks * 2) * 4096);
                wcl[ks] = *reinterpret_cast<const pf_half8*>(wb + (ks * 2 + 1) * 4096);
            }
            const int shift = (tap / 3) * RW + tap % 3;
#pragma unroll
            for (int j = 0; j < MAXTB; ++j) {
                if (mgA + 4 * j < MRD / 16) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        pf_half8 zh, zl;
                        det_frag(s_m, MR, ks, rowB[j] + shift, g, zh, zl);
                        accB[j] = pf_mfma_16x16x32_f16(wcl[ks], zh, accB[j]);
                        accB[j] = pf_mfma_16x16x32_f16(wch[ks], zl, accB[j]);
                        accB[j] = pf_mfma_16x16x32_f16(wch[ks], zh, accB[j]);
                    }
                }
            }
            if (tap + 3 < 9) {                                      // slot (tap + 3) & 3 was last read during tap - 1: behind the previous barrier
                *reinterpret_cast<pf_half8*>(s_x + ((tap + 3) & 3) * 16384 + w2dst) = w2q[(tap + 3) & 1];
                pf_wait_vm_barrier<63>();
            }
        }
    }
    // weights of conv3: in flight across the two barriers
    pf_half8 w3h[2], w3l[2];
    det_wfrag<2>(a.w3, ntC, lane, w3h, w3l);
    pf_f32x4 b3v = *reinterpret_cast<const pf_f32x4*>(a.b3 + ntC * 16 + g4);
    if constexpr (DS) {
        const pf_f32x4 bdv = *reinterpret_cast<const pf_f32x4*>(a.bd + ntC * 16 + g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) b3v[e] += bdv[e];
    }
    __syncthreads();                                                // every wave is done reading mid1
    {
        const pf_f32x4 b2v = *reinterpret_cast<const pf_f32x4*>(a.b2 + ntA * 16 + g4);
#pragma unroll
        for (int j = 0; j < MAXTB; ++j) {
            const int mt = mgA + 4 * j;
            if (mt < MRD / 16) {
                const int p = mt * 16 + (lane & 15);
                pf_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = fmaf(accB[j][e], a.s2, b2v[e]); v[e] = v[e] > 0.f ? v[e] : 0.f; }
                if (p >= P) v = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                det_park4(s_m, MRD, p, ntA * 4 + g, v, amax);
            }
        }
    }
    __syncthreads();
    const unsigned long long t2 = prof ? pf_clock() : 0;

    // ---- conv3 (+ shortcut) + relu -> global -------------------------------------------------------------------------------------------
    float* out = a.out + (size_t)b * a.H * a.W * a.outLd;
    const int n = ntC * 16 + g4;
#pragma unroll
    for (int mt0 = 0; mt0 < MAXTC; mt0 += 4) {
        if (mt0 >= MRD / 16) break;
        // the identity shortcut of four pixel sub-tiles at a time: one exposed L2 round trip per four tiles, not per tile (the first
        // cut loaded each tile's residual right in front of its use: ~3 k cycles x 8 per workgroup)
        pf_f32x4 res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            res[q] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!DS) {
                const int p = (mt0 + q) * 16 + (lane & 15);
                const int py = p / a.TW, px = p - py * a.TW;
                if (p < P && oy0 + py < a.H && ox0 + px < a.W) res[q] = *reinterpret_cast<const pf_f32x4*>(x + ((size_t)(oy0 + py) * a.W + ox0 + px) * a.xLd + n);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mt = mt0 + q;
            if (mt >= MRD / 16) break;
            const int p = mt * 16 + (lane & 15);
            const int pc = p < P ? p : 0;
            const int py = pc / a.TW, px = pc - py * a.TW;
            const int oy = oy0 + py, ox = ox0 + px;
            const bool ok = p < P && oy < a.H && ox < a.W;
            const pf_f32x4 acc3 = det_tile<2>(s_m, MRD, mt * 16, lane, w3h, w3l);
            pf_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc3[e], a.s3, b3v[e]) + res[q][e];
            if constexpr (DS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(accd[mt0 + q][e], a.sd, v[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            if (ok) *reinterpret_cast<pf_f32x4*>(out + ((size_t)oy * a.W + ox) * a.outLd + n) = v;
        }
    }
    pf_amax_commit(a.range_slot, amax, amax_seen);
    if (prof && tid == 0) {
        const unsigned long long t3 = pf_clock();
        atomicAdd(a.prof + 0, t1 - t0); atomicAdd(a.prof + 1, t2 - t1); atomicAdd(a.prof + 2, t3 - t2); atomicAdd(a.prof + 3, 1ull);
    }
}
