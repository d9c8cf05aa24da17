// Arguments and host-side launcher of the input-stationary inverted-residual block kernel (k_mbx.h).  The kernel itself is
// compiled in its own translation unit (mbx_launch.cpp; the two units build in parallel), so engine.cpp sees only this.
#pragma once
#include <hip/hip_runtime.h>

struct MbxArgs {
    const float* in;           // [B][256][inLd]
    float* out;                // [B][256][outLd]           MODE 0 / 2: block output; MODE 3: the activated depthwise map (outLd >= CEXP)
    const float* res;          // residual [B][256][resLd] or nullptr
    float* gap_out;            // [B][CEXP] channel means   MODE 1 / 3
    const float* gate;         // [B][CEXP] SE gate         MODE 2
    const unsigned char* w1;   // [32 T][KS][hi 32 | lo 32] f16, rows beyond CEXP zero
    const float* ctile;        // [T][K K + 2][32]: depthwise taps, expand bias, depthwise bias
    const unsigned char* w2;   // [COUT][T][hi 32 | lo 32] f16
    const float* b2;           // [COUT]
    int B, inC, inLd, outLd, resLd, T, CEXP, act;
    int nsplit;                // MODE 1 / 3: a face's T channel tiles are independent there, so a face is `nsplit` work units (tile ranges) -- 384
                               // faces on 256 CUs are two rounds of whole faces but three of half faces (1.5 instead of 2 face times); 1 otherwise
    float scale1, scale2;      // 1 / (power-of-two weight scales)
    unsigned* range_slot;
    unsigned long long* prof;  // ablation build, dbg & 64: per-wave cycle totals {prologue + expand(0), project, wait a, depthwise, expand, wait b, epilogue, waves}
    int dbg;                   // timing ablations (ablation build only; results are WRONG when set): 1 no DMA after the first tile, 2 no depthwise
                               // taps, 4 no MFMAs, 16 no output stores
};


// Launches mbx_kernel<nw, KS, Cout / 16, K, dil, mode> on `stream` with `grid` persistent workgroups of nw waves (8 or 16).
// Returns 0, -1 when no instantiation matches (engine.cpp reports the shape), or the hipError_t of a failed launch.
int pf_mbx_launch(const MbxArgs& a, int nw, int KS, int Cout, int K, int dil, int mode, int grid, hipStream_t stream);
