// Device half of the JPEG decoder behind pf_decode_jpeg (frame ingest, SURVEY 8 next-row N2: replaces cv2.imread at
// demo.py:76 for .jpg files).  The entropy-coded segment is decoded on the host (jpeg.inl: it is a serial bit stream);
// everything after it is per-block / per-pixel work and runs here, bit-identical with what cv2.imread / libjpeg(-turbo)
// produce with their defaults (JDCT_ISLOW, fancy upsampling, YCbCr -> RGB by the 16-bit fixed-point tables):
//   jpeg_idct_kernel   dequantisation + the 8x8 "islow" integer inverse DCT of the IJG code (jidctint.c: CONST_BITS 13,
//                      PASS1_BITS 2, column pass then row pass, +128, range limit), one thread per block;
//   jpeg_color_kernel  chroma upsampling -- none (4:4:4), h2v1 / h2v2 "fancy" triangle filters (jdsample.c, edge columns and
//                      rows replicated exactly as the IJG main controller feeds them) -- and YCbCr -> BGR (jdcolor.c), or
//                      grey -> BGR, written as packed 8-bit BGR rows like cv2.imread returns.
#pragma once
#include "pf_common.h"

// per-handle buffers of the decoder (jpeg.inl)
struct JpegState {
    // two alternating slots: a batch decode of step k + 1 (host Huffman loop, uploads on the copy stream) overlaps the pipeline
    // that is still reading the frames of step k on the engine's stream
    struct Slot {
        unsigned char* h_pack = nullptr;  // page-locked: per frame [block table u32 x blocks][records: length, coefficients in zigzag
        unsigned char* d_pack = nullptr;  //   order up to the last non-zero one] -- what the entropy decoder writes and PCIe carries
        size_t pack_cap = 0;
        unsigned char* d_bgr = nullptr;   // the decoded frames, packed BGR
        size_t bgr_cap = 0;
        hipEvent_t uploaded = nullptr;    // this slot's records have arrived (copy stream)
        hipEvent_t consumed = nullptr;    // ... and have been expanded (engine stream): the slot may be refilled
        bool in_flight = false;
    } slot[2];
    int cur = 0;
    hipStream_t copy_stream = nullptr;
    short* d_coef = nullptr;          // dense [block][64] natural-order blocks, expanded on the device
    size_t coef_cap = 0;
    unsigned char* d_planes = nullptr;    // component planes after the inverse DCT
    size_t planes_cap = 0;
    unsigned short* d_quant = nullptr;    // [frames][3][64]
    size_t quant_cap = 0;
    struct JpegFrameDesc* d_desc = nullptr;   // [frames]
    size_t desc_cap = 0;
    unsigned* d_sub = nullptr;            // sub-sequence records of the self-synchronising decode (jpeg_sync_kernel)
    size_t sub_cap = 0;                   // in words
    void release();
};

struct JpegFrameDesc {                    // per file of the batch, in the frame's region of the pack buffer (16-byte aligned fields)
    unsigned scan_off, scan_len;          // the entropy-coded segment inside the region
    unsigned offs_off, n_intervals;       // u32 offsets (relative to scan_off) of each interval's first byte
    unsigned tables_off;
    unsigned restart;                     // MCUs per interval
    unsigned tdta;                        // bit c: component c uses DC table 1; bit 4 + c: AC table 1
    unsigned pad;
};

struct JpegUnpackArgs {
    const unsigned char* pack;    // [frame] regions of frame_pack_bytes
    size_t frame_pack_bytes;
    short* coef;                  // [frame][blocks][64]
    int blocks;
    const JpegFrameDesc* desc;   // frames with n_intervals != 0 are decoded by jpeg_huffman_kernel, not expanded here
};

__device__ __forceinline__ int pf_jpeg_zigzag(int k) {      // zigzag position -> natural (row-major) index
    constexpr unsigned char t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return t[k];
}

// packed records -> dense natural-order blocks (zeros where the record ends or no scan reached the block)
__global__ __launch_bounds__(64) void jpeg_unpack_kernel(JpegUnpackArgs a) {
    const int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.blocks || a.desc[blockIdx.y].n_intervals != 0) return;
    const unsigned char* region = a.pack + (size_t)blockIdx.y * a.frame_pack_bytes;
    const unsigned start = reinterpret_cast<const unsigned*>(region)[blk];
    short* dst = a.coef + ((size_t)blockIdx.y * a.blocks + blk) * 64;
    pf_f32x4 z = pf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<pf_f32x4*>(dst)[i] = z;
    if (start == 0xFFFFFFFFu) return;
    const short* rec = reinterpret_cast<const short*>(region + (size_t)a.blocks * 4) + start;
    const int len = rec[0];
    for (int k = 0; k < len && k < 64; ++k) {
        const short v = rec[1 + k];
        if (v) dst[pf_jpeg_zigzag(k)] = v;
    }
}

// ---- entropy decoding on the device, for files that carry restart markers ---------------------------------------------------------
// A restart interval is an independent, byte-aligned piece of the scan (DC predictions reset, ITU T.81 E.1.4): one thread per
// interval walks its bit stream exactly as the host decoder does (9-bit lookahead tables, combined code + magnitude lookup for
// short AC codes, canonical-code search beyond) and writes the coefficients into the dense, pre-zeroed block buffer.  Only the
// compressed scan, the interval offsets and the tables cross PCIe.
struct JpegGpuTables {                    // index 0, 1: DC tables 0 / 1; 2, 3: AC tables 0 / 1
    unsigned short look[4][512];
    short fast_ac[4][512];                // rows 2, 3 used
    int maxcode[4][18];
    int valoff[4][17];
    unsigned char vals[4][256];
    int limit[4][17];                     // left-justified (16-bit) exclusive upper bound of the codes of each length; [0] = 0
};


struct JpegHuffArgs {
    const unsigned char* pack;            // [frame] regions of frame_pack_bytes
    size_t frame_pack_bytes;
    const JpegFrameDesc* desc;            // [frame]
    short* coef;                          // [frame][blocks][64], zeroed
    int blocks;                           // per frame
    int ncomp, mcux, total_mcus;
    int ch[3], cv[3], cblock0[3], cbw[3];
};

struct PfJpegBits {
    const unsigned char* base;            // 8-byte aligned start of the frame's region (the scan in it has no byte stuffing)
    unsigned pos, limit;                  // byte offset from base; the stream ends at limit (a corrupt stream must not walk off)
    unsigned long long chunk, ahead;      // the aligned 8 bytes that contain pos, and the 8 after them, requested one chunk early
    unsigned chunk_at;                    //   (a dependent global load per refill would cost a full L2 round trip each)
    unsigned long long acc;
    int n;
};

// Branch-light refill: four bytes at a time out of the (chunk, ahead) window.  (With the stuffed stream each lane looped over
// single bytes and the wave paid the longest loop on every symbol: ~0.8 us per symbol.)
__device__ __forceinline__ void pf_jpeg_fill(PfJpegBits& br) {
    if (br.n > 32) return;
    if (br.pos >= br.limit) { br.acc <<= 32; br.n += 32; return; }       // past the end: zeros, like the host decoder
    const unsigned at = br.pos & ~7u;
    if (at != br.chunk_at) {
        br.chunk = at == br.chunk_at + 8 ? br.ahead : *reinterpret_cast<const unsigned long long*>(br.base + at);
        br.ahead = *reinterpret_cast<const unsigned long long*>(br.base + at + 8);     // in flight while this chunk is consumed
        br.chunk_at = at;
    }
    const unsigned sh = 8u * (br.pos & 7u);
    const unsigned long long w = sh ? (br.chunk >> sh) | (br.ahead << (64u - sh)) : br.chunk;
    br.acc = (br.acc << 32) | __builtin_bswap32((unsigned)w);             // bytes pos .. pos + 3, stream order
    br.n += 32;
    br.pos += 4;
}
__device__ __forceinline__ int pf_jpeg_peek(const PfJpegBits& br, int k) { return (int)((br.acc >> (br.n - k)) & ((1u << k) - 1)); }

__device__ __forceinline__ int pf_jpeg_huff(PfJpegBits& br, const JpegGpuTables* t, int ti) {
    const int lk = t->look[ti][pf_jpeg_peek(br, 9)];
    if (lk) { br.n -= lk >> 8; return lk & 0xFF; }
    int code = pf_jpeg_peek(br, 9), l = 9;
    br.n -= 9;
    for (;;) {
        code = (code << 1) | pf_jpeg_peek(br, 1);
        br.n -= 1;
        ++l;
        if (l > 16) return 0;
        if (t->maxcode[ti][l] >= 0 && code <= t->maxcode[ti][l]) return t->vals[ti][(code + t->valoff[ti][l]) & 0xFF];
    }
}

// The same symbol without the bit-by-bit search: a canonical code's length is the number of per-length upper bounds its 16-bit
// left-justified window has reached.  In a wave whose 64 lanes sit at unrelated places of their streams SOME lane needs the
// search on almost every step, and the wave paid for the longest loop (up to seven rounds of shift / compare / branch).
__device__ __forceinline__ int pf_jpeg_huff_flat(PfJpegBits& br, const JpegGpuTables* t, int ti) {
    const int lk = t->look[ti][pf_jpeg_peek(br, 9)];
    if (lk) { br.n -= lk >> 8; return lk & 0xFF; }
    const int c = pf_jpeg_peek(br, 16);
    int l = 10;
#pragma unroll
    for (int q = 10; q <= 15; ++q) l += c >= t->limit[ti][q] ? 1 : 0;
    br.n -= l;
    if (c >= t->limit[ti][16]) return 0;                  // no such code (corrupt stream)
    return t->vals[ti][((c >> (16 - l)) + t->valoff[ti][l]) & 0xFF];
}

__device__ __forceinline__ int pf_jpeg_extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(JpegHuffArgs a) {
    const unsigned char* region = a.pack + (size_t)blockIdx.y * a.frame_pack_bytes;
    const JpegFrameDesc d = a.desc[blockIdx.y];
    if (d.restart == 0) return;                           // no restart markers: jpeg_sync_kernel's frame
    const int iv = blockIdx.x * 64 + threadIdx.x;
    const unsigned* offs = reinterpret_cast<const unsigned*>(region + d.offs_off);
    // the file's lookup tables and the zigzag map into LDS: every lookup is on the thread's critical path, from global / constant
    // memory each would be an L2 round trip per symbol (8.5 -> 6.8 ms for the 120-MCU intervals of a 1080p file)
    __shared__ JpegGpuTables tables;
    __shared__ unsigned char zz[64];
    // the block being decoded lives in LDS and leaves as eight 16-byte stores when it is complete: on this ISA stores share the
    // loads' counter, so every wait for the next stream chunk is also a wait for ALL outstanding stores -- one 2-byte global store
    // per coefficient made that a full store round trip every few symbols
    constexpr int BLK_STRIDE = 72;                        // shorts per thread: 144 bytes, 16-byte aligned, rows on staggered banks
    __shared__ __attribute__((aligned(16))) short blkbuf[64 * BLK_STRIDE];
    short* mine = blkbuf + threadIdx.x * BLK_STRIDE;
    zz[threadIdx.x] = (unsigned char)pf_jpeg_zigzag(threadIdx.x);
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(region + d.tables_off);
        unsigned* dst = reinterpret_cast<unsigned*>(&tables);
        for (unsigned i = threadIdx.x; i < sizeof(JpegGpuTables) / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (iv >= (int)d.n_intervals) return;
    const JpegGpuTables* t = &tables;
    PfJpegBits br;
    br.base = region;
    br.pos = d.scan_off + offs[iv];
    br.limit = d.scan_off + d.scan_len + 8;              // (staged with 32 zero bytes behind the scan)
    br.chunk = 0; br.ahead = 0; br.chunk_at = 0xFFFFFFF0u;
    br.acc = 0; br.n = 0;
    short* coef = a.coef + (size_t)blockIdx.y * a.blocks * 64;
    int pred[3] = {0, 0, 0};
    const int m0 = iv * (int)d.restart;
    const int m1 = min(m0 + (int)d.restart, a.total_mcus);
    for (int m = m0; m < m1; ++m) {
        const int my = m / a.mcux, mx = m - my * a.mcux;
        for (int c = 0; c < a.ncomp; ++c) {
            const int nb = a.ch[c] * a.cv[c];
            for (int bi = 0; bi < nb; ++bi) {
                const int v = bi / a.ch[c], hh = bi - v * a.ch[c];
                short* blk = coef + ((size_t)a.cblock0[c] + (size_t)(my * a.cv[c] + v) * a.cbw[c] + (mx * a.ch[c] + hh)) * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) reinterpret_cast<pf_f32x4*>(mine)[i] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                pf_jpeg_fill(br);
                const int s = pf_jpeg_huff(br, t, (d.tdta >> c) & 1);
                const int sb = s > 15 ? 15 : s;
                pf_jpeg_fill(br);
                const int diff = sb ? pf_jpeg_extend(pf_jpeg_peek(br, sb), sb) : 0;
                br.n -= sb;
                pred[c] += diff;
                mine[0] = (short)pred[c];
                const int ta = 2 + ((d.tdta >> (4 + c)) & 1);
                for (int k = 1; k < 64;) {
                    pf_jpeg_fill(br);
                    const int fa = t->fast_ac[ta][pf_jpeg_peek(br, 9)];
                    int r, val;
                    if (fa) {
                        br.n -= fa & 15;
                        r = (fa >> 4) & 15;
                        val = fa >> 8;
                    } else {
                        const int rs = pf_jpeg_huff(br, t, ta);
                        const int sz = rs & 15;
                        r = rs >> 4;
                        if (sz == 0) {
                            if (r != 15) break;
                            k += 16;
                            continue;
                        }
                        val = pf_jpeg_extend(pf_jpeg_peek(br, sz), sz);
                        br.n -= sz;
                    }
                    k += r;
                    if (k > 63) break;
                    mine[zz[k]] = (short)val;
                    ++k;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) reinterpret_cast<pf_f32x4*>(blk)[i] = reinterpret_cast<const pf_f32x4*>(mine)[i];
            }
        }
    }
}

// ---- entropy decoding on the device, for files WITHOUT restart markers: self-synchronising sub-sequence decode --------------
// An ordinary camera file is one unbroken Huffman stream, but Huffman codes re-synchronise: a decoder started at an arbitrary bit
// soon falls into step with the true symbol boundaries.  The (byte-unstuffed) scan is cut into sub-sequences of PF_JPEG_SUBSEQ_BITS
// bits, one thread each:
//   1. jpeg_sync_kernel, first pass: thread i decodes sub-sequence i from the GUESS "a block starts at my first bit" and records
//      where it leaves: (bit position of the first symbol past its end, block of the MCU, coefficient index) and how many blocks
//      it completed.  Thread 0's guess is the truth.
//   2. jpeg_sync_kernel, rounds: thread i takes the recorded exit of sub-sequence i - 1 as its entry, decodes again and replaces
//      its record if it differs (counting the change).  A thread works only if its predecessor's record changed in the round
//      before, so after the first round almost nobody does.  When a round changes nothing every record is consistent with its
//      predecessor, hence -- by induction from thread 0 -- correct.  The host queues a fixed number of rounds and the verdict
//      (changes in the LAST round) travels with the engine's status word: no silent garbage if a stream needed more.
//   3. jpeg_subseq_scan_kernel: exclusive prefix sum of the completed-block counts = each sub-sequence's first block ordinal.
//   4. jpeg_sync_kernel, write pass: decode once more from the now-known entry and write the coefficients (DC as the raw
//      difference) into the dense, pre-zeroed block buffer; 5. jpeg_dc_prefix_kernel turns the differences into DC values (a
//      prefix sum per component in scan order -- without restart markers the prediction runs through the whole scan).
// Same tables, bit reader and symbol decoding as jpeg_huffman_kernel; output bit-identical with the host decoder (ITU T.81 F.2).
#define PF_JPEG_SUBSEQ_BITS 1024
#define PF_JPEG_SYNC_ROUNDS 10
#define PF_JPEG_WALK 64              // sub-sequences a thread whose record changed walks on through in one round

struct JpegSubseqArgs {
    const unsigned char* pack;            // [frame] regions of frame_pack_bytes
    size_t frame_pack_bytes;
    const JpegFrameDesc* desc;            // [frame]; restart == 0 marks a frame decoded this way, n_intervals = its sub-sequences
    short* coef;                          // [frame][blocks][64], zeroed
    int blocks;                           // per frame
    int ncomp, mcux, total_mcus, bpm;     // bpm = blocks per MCU
    int ch[3], cv[3], cblock0[3], cbw[3];
    unsigned* exit_p;                     // [frame][max_sub] bit position of the first symbol past the sub-sequence
    unsigned* exit_s;                     // [frame][max_sub] (block of the MCU << 8) | coefficient index
    unsigned* nblk;                       // [frame][max_sub] blocks completed inside the sub-sequence
    unsigned* blk0;                       // [frame][max_sub] first block ordinal (exclusive prefix sum of nblk)
    unsigned* stamp;                      // [frame][max_sub] round in which the record last changed
    unsigned* changed;                    // [PF_JPEG_SYNC_ROUNDS + 1] records changed per round (all frames)
    int max_sub;
    int round;                            // 0 = first pass, 1.. = synchronisation rounds, -1 = write pass
};

template <int MODE>                       // 0 first pass, 1 synchronisation round, 2 write pass
__global__ __launch_bounds__(64) void jpeg_sync_kernel(JpegSubseqArgs a) {
    const unsigned char* region = a.pack + (size_t)blockIdx.y * a.frame_pack_bytes;
    const JpegFrameDesc d = a.desc[blockIdx.y];
    const int i = blockIdx.x * 64 + threadIdx.x;
    __shared__ JpegGpuTables tables;
    __shared__ unsigned char zz[64];
    constexpr int BLK_STRIDE = 72;                        // shorts per thread, as in jpeg_huffman_kernel
    __shared__ __attribute__((aligned(16))) short blkbuf[MODE == 2 ? 64 * BLK_STRIDE : 8];
    if (d.restart != 0 || blockIdx.x * 64 >= (int)d.n_intervals) return;      // (whole workgroup: not a frame / tile of this path)
    if (MODE == 1) {                                      // nobody's predecessor changed in the round before: nothing to load tables for
        const bool dirty = i > 0 && i < (int)d.n_intervals && a.stamp[(size_t)blockIdx.y * a.max_sub + i - 1] == (unsigned)(a.round - 1);
        __shared__ int any_dirty;
        if (threadIdx.x == 0) any_dirty = 0;
        __syncthreads();
        if (dirty) any_dirty = 1;
        __syncthreads();
        if (!any_dirty) return;
    }
    zz[threadIdx.x] = (unsigned char)pf_jpeg_zigzag(threadIdx.x);
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(region + d.tables_off);
        unsigned* dst = reinterpret_cast<unsigned*>(&tables);
        for (unsigned k = threadIdx.x; k < sizeof(JpegGpuTables) / 4; k += 64) dst[k] = src[k];
    }
    __syncthreads();
    const bool idle = i >= (int)d.n_intervals;            // (write pass: the lane still helps to move its neighbours' blocks)
    if (idle && MODE != 2) return;
    const size_t rec = (size_t)blockIdx.y * a.max_sub + (idle ? 0 : i);
    if (MODE == 1 && (i == 0 || a.stamp[rec - 1] != (unsigned)(a.round - 1))) return;   // predecessor unchanged: so is this record
    const JpegGpuTables* t = &tables;
    const unsigned total_bits = d.scan_len * 8u;
    const unsigned end_bits = min((unsigned)(i + 1) * PF_JPEG_SUBSEQ_BITS, total_bits);      // (write pass)
    unsigned p0 = idle ? 0u : (unsigned)i * PF_JPEG_SUBSEQ_BITS;
    int bi = 0, z = 0;
    if (MODE != 0 && i > 0 && !idle) {
        p0 = a.exit_p[rec - 1];
        const unsigned st = a.exit_s[rec - 1];
        bi = (int)(st >> 8);
        z = (int)(st & 255u);
    }
    // block of the MCU -> component and position inside the MCU
    int comp_of[10], v_of[10], h_of[10];
    {
        int j = 0;
        for (int c = 0; c < a.ncomp; ++c)
            for (int q = 0; q < a.ch[c] * a.cv[c] && j < 10; ++q, ++j) { comp_of[j] = c; v_of[j] = q / a.ch[c]; h_of[j] = q - v_of[j] * a.ch[c]; }
    }
    PfJpegBits br;
    br.base = region;
    br.pos = d.scan_off + (p0 >> 3);
    br.limit = d.scan_off + d.scan_len + 8;              // (staged with 32 zero bytes behind the scan)
    br.chunk = 0; br.ahead = 0; br.chunk_at = 0xFFFFFFF0u;
    br.acc = 0; br.n = 0;
    pf_jpeg_fill(br);
    br.n -= (int)(p0 & 7u);
    // bits consumed so far = 8 * (bytes taken into the accumulator) - bits still in it (the reader never runs past `limit` here:
    // decoding stops at total_bits, at most 8 + 32 bits before it)
    auto bitpos = [&]() { return (br.pos - d.scan_off) * 8u - (unsigned)br.n; };
    unsigned done_blocks = 0;
    unsigned ordinal = MODE == 2 ? a.blk0[rec] : 0u;
    short* coef = a.coef + (size_t)blockIdx.y * a.blocks * 64;
    short* mine = blkbuf + (MODE == 2 ? threadIdx.x * BLK_STRIDE : 0);
    bool own = z == 0;                                    // this thread decodes the block from its DC on: it may store it whole
    short* blk = nullptr;
    auto block_ptr = [&](unsigned q) -> short* {
        const int m = (int)(q / (unsigned)a.bpm), j = (int)(q - (unsigned)m * a.bpm);
        const int my = m / a.mcux, mx = m - my * a.mcux, c = comp_of[j];
        return coef + ((size_t)a.cblock0[c] + (size_t)(my * a.cv[c] + v_of[j]) * a.cbw[c] + (mx * a.ch[c] + h_of[j])) * 64;
    };
    if (MODE == 2) {
        if (ordinal < (unsigned)a.total_mcus * a.bpm) blk = block_ptr(ordinal);
#pragma unroll
        for (int k = 0; k < 8; ++k) reinterpret_cast<pf_f32x4*>(mine)[k] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned total_blocks = (unsigned)a.total_mcus * a.bpm;
    auto symbol_step = [&]() {
        // ONE symbol per iteration, the same instruction sequence for the DC difference and for an AC (run, size) symbol: the 64
        // lanes of a wave are at unrelated places of their blocks, and separate DC / AC / fast-path branches made the wave pay
        // for every path on every step
        const int c = comp_of[bi];
        pf_jpeg_fill(br);                                 // > 32 bits: a code (<= 16) and its magnitude bits (<= 15)
        const bool dc = z == 0;
        const int sym = pf_jpeg_huff_flat(br, t, dc ? (int)((d.tdta >> c) & 1u) : 2 + (int)((d.tdta >> (4 + c)) & 1u));
        const int sz = dc ? (sym > 15 ? 15 : sym) : (sym & 15);
        const int r = dc ? 0 : (sym >> 4);
        const int raw = pf_jpeg_peek(br, sz ? sz : 1);
        const int val = sz ? pf_jpeg_extend(raw, sz) : 0;
        br.n -= sz;
        if (dc) {
            if (MODE == 2) mine[0] = (short)val;          // (a thread that meets a DC owns the block)
            z = 1;
        } else if (sz == 0) {
            z = r == 15 ? z + 16 : 64;                    // sixteen zeros / end of block
        } else {
            z += r;
            if (z <= 63) {
                if (MODE == 2) {
                    if (own) mine[zz[z]] = (short)val;
                    else if (blk) blk[zz[z]] = (short)val;           // a block another thread began: only what is decoded here
                }
                ++z;
            } else {
                z = 64;                                   // a run past the block: the host decoder ends the block here too
            }
        }
    };
    auto decode_until = [&](unsigned stop_bits) {         // (guess pass and synchronisation rounds: nothing is written)
        while (bitpos() < stop_bits) {
            symbol_step();
            if (z >= 64) {                                // block complete
                ++done_blocks;
                z = 0;
                bi = bi + 1 == a.bpm ? 0 : bi + 1;
            }
        }
    };
    if (MODE == 2) {
        // Write pass.  A completed block the lane owns (it decoded it from the DC on) is moved from LDS to the dense buffer by the
        // WHOLE wave -- lane k stores coefficient k: one 2-byte store instruction per block -- because completions happen at
        // unrelated moments in the 64 lanes: as a per-lane branch (eight 16-byte stores, eight LDS reads, eight LDS writes to
        // clear, the block address arithmetic) that path ran ~320 times per wave and was more than half of this pass's
        // instructions (SQ_INSTS_VALU 47 k per wave against 20 k for the guess pass, profiles/r03_run33_pmc_sq_jpeg_sync_kernels.json).
        const int lane = threadIdx.x;
        bool active = !idle && bitpos() < end_bits && ordinal < total_blocks;
        while (__ballot(active)) {
            bool finished = false;
            if (active) {
                symbol_step();
                finished = z >= 64;
            }
            unsigned long long m = __ballot(finished && own && blk != nullptr);
            const unsigned long long bp = reinterpret_cast<unsigned long long>(blk);
            while (m) {
                const int L = __builtin_ctzll(m);
                m &= m - 1;
                const unsigned lo = (unsigned)pf_shfl_i32((int)(unsigned)bp, L), hi = (unsigned)pf_shfl_i32((int)(unsigned)(bp >> 32), L);
                short* dst = reinterpret_cast<short*>(((unsigned long long)hi << 32) | lo);
                short* src = blkbuf + L * BLK_STRIDE;
                dst[lane] = src[lane];
                src[lane] = 0;
            }
            if (finished) {
                ++ordinal;
                blk = ordinal < total_blocks ? block_ptr(ordinal) : nullptr;
                own = true;
                z = 0;
                bi = bi + 1 == a.bpm ? 0 : bi + 1;
            }
            active = active && bitpos() < end_bits && ordinal < total_blocks;
        }
        // the block still open at the exit continues in the next sub-sequence: hand over what was decoded here, entry by entry
        if (!idle && own && z > 0 && blk) {
            for (int k = 0; k < 64; ++k)
                if (mine[k]) blk[k] = mine[k];
        }
        return;
    }
    // A thread whose record turns out wrong walks on into the following sub-sequences (its exit is their better entry) until it
    // meets a record that agrees -- otherwise a stretch that does not re-synchronise (long blocks without end-of-block codes:
    // quality-100 noise) would move one sub-sequence per round.  Records it rewrites carry this round's stamp, so their successors
    // check themselves in the next round; whichever of two racing writers wins, the loser's successor is marked too.
    int cur = i;
    for (int walked = 0;; ++walked) {
        const size_t rc = (size_t)blockIdx.y * a.max_sub + cur;
        done_blocks = 0;
        decode_until(min((unsigned)(cur + 1) * PF_JPEG_SUBSEQ_BITS, total_bits));
        const unsigned ep = bitpos(), es = ((unsigned)bi << 8) | (unsigned)z;
        if (MODE == 1 && a.exit_p[rc] == ep && a.exit_s[rc] == es && a.nblk[rc] == done_blocks) break;
        a.exit_p[rc] = ep;
        a.exit_s[rc] = es;
        a.nblk[rc] = done_blocks;
        a.stamp[rc] = (unsigned)a.round;
        if (MODE == 1) atomicAdd(a.changed + a.round, 1u);
        // (round 1: every thread re-decodes anyway, and round 2 re-checks the successors of what changed -- a lane that walked would
        //  only stretch its wave)
        if (MODE == 0 || a.round == 1 || walked + 1 >= PF_JPEG_WALK || cur + 1 >= (int)d.n_intervals) break;
        ++cur;
    }
}

// exclusive prefix sum of the completed-block counts of one frame's sub-sequences (one workgroup per frame)
__global__ __launch_bounds__(256) void jpeg_subseq_scan_kernel(JpegSubseqArgs a) {
    const JpegFrameDesc d = a.desc[blockIdx.x];
    if (d.restart != 0 || d.n_intervals == 0) return;
    __shared__ unsigned part[256];
    const int n = (int)d.n_intervals, t = threadIdx.x;
    const int per = (n + 255) / 256;
    const size_t base = (size_t)blockIdx.x * a.max_sub;
    unsigned sum = 0;
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) sum += a.nblk[base + k];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        unsigned run = 0;
        for (int k = 0; k < 256; ++k) { const unsigned v = part[k]; part[k] = run; run += v; }
    }
    __syncthreads();
    unsigned run = part[t];
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) { a.blk0[base + k] = run; run += a.nblk[base + k]; }
}

// DC differences -> DC values: prefix sum over a component's blocks in scan order (one workgroup per (component, frame))
__global__ __launch_bounds__(256) void jpeg_dc_prefix_kernel(JpegSubseqArgs a) {
    const JpegFrameDesc d = a.desc[blockIdx.y];
    const int c = blockIdx.x;
    if (d.restart != 0 || d.n_intervals == 0 || c >= a.ncomp) return;
    __shared__ int part[256];
    const int nb = a.ch[c] * a.cv[c];
    const int n = a.total_mcus * nb, t = threadIdx.x;
    const int per = (n + 255) / 256;
    short* coef = a.coef + (size_t)blockIdx.y * a.blocks * 64;
    auto dc_of = [&](int k) -> short* {
        const int m = k / nb, r = k - m * nb;
        const int v = r / a.ch[c], hh = r - v * a.ch[c];
        const int my = m / a.mcux, mx = m - my * a.mcux;
        return coef + ((size_t)a.cblock0[c] + (size_t)(my * a.cv[c] + v) * a.cbw[c] + (mx * a.ch[c] + hh)) * 64;
    };
    int sum = 0;
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) sum += *dc_of(k);
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) { const int v = part[k]; part[k] = run; run += v; }
    }
    __syncthreads();
    int run = part[t];
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) {
        short* p = dc_of(k);
        run += *p;
        *p = (short)run;
    }
}

// a stream that needed more rounds than were queued must not pass silently: code 3 in the engine's status word
__global__ void jpeg_subseq_verdict_kernel(const unsigned* changed_last, int* status) {
    if (*changed_last != 0 && status[0] == 0) { status[1] = (int)*changed_last; status[0] = 3; }
}

struct JpegIdctArgs {
    const short* coef;            // [block][64] natural (row-major) order, NOT yet dequantised
    int ncomp;
    int block0[4];                // first block of each component (block0[ncomp] = total)
    int bw[3];                    // blocks per row of the component's plane
    unsigned char* plane[3];      // [bh * 8][bw * 8]
    const unsigned short* quant;  // [frame][3][64] natural order
    size_t frame_blocks;          // blockIdx.y = frame of a batch of equally shaped images: coefficient / plane stride in blocks
};

__device__ __forceinline__ int pf_jpeg_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// IJG sample_range_limit behind "& RANGE_MASK": the 10-bit value is a signed sample offset; +128, clamp to [0, 255]
__device__ __forceinline__ unsigned char pf_jpeg_range_limit(int v) {
    const int x = v & 1023;
    return (unsigned char)(x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896)));
}

__device__ __forceinline__ void pf_jpeg_idct_1d(const int (&in)[8], int (&out)[8], int shift) {
    constexpr int CONST_BITS = 13;
    // even part
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;                      // FIX_0_541196100
    int tmp2 = z1 + z3 * (-15137);                  // FIX_1_847759065
    int tmp3 = z1 + z2 * 6270;                      // FIX_0_765366865
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << CONST_BITS;
    int tmp1 = (z2 - z3) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    // odd part
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;                // FIX_1_175875602
    tmp0 *= 2446;                                   // FIX_0_298631336
    tmp1 *= 16819;                                  // FIX_2_053119869
    tmp2 *= 25172;                                  // FIX_3_072711026
    tmp3 *= 12299;                                  // FIX_1_501321110
    z1 *= -7373;                                    // FIX_0_899976223
    z2 *= -20995;                                   // FIX_2_562915447
    z3 *= -16069;                                   // FIX_1_961570560
    z4 *= -3196;                                    // FIX_0_390180644
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = pf_jpeg_descale(tmp10 + tmp3, shift); out[7] = pf_jpeg_descale(tmp10 - tmp3, shift);
    out[1] = pf_jpeg_descale(tmp11 + tmp2, shift); out[6] = pf_jpeg_descale(tmp11 - tmp2, shift);
    out[2] = pf_jpeg_descale(tmp12 + tmp1, shift); out[5] = pf_jpeg_descale(tmp12 - tmp1, shift);
    out[3] = pf_jpeg_descale(tmp13 + tmp0, shift); out[4] = pf_jpeg_descale(tmp13 - tmp0, shift);
}

__global__ __launch_bounds__(64) void jpeg_idct_kernel(JpegIdctArgs a) {
    const int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.block0[a.ncomp]) return;
    int c = 0;
    while (c + 1 < a.ncomp && blk >= a.block0[c + 1]) ++c;
    const int local = blk - a.block0[c];
    const int by = local / a.bw[c], bx = local - by * a.bw[c];
    const size_t fo = (size_t)blockIdx.y * a.frame_blocks;
    const short* __restrict__ src = a.coef + (fo + blk) * 64;
    const unsigned short* __restrict__ q = a.quant + ((size_t)blockIdx.y * 3 + c) * 64;
    int ws[8][8];                                    // [row][col]
    // pass 1: columns (results scaled up by 2^PASS1_BITS)
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)src[r * 8 + col] * (int)q[r * 8 + col];
        pf_jpeg_idct_1d(in, out, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = out[r];
    }
    // pass 2: rows, descale by 2^(CONST_BITS + PASS1_BITS + 3), +128 and clamp
    unsigned char* dst = a.plane[c] + fo * 64 + ((size_t)by * 8) * ((size_t)a.bw[c] * 8) + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int out[8];
        pf_jpeg_idct_1d(ws[r], out, 13 + 2 + 3);
        unsigned long long pk = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) pk |= (unsigned long long)pf_jpeg_range_limit(out[i]) << (8 * i);
        *reinterpret_cast<unsigned long long*>(dst + (size_t)r * a.bw[c] * 8) = pk;
    }
}

struct JpegColorArgs {
    const unsigned char* y;  int ys;       // luma plane and its row stride
    const unsigned char* cb; const unsigned char* cr; int cs;   // chroma planes (null for greyscale) and their stride
    int cw, ch;              // chroma plane size that exists (downsampled_width / height: edge replication starts there)
    int mode;                // 0 grey, 1 4:4:4, 2 h2v1, 3 h2v2; +4: plain replication instead of the triangle filter (width <= 2)
    int W, H;
    unsigned char* out;      // [frame][H][W][3] BGR
    size_t frame_plane_bytes;   // blockIdx.y = frame: stride of the plane set
};

// one chroma sample at output pixel (x, yy), IJG upsampling rules
__device__ __forceinline__ int pf_jpeg_chroma(const unsigned char* __restrict__ p, int cs, int cw, int ch, int mode, int x, int yy) {
    if (mode == 1) return p[(size_t)yy * cs + x];
    const int c = x >> 1;
    if (mode == 2 || mode == 6) {
        const unsigned char* r = p + (size_t)yy * cs;
        if (mode == 6) return r[c];
        const int v = r[c];
        if (x & 1) return c == cw - 1 ? v : (v * 3 + r[c + 1] + 2) >> 2;
        return c == 0 ? v : (v * 3 + r[c - 1] + 1) >> 2;
    }
    const int inrow = yy >> 1;
    if (mode == 7) return p[(size_t)inrow * cs + c];
    int other = (yy & 1) ? inrow + 1 : inrow - 1;     // upper output row leans on the row above, lower on the row below
    other = other < 0 ? 0 : (other > ch - 1 ? ch - 1 : other);
    const unsigned char* r0 = p + (size_t)inrow * cs;
    const unsigned char* r1 = p + (size_t)other * cs;
    const int cur = r0[c] * 3 + r1[c];
    if (x & 1) {
        if (c == cw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + (r0[c + 1] * 3 + r1[c + 1]) + 7) >> 4;
    }
    if (c == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[c - 1] * 3 + r1[c - 1]) + 8) >> 4;
}

__device__ __forceinline__ unsigned char pf_jpeg_clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegColorArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.W * a.H) return;
    const int yy = (int)(i / a.W), x = (int)(i - (long long)yy * a.W);
    const size_t po = (size_t)blockIdx.y * a.frame_plane_bytes;
    const int Y = a.y[po + (size_t)yy * a.ys + x];
    unsigned char* o = a.out + ((size_t)blockIdx.y * a.W * a.H + (size_t)i) * 3;
    if (a.mode == 0) { o[0] = o[1] = o[2] = (unsigned char)Y; return; }
    const int cb = pf_jpeg_chroma(a.cb + po, a.cs, a.cw, a.ch, a.mode, x, yy) - 128;
    const int cr = pf_jpeg_chroma(a.cr + po, a.cs, a.cw, a.ch, a.mode, x, yy) - 128;
    // jdcolor.c build_ycc_rgb_table, SCALEBITS 16: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
    const int r = Y + ((91881 * cr + 32768) >> 16);
    const int b = Y + ((116130 * cb + 32768) >> 16);
    const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    o[0] = pf_jpeg_clamp8(b); o[1] = pf_jpeg_clamp8(g); o[2] = pf_jpeg_clamp8(r);
}

// Four pixels of a row per thread (W % 4 == 0): one 4-byte luma load and three 4-byte stores instead of four and twelve single
// bytes -- the per-pixel kernel above ran at ~1 TB/s on byte stores.  Same arithmetic, pixel by pixel.
__global__ __launch_bounds__(256) void jpeg_color4_kernel(JpegColorArgs a) {
    const int W4 = a.W >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)W4 * a.H) return;
    const int yy = (int)(i / W4), x0 = (int)(i - (long long)yy * W4) * 4;
    const size_t po = (size_t)blockIdx.y * a.frame_plane_bytes;
    const unsigned y4 = *reinterpret_cast<const unsigned*>(a.y + po + (size_t)yy * a.ys + x0);
    unsigned char px[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int Y = (int)((y4 >> (8 * k)) & 255u);
        if (a.mode == 0) {
            px[3 * k] = px[3 * k + 1] = px[3 * k + 2] = (unsigned char)Y;
        } else {
            const int cb = pf_jpeg_chroma(a.cb + po, a.cs, a.cw, a.ch, a.mode, x0 + k, yy) - 128;
            const int cr = pf_jpeg_chroma(a.cr + po, a.cs, a.cw, a.ch, a.mode, x0 + k, yy) - 128;
            px[3 * k + 2] = pf_jpeg_clamp8(Y + ((91881 * cr + 32768) >> 16));
            px[3 * k] = pf_jpeg_clamp8(Y + ((116130 * cb + 32768) >> 16));
            px[3 * k + 1] = pf_jpeg_clamp8(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
        }
    }
    unsigned* o = reinterpret_cast<unsigned*>(a.out + ((size_t)blockIdx.y * a.W * a.H + (size_t)yy * a.W + x0) * 3);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        o[q] = (unsigned)px[4 * q] | ((unsigned)px[4 * q + 1] << 8) | ((unsigned)px[4 * q + 2] << 16) | ((unsigned)px[4 * q + 3] << 24);
}
