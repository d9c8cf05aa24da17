// Pre/post-processing kernels of the FaceAna pipeline (all HBM-bound byte/float shuffling).
//
//   letterbox_kernel      FaceDetector.preprocess        face_detector.py:45-63  (K1)
//   nms_kernel            xywh2xyxy + score filter + sort + greedy NMS + scale_coords
//                         face_detector.py:31-37,73-136 ; then FaceAna.sort_and_filter facer.py:120-142 (K4)
//   crop_params_kernel    box arithmetic of FaceLandmark.preprocess  face_landmark.py:74-93 (numpy-1.23
//                         promotion: float32 boxes, float64 for (1+2*extend)*w and every // 2)
//   crop_resize_kernel    zero-pad + crop + cv2.resize  face_landmark.py:79-98  (K5)
//
// cv2.resize(INTER_LINEAR, uint8) is reproduced bit-for-bit as OpenCV's fixed-point bilinear
// (11-bit coefficients, int32 horizontal pass, ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2)>>2 vertical
// pass, exact-2x shrink -> 2x2 box average) -- third-party algorithm, restated (see oracle/prepost.py).
// Float box arithmetic uses the non-contracting __f*_rn intrinsics so results equal numpy's.
#pragma once
#include "pf_common.h"


struct PipelineScratch {
    // set by the pipeline before running the landmark program so that hm_decode_kernel also
    // back-projects to frame coordinates (face_landmark.py:112-113); nullptr otherwise
    const float* d_crop_for_decode = nullptr;
    float* d_kps_for_decode = nullptr;
    // device scratch owned by the handle (allocated lazily by pipeline.inl)
    unsigned char* d_frames = nullptr; size_t frames_bytes = 0;
    unsigned char* d_letterbox = nullptr; size_t letterbox_bytes = 0;
    // resident current / previous frame of a video stream (pf_set_frame; FaceAna.diff_frames, facer.py:98-118)
    unsigned char* d_cur = nullptr; size_t cur_bytes = 0;
    unsigned char* d_prev = nullptr; size_t prev_bytes = 0;
    unsigned long long* d_diff_sum = nullptr;
    int cur_h = 0, cur_w = 0, prev_h = 0, prev_w = 0;
    bool have_cur = false, have_prev = false;
    unsigned char* d_crops = nullptr; size_t crops_bytes = 0;
    float* d_rows_planted = nullptr; size_t rows_planted_bytes = 0;
    float* d_keep_rows = nullptr;    // [F][max_keep][16]
    int* d_keep_count = nullptr;     // [F]
    float* d_sel_boxes = nullptr;    // [F][top_k][4]
    double* d_boxes64 = nullptr; size_t boxes64_bytes = 0;   // float64 boxes of pf_landmarks_f64
    int* d_sel_count = nullptr;      // [F]
    int* d_cand_count = nullptr;     // [F] candidates per frame (nms_compact_kernel)
    int* d_crop_params = nullptr;    // [faces][8]
    float* d_cropf = nullptr;        // [faces][5]
    float* d_kps = nullptr;          // [faces][98][2]
    unsigned long long* d_nms_keys = nullptr;  // [F][cap]
    unsigned char* d_nms_flags = nullptr;      // [F][cap]
    int cap_frames = 0, cap_faces = 0, cap_keep = 0, cap_topk = 0, cap_rows = 0;
    void release() {
        void* ptrs[] = {d_cur, d_prev, d_diff_sum, d_frames, d_letterbox, d_crops, d_rows_planted, d_boxes64, d_keep_rows, d_keep_count,
                        d_sel_boxes, d_sel_count, d_cand_count, d_crop_params, d_cropf, d_kps, d_nms_keys, d_nms_flags};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        *this = PipelineScratch();
    }
};

// --------------------------------------------------------------------------------------------
// OpenCV fixed-point bilinear taps
struct LinTap { int i0, i1, a0, a1; };

// horizontal flavour: weights reset at the borders (resize.cpp: "fx = 0, sx = 0" / "sx = ssize-1")
__device__ __forceinline__ LinTap pf_cv_tap_h(int d, double scale, int src_len) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src_len - 1) { s = src_len - 1; f = 0.f; }
    LinTap t;
    t.i0 = s;
    t.i1 = s + 1 < src_len ? s + 1 : src_len - 1;
    t.a0 = (int)rintf((1.f - f) * 2048.f);
    t.a1 = (int)rintf(f * 2048.f);
    return t;
}
// vertical flavour: weights kept, rows clamped
__device__ __forceinline__ LinTap pf_cv_tap_v(int d, double scale, int src_len) {
    float f = (float)((d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    LinTap t;
    t.i0 = s < 0 ? 0 : (s > src_len - 1 ? src_len - 1 : s);
    t.i1 = s + 1 < 0 ? 0 : (s + 1 > src_len - 1 ? src_len - 1 : s + 1);
    t.a0 = (int)rintf((1.f - f) * 2048.f);
    t.a1 = (int)rintf(f * 2048.f);
    return t;
}
__device__ __forceinline__ int pf_cv_vmix(int h0, int h1, int b0, int b1) {
    return ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xFF;
}

// --------------------------------------------------------------------------------------------
struct LetterboxArgs {
    const unsigned char* frames;  // [F][H][row_stride] BGR
    unsigned char* out;           // [F][outH][outW][3] RGB
    int F, H, W, row_stride, outH, outW, rw, rh, top, left;
    double scale_x, scale_y;      // 1/(rw/W), 1/(rh/H) as OpenCV derives them
    int pad_value;
    int keep_order;               // 1: no BGR -> RGB swap (plain cv2.resize, pf_resize)
};

__global__ __launch_bounds__(256) void letterbox_kernel(LetterboxArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (idx >= a.outH * a.outW) return;
    const int oy = idx / a.outW, ox = idx - oy * a.outW;
    unsigned char* o = a.out + ((size_t)f * a.outH * a.outW + idx) * 3;
    const int ry = oy - a.top, rx = ox - a.left;
    if ((unsigned)ry >= (unsigned)a.rh || (unsigned)rx >= (unsigned)a.rw) {
        o[0] = o[1] = o[2] = (unsigned char)a.pad_value;
        return;
    }
    const unsigned char* src = a.frames + (size_t)f * a.H * a.row_stride;
    int r[3];
    if (a.W == 2 * a.rw && a.H == 2 * a.rh) {
        const unsigned char* p0 = src + (size_t)(2 * ry) * a.row_stride + (size_t)(2 * rx) * 3;
        const unsigned char* p1 = p0 + a.row_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
    } else {
        const LinTap tx = pf_cv_tap_h(rx, a.scale_x, a.W);
        const LinTap ty = pf_cv_tap_v(ry, a.scale_y, a.H);
        const unsigned char* r0 = src + (size_t)ty.i0 * a.row_stride;
        const unsigned char* r1 = src + (size_t)ty.i1 * a.row_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[tx.i0 * 3 + c] * tx.a0 + r0[tx.i1 * 3 + c] * tx.a1;
            const int h1 = r1[tx.i0 * 3 + c] * tx.a0 + r1[tx.i1 * 3 + c] * tx.a1;
            r[c] = pf_cv_vmix(h0, h1, ty.a0, ty.a1);
        }
    }
    o[0] = (unsigned char)r[a.keep_order ? 0 : 2];  // BGR -> RGB (face_detector.py:47)
    o[1] = (unsigned char)r[1];
    o[2] = (unsigned char)r[a.keep_order ? 2 : 0];
}

// --------------------------------------------------------------------------------------------
struct NmsArgs {
    const float* rows;     // [F][R][16] decoded detector rows (cx,cy,w,h,score,...) letterboxed pixels
    float lb_scale, lb_left, lb_top;   // letterbox geometry (same for every frame of the call), face_detector.py:71
    float* keep_rows;      // [F][max_keep][16]  kept rows, xyxy un-letterboxed in cols 0:4
    int* keep_count;       // [F]
    float* sel_boxes;      // [F][top_k][4]  after sort_and_filter (may be nullptr)
    int* sel_count;        // [F]
    unsigned long long* keys;  // [F][cap] candidate keys (cap = power of two >= R), written by nms_compact_kernel
    unsigned char* flags;      // [F][cap] scratch (only used when a frame has more than PF_NMS_LDS_KEYS candidates)
    int* cand_count;           // [F] number of candidates, zeroed before nms_compact_kernel
    int R, cap, max_keep, top_k;
    float score_thres, iou_thres, min_face;
};

__device__ __forceinline__ unsigned pf_orderable(float v) {
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Stage 1 (many workgroups per frame): score filter.  Every workgroup scans 1024 rows, keeps score > thres (strict,
// face_detector.py:97) and appends a sortable key (orderable score << 32 | ~row) to the frame's candidate list through
// one atomicAdd per wave.  The list order is arbitrary; stage 2 sorts it, and keys are unique, so the result is not.
// 15 x F workgroups read the 968 KB of rows of a 384 x 640 frame instead of one workgroup per frame.
__global__ __launch_bounds__(1024) void nms_compact_kernel(NmsArgs a) {
    const int f = blockIdx.y;
    const int r = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const float* rows = a.rows + (size_t)f * a.R * 16;
    float sc = 0.f;
    bool pass = false;
    if (r < a.R) {
        sc = rows[(size_t)r * 16 + 4];
        pass = sc > a.score_thres;
    }
    const unsigned long long m = __ballot(pass);
    if (m == 0ull) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(a.cand_count + f, __popcll(m));
    base = pf_shfl_i32(base, 0);
    if (pass) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        a.keys[(size_t)f * a.cap + pos] = ((unsigned long long)pf_orderable(sc) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)r);
    }
}

// Stage 2: one 256-thread workgroup per frame.  Bitonic sort of the (score desc, row asc) keys -- in LDS when the frame
// has at most PF_NMS_LDS_KEYS candidates (the usual case: tens to hundreds), in the global key buffer otherwise -- then the
// reference's greedy loop with every surviving candidate tested in parallel per pick.
#define PF_NMS_LDS_KEYS 2048
__global__ __launch_bounds__(256) void nms_kernel(NmsArgs a) {
    constexpr int NT = 256;
    __shared__ unsigned long long s_keys[PF_NMS_LDS_KEYS];
    __shared__ unsigned char s_flags[PF_NMS_LDS_KEYS];
    __shared__ int s_nkeep;
    __shared__ int s_keep[1024];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const float* rows = a.rows + (size_t)f * a.R * 16;
    const int C = min(a.cand_count[f], a.cap);
    int n2 = 1;
    while (n2 < C) n2 <<= 1;
    const bool in_lds = n2 <= PF_NMS_LDS_KEYS;
    unsigned long long* keys = in_lds ? s_keys : a.keys + (size_t)f * a.cap;
    unsigned char* flags = in_lds ? s_flags : a.flags + (size_t)f * a.cap;
    if (tid == 0) s_nkeep = 0;
    if (in_lds)
        for (int i = tid; i < C; i += NT) s_keys[i] = a.keys[(size_t)f * a.cap + i];
    for (int i = C + tid; i < n2; i += NT) keys[i] = 0ull;
    for (int i = tid; i < n2; i += NT) flags[i] = 0;
    __syncthreads();
    // 2. bitonic sort, descending
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += NT) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = keys[i], y = keys[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[p] = x; }
                }
            }
            __syncthreads();
        }
    }
    // 3. greedy suppression (face_detector.py:110-134).  With the candidates in LDS their xyxy boxes are gathered once, in
    // sorted order, so a pick costs LDS latencies instead of two dependent global round trips.
    __shared__ float s_box[PF_NMS_LDS_KEYS > 1024 ? 1024 : PF_NMS_LDS_KEYS][4];
    const bool boxes_in_lds = C <= (int)(sizeof(s_box) / sizeof(s_box[0]));
    if (boxes_in_lds) {
        for (int i = tid; i < C; i += NT) {
            const float* b = rows + (size_t)(int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull)) * 16;
            const float hw = b[2] / 2.f, hh = b[3] / 2.f;
            s_box[i][0] = __fsub_rn(b[0], hw); s_box[i][1] = __fsub_rn(b[1], hh);
            s_box[i][2] = __fadd_rn(b[0], hw); s_box[i][3] = __fadd_rn(b[1], hh);
        }
        __syncthreads();
    }
    for (int i = 0; i < C; ++i) {
        if (flags[i]) continue;  // uniform: written before the barrier that ended the previous pick
        const int ri = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        float x1, y1, x2, y2;
        if (boxes_in_lds) { x1 = s_box[i][0]; y1 = s_box[i][1]; x2 = s_box[i][2]; y2 = s_box[i][3]; }
        else {
            const float* bi = rows + (size_t)ri * 16;
            const float hw = bi[2] / 2.f, hh = bi[3] / 2.f;
            x1 = __fsub_rn(bi[0], hw); y1 = __fsub_rn(bi[1], hh);
            x2 = __fadd_rn(bi[0], hw); y2 = __fadd_rn(bi[1], hh);
        }
        const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
        if (tid == 0) { if (s_nkeep < 1024) s_keep[s_nkeep] = ri; s_nkeep++; }
        for (int j = i + 1 + tid; j < C; j += NT) {
            if (flags[j]) continue;
            float u1, v1, u2, v2;
            if (boxes_in_lds) { u1 = s_box[j][0]; v1 = s_box[j][1]; u2 = s_box[j][2]; v2 = s_box[j][3]; }
            else {
                const int rj = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
                const float* bj = rows + (size_t)rj * 16;
                const float jw = bj[2] / 2.f, jh = bj[3] / 2.f;
                u1 = __fsub_rn(bj[0], jw); v1 = __fsub_rn(bj[1], jh);
                u2 = __fadd_rn(bj[0], jw); v2 = __fadd_rn(bj[1], jh);
            }
            const float iw = fmaxf(0.f, __fsub_rn(fminf(x2, u2), fmaxf(x1, u1)));
            const float ih = fmaxf(0.f, __fsub_rn(fminf(y2, v2), fmaxf(y1, v1)));
            const float inter = __fmul_rn(ih, iw);
            const float aj = __fmul_rn(__fsub_rn(v2, v1), __fsub_rn(u2, u1));
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area, aj), inter));
            if (!(iou < a.iou_thres)) flags[j] = 1;  // NaN is suppressed too, like np.where(iou < thr)
        }
        __syncthreads();
    }
    __syncthreads();
    // 4. emit kept rows with scale_coords applied (face_detector.py:37,82-93); the un-letterboxed boxes and their areas
    // stay in LDS for the selection below (its single thread would otherwise chase global-memory latencies)
    __shared__ float s_kbox[1024][4];
    __shared__ float s_area[1024];
    const int nk = s_nkeep < a.max_keep ? (s_nkeep < 1024 ? s_nkeep : 1024) : a.max_keep;
    const float scale = a.lb_scale, left = a.lb_left, top = a.lb_top;
    float* out = a.keep_rows + (size_t)f * a.max_keep * 16;
    for (int k = tid; k < nk; k += NT) {
        const float* b = rows + (size_t)s_keep[k] * 16;
        const float hw = b[2] / 2.f, hh = b[3] / 2.f;
        float* o = out + (size_t)k * 16;
        const float o0 = __fdiv_rn(__fsub_rn(__fsub_rn(b[0], hw), left), scale);
        const float o1 = __fdiv_rn(__fsub_rn(__fsub_rn(b[1], hh), top), scale);
        const float o2 = __fdiv_rn(__fsub_rn(__fadd_rn(b[0], hw), left), scale);
        const float o3 = __fdiv_rn(__fsub_rn(__fadd_rn(b[1], hh), top), scale);
        o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
        for (int c = 4; c < 16; ++c) o[c] = b[c];
        s_kbox[k][0] = o0; s_kbox[k][1] = o1; s_kbox[k][2] = o2; s_kbox[k][3] = o3;
        s_area[k] = __fmul_rn(__fsub_rn(o2, o0), __fsub_rn(o3, o1));
    }
    if (tid == 0) a.keep_count[f] = nk;
    __syncthreads();
    // 5. FaceAna.sort_and_filter (facer.py:120-142): area > min_face, top_k by area if more remain
    if (a.sel_boxes && tid == 0) {
        float* sb = a.sel_boxes + (size_t)f * a.top_k * 4;
        int nsel = 0, npass = 0;
        for (int k = 0; k < nk; ++k) npass += s_area[k] > a.min_face ? 1 : 0;
        if (npass <= a.top_k) {
            for (int k = 0; k < nk; ++k) {
                if (s_area[k] > a.min_face) {
                    for (int c = 0; c < 4; ++c) sb[nsel * 4 + c] = s_kbox[k][c];
                    nsel++;
                }
            }
        } else {
            float last_area = 3.0e38f;
            int last_k = -1;
            for (int s = 0; s < a.top_k; ++s) {  // selection by descending area (ties: later index first,
                float best = -1.f;               // i.e. the reversed ascending argsort of the reference)
                int bk = -1;
                for (int k = nk - 1; k >= 0; --k) {
                    const float ar = s_area[k];
                    if (!(ar > a.min_face)) continue;
                    const bool before = ar > last_area || (ar == last_area && k >= last_k);
                    if (before) continue;
                    if (ar > best) { best = ar; bk = k; }
                }
                if (bk < 0) break;
                for (int c = 0; c < 4; ++c) sb[nsel * 4 + c] = s_kbox[bk][c];
                nsel++;
                last_area = best;
                last_k = bk;
            }
        }
        a.sel_count[f] = nsel;
    }
}

// --------------------------------------------------------------------------------------------
struct CropParamArgs {
    const float* boxes;   // [n][4] xyxy float32 (frame coordinates)
    const double* boxes64;  // or [n][4] float64 rows (tracked frames: FaceAna.track_box is float64, facer.py:75-81) -- then
                            // every step of face_landmark.py:74-93 is float64 arithmetic, no float32 store-back rounding
    const int* counts;    // optional [F]: faces actually present per frame (slots beyond are invalid)
    int* params;          // [n][8] = valid, add, x0, y0, xs, ys, w_crop, h_crop
    float* cropf;         // [n][5] = w_crop, h_crop, x0, y0, add  (face_landmark.py:104 "detail")
    int n, per_frame, H, W;
    float min_face;       // 20 (face_landmark.py:26)
    double width_factor;  // 1 + 2*extend[0]  (face_landmark.py:83)
    const int* boxes64_f32;   // device flag: the rows of boxes64 hold float32 values and numpy would compute in float32 (k_track.h)
};

__global__ __launch_bounds__(64) void crop_params_kernel(CropParamArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    int* p = a.params + (size_t)i * 8;
    float* cf = a.cropf + (size_t)i * 5;
    for (int k = 0; k < 8; ++k) p[k] = 0;
    for (int k = 0; k < 5; ++k) cf[k] = 0.f;
    if (a.counts && (i % a.per_frame) >= a.counts[i / a.per_frame]) return;
    float bf[4];
    const bool as_f32 = a.boxes64 && a.boxes64_f32 && *a.boxes64_f32 != 0;
    if (as_f32) for (int k = 0; k < 4; ++k) bf[k] = (float)a.boxes64[(size_t)i * 4 + k];
    if (a.boxes64 && !as_f32) {
        const double* b = a.boxes64 + (size_t)i * 4;
        const double w = b[2] - b[0], h = b[3] - b[1];
        if (w <= (double)a.min_face || h <= (double)a.min_face || !(w == w) || !(h == h)) return;
        const int add = (int)fmax(w, h);
        const double da = (double)add;
        const double b0 = b[0] + da, b1 = b[1] + da, b2 = b[2] + da, b3 = b[3] + da;
        const double cx = floor((b0 + b2) / 2.0), cy = floor((b1 + b3) / 2.0);
        const double half = floor(a.width_factor * w / 2.0);
        const int x0 = (int)(cx - half), y0 = (int)(cy - half), x1 = (int)(cx + half), y1 = (int)(cy + half);
        const int ph = a.H + 2 * add, pw = a.W + 2 * add;
        const int xs = min(max(x0, 0), pw), xe = min(max(x1, 0), pw);
        const int ys = min(max(y0, 0), ph), ye = min(max(y1, 0), ph);
        const int wc = xe - xs, hc = ye - ys;
        if (wc <= 0 || hc <= 0) return;
        p[0] = 1; p[1] = add; p[2] = x0; p[3] = y0; p[4] = xs; p[5] = ys; p[6] = wc; p[7] = hc;
        cf[0] = (float)wc; cf[1] = (float)hc; cf[2] = (float)x0; cf[3] = (float)y0; cf[4] = (float)add;
        return;
    }
    const float* b = as_f32 ? bf : a.boxes + (size_t)i * 4;
    const float w = __fsub_rn(b[2], b[0]), h = __fsub_rn(b[3], b[1]);
    if (w <= a.min_face || h <= a.min_face || !(w == w) || !(h == h)) return;
    const int add = (int)fmaxf(w, h);
    const float fa = (float)add;
    const float b0 = __fadd_rn(b[0], fa), b1 = __fadd_rn(b[1], fa), b2 = __fadd_rn(b[2], fa), b3 = __fadd_rn(b[3], fa);
    const double face_width = a.width_factor * (double)w;
    const double cx = floor((double)__fadd_rn(b0, b2) / 2.0);
    const double cy = floor((double)__fadd_rn(b1, b3) / 2.0);
    const double half = floor(face_width / 2.0);
    const int x0 = (int)(float)(cx - half), y0 = (int)(float)(cy - half);
    const int x1 = (int)(float)(cx + half), y1 = (int)(float)(cy + half);
    const int ph = a.H + 2 * add, pw = a.W + 2 * add;
    const int xs = min(max(x0, 0), pw), xe = min(max(x1, 0), pw);
    const int ys = min(max(y0, 0), ph), ye = min(max(y1, 0), ph);
    const int wc = xe - xs, hc = ye - ys;
    if (wc <= 0 || hc <= 0) return;
    p[0] = 1; p[1] = add; p[2] = x0; p[3] = y0; p[4] = xs; p[5] = ys; p[6] = wc; p[7] = hc;
    cf[0] = (float)wc; cf[1] = (float)hc; cf[2] = (float)x0; cf[3] = (float)y0; cf[4] = fa;
}

// --------------------------------------------------------------------------------------------
struct CropResizeArgs {
    const unsigned char* frames;  // [F][H][row_stride] BGR
    const int* params;            // [n][8]
    unsigned char* out;           // [n][S][S][3]  (channel order untouched, face_landmark.py:42-46)
    int n, per_frame, H, W, row_stride, S;
};

__device__ __forceinline__ int pf_padded_px(const unsigned char* src, int row_stride, int H, int W, int py, int px, int add, int c) {
    const int y = py - add, x = px - add;  // padded -> frame coordinates; outside = 0 (copyMakeBorder constant 0)
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return 0;
    return src[(size_t)y * row_stride + (size_t)x * 3 + c];
}

// One output pixel straight from the frame (byte loads): FaceLandmark.preprocess's zero-pad + slice + cv2.resize.
__device__ __forceinline__ void pf_crop_pixel_direct(const CropResizeArgs& a, const int* p, const unsigned char* src, int dy, int dx,
                                                     unsigned char* o) {
    const int add = p[1], xs = p[4], ys = p[5], wc = p[6], hc = p[7];
    if (wc == 2 * a.S && hc == 2 * a.S) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s = pf_padded_px(src, a.row_stride, a.H, a.W, ys + 2 * dy, xs + 2 * dx, add, c) +
                          pf_padded_px(src, a.row_stride, a.H, a.W, ys + 2 * dy, xs + 2 * dx + 1, add, c) +
                          pf_padded_px(src, a.row_stride, a.H, a.W, ys + 2 * dy + 1, xs + 2 * dx, add, c) +
                          pf_padded_px(src, a.row_stride, a.H, a.W, ys + 2 * dy + 1, xs + 2 * dx + 1, add, c);
            o[c] = (unsigned char)((s + 2) >> 2);
        }
        return;
    }
    const double scale_x = 1.0 / ((double)a.S / (double)wc);
    const double scale_y = 1.0 / ((double)a.S / (double)hc);
    const LinTap tx = pf_cv_tap_h(dx, scale_x, wc);
    const LinTap ty = pf_cv_tap_v(dy, scale_y, hc);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = pf_padded_px(src, a.row_stride, a.H, a.W, ys + ty.i0, xs + tx.i0, add, c) * tx.a0 +
                       pf_padded_px(src, a.row_stride, a.H, a.W, ys + ty.i0, xs + tx.i1, add, c) * tx.a1;
        const int h1 = pf_padded_px(src, a.row_stride, a.H, a.W, ys + ty.i1, xs + tx.i0, add, c) * tx.a0 +
                       pf_padded_px(src, a.row_stride, a.H, a.W, ys + ty.i1, xs + tx.i1, add, c) * tx.a1;
        o[c] = (unsigned char)pf_cv_vmix(h0, h1, ty.a0, ty.a1);
    }
}

// General path (output sizes the tiled kernel does not take): one thread per output pixel.
__global__ __launch_bounds__(256) void crop_resize_direct_kernel(CropResizeArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int face = blockIdx.y;
    if (idx >= a.S * a.S) return;
    const int* p = a.params + (size_t)face * 8;
    unsigned char* o = a.out + ((size_t)face * a.S * a.S + idx) * 3;
    if (!p[0]) { o[0] = o[1] = o[2] = 0; return; }
    const unsigned char* src = a.frames + (size_t)(face / a.per_frame) * a.H * a.row_stride;
    const int dy = idx / a.S;
    pf_crop_pixel_direct(a, p, src, dy, idx - dy * a.S, o);
}

// Tiled path: a workgroup produces PF_CROP_TY output rows of one face.  The source rows those outputs touch
// (hc / S * TY + 2 rows of wc pixels) are fetched ONCE, as aligned 32-bit words with the zero border of
// copyMakeBorder materialised, into LDS; the fixed-point taps read them there, the finished rows are assembled in LDS
// too and leave as aligned 32-bit words.  The per-pixel path issues 12 one-byte global loads and 3 one-byte stores
// per output pixel (0.88 TB/s = 11 % of HBM peak on 280 -> 256 crops in round 1).  A tile whose source rows do not fit
// the LDS budget (a very large face) computes its pixels the per-pixel way, decided per workgroup on the device.
#define PF_CROP_TY 8
#define PF_CROP_SRC_BYTES 16384
__global__ __launch_bounds__(256) void crop_resize_kernel(CropResizeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_src[PF_CROP_SRC_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char s_out[PF_CROP_TY * 256 * 3];   // S <= 256 (host)
    const int t = threadIdx.x;
    const int face = blockIdx.y;
    const int oy0 = blockIdx.x * PF_CROP_TY;
    const int S = a.S;
    const int* p = a.params + (size_t)face * 8;
    const int rows_out = min(PF_CROP_TY, S - oy0);
    unsigned char* out = a.out + ((size_t)face * S + oy0) * S * 3;
    const int out_words = rows_out * S * 3 / 4;          // S * 3 is a multiple of 4 for every S the host sends here
    if (!p[0]) {
        for (int i = t; i < out_words; i += 256) reinterpret_cast<unsigned*>(out)[i] = 0u;
        return;
    }
    const int add = p[1], xs = p[4], ys = p[5], wc = p[6], hc = p[7];
    const unsigned char* src = a.frames + (size_t)(face / a.per_frame) * a.H * a.row_stride;
    const bool exact2 = wc == 2 * S && hc == 2 * S;
    const double scale_x = 1.0 / ((double)S / (double)wc);
    const double scale_y = 1.0 / ((double)S / (double)hc);
    // source rows [r0, r1] (crop coordinates) this tile's outputs read
    int r0, r1;
    if (exact2) { r0 = 2 * oy0; r1 = 2 * (oy0 + rows_out) - 1; }
    else {
        const LinTap ta = pf_cv_tap_v(oy0, scale_y, hc), tb = pf_cv_tap_v(oy0 + rows_out - 1, scale_y, hc);
        r0 = min(ta.i0, ta.i1); r1 = max(tb.i0, tb.i1);
    }
    const int nrows = r1 - r0 + 1;
    const int row_bytes = wc * 3;
    const int lds_stride = (row_bytes + 3 + 3) & ~3;     // room for the alignment shift, multiple of 4
    if ((long long)nrows * lds_stride > PF_CROP_SRC_BYTES) {   // a very large face: its source rows do not fit -- per-pixel path
        for (int i = t; i < rows_out * S; i += 256) {
            const int dy = i / S;
            pf_crop_pixel_direct(a, p, src, oy0 + dy, i - dy * S, s_out + (size_t)i * 3);
        }
        __syncthreads();
        for (int i = t; i < out_words; i += 256) reinterpret_cast<unsigned*>(out)[i] = reinterpret_cast<const unsigned*>(s_out)[i];
        return;
    }
    // stage: row r of the crop = padded row ys + r0 + r -> frame row (ys + r0 + r - add), bytes [(xs - add) * 3, +row_bytes)
    const long long fx0 = (long long)(xs - add) * 3;     // first byte of the crop inside a frame row (may be negative)
    const int shift = (int)(((fx0 % 4) + 4) % 4);        // crop byte b lives at LDS byte shift + b of its row
    const long long fxa = fx0 - shift;                   // 4-byte aligned start (frame rows are row_stride apart)
    const int words = (shift + row_bytes + 3) / 4;
    const long long frame_bytes = (long long)a.H * a.row_stride;
    const bool rows_aligned = (a.row_stride & 3) == 0 && ((size_t)src & 3) == 0;
    for (int i = t; i < nrows * words; i += 256) {
        const int r = i / words, wd = i - r * words;
        const int fy = ys + r0 + r - add;
        const long long bx = fxa + 4LL * wd;             // byte offset inside the frame row
        unsigned v = 0u;
        if ((unsigned)fy < (unsigned)a.H) {
            const long long off = (long long)fy * a.row_stride + bx;
            if (rows_aligned && bx >= 0 && bx + 4 <= (long long)a.W * 3 && off + 4 <= frame_bytes) {
                v = *reinterpret_cast<const unsigned*>(src + off);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long long b = bx + k;
                    if (b >= 0 && b < (long long)a.W * 3) v |= (unsigned)src[(long long)fy * a.row_stride + b] << (8 * k);
                }
            }
        }
        *reinterpret_cast<unsigned*>(s_src + (size_t)r * lds_stride + 4 * wd) = v;
    }
    // the vertical taps of the tile's rows, once per ROW instead of once per pixel (round 5): two double-precision operations and a
    // floor each, identical for the 256 pixels of a row -- and this kernel is VALU-bound (83 % busy, profiles/r05_run37)
    __shared__ LinTap s_ty[PF_CROP_TY];
    if (!exact2 && t < rows_out) s_ty[t] = pf_cv_tap_v(oy0 + t, scale_y, hc);
    __syncthreads();
    // 256 % S == 0 (host): a thread keeps its output column for every row it computes, so its horizontal taps (two
    // double-precision operations each, like OpenCV derives them) are computed once
    const int dx = t % S;
    const LinTap tx = pf_cv_tap_h(dx, scale_x, wc);
    const unsigned m_s = pf_div_magic(S);                // i < PF_CROP_TY * 256, S <= 256: inside pf_div_small's exact range
    for (int i = t; i < rows_out * S; i += 256) {
        const int dy = pf_div_small(i, m_s);
        unsigned char* o = s_out + (size_t)i * 3;
        if (exact2) {
            const unsigned char* p0 = s_src + (size_t)(2 * dy) * lds_stride + shift + (2 * dx) * 3;   // r0 = 2 * oy0
            const unsigned char* p1 = p0 + lds_stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (unsigned char)((p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2);
        } else {
            const LinTap ty = s_ty[dy];
            const unsigned char* q0 = s_src + (size_t)(ty.i0 - r0) * lds_stride + shift;
            const unsigned char* q1 = s_src + (size_t)(ty.i1 - r0) * lds_stride + shift;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = q0[tx.i0 * 3 + c] * tx.a0 + q0[tx.i1 * 3 + c] * tx.a1;
                const int h1 = q1[tx.i0 * 3 + c] * tx.a0 + q1[tx.i1 * 3 + c] * tx.a1;
                o[c] = (unsigned char)pf_cv_vmix(h0, h1, ty.a0, ty.a1);
            }
        }
    }
    __syncthreads();
    for (int i = t; i < out_words; i += 256) reinterpret_cast<unsigned*>(out)[i] = reinterpret_cast<const unsigned*>(s_out)[i];
}

// --------------------------------------------------------------------------------------------
// K13: frame-difference gate of FaceAna.diff_frames (facer.py:111-115): sum |prev - cur| over all bytes.
// Integer sum (exact, order independent); the host divides by H*W*3 like the reference does.
struct AbsDiffArgs {
    const unsigned char* a;
    const unsigned char* b;
    unsigned long long* sum;   // zeroed before the launch
    size_t n;                  // bytes
};

__global__ __launch_bounds__(256) void absdiff_sum_kernel(AbsDiffArgs a) {
    __shared__ unsigned int s_part[4];
    const size_t nvec = a.n / 16;
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const uint4 x = *reinterpret_cast<const uint4*>(a.a + i * 16);
        const uint4 y = *reinterpret_cast<const uint4*>(a.b + i * 16);
        const unsigned int xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = (int)((xs[w] >> (8 * k)) & 0xFF) - (int)((ys[w] >> (8 * k)) & 0xFF);
                acc += (unsigned int)(d < 0 ? -d : d);
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * 16; i < a.n; ++i) { const int d = (int)a.a[i] - (int)a.b[i]; acc += (unsigned int)(d < 0 ? -d : d); }
    for (int mask = 1; mask < 64; mask <<= 1) acc += (unsigned int)pf_shfl_xor_i32((int)acc, mask);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.sum, (unsigned long long)s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
