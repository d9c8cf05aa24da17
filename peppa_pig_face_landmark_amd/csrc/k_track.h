// Tracking state of one video stream on the device.
//
// FaceAna.run() keeps three pieces of state between frames (Skps/core/api/facer.py:40-42, core/smoother/lk.py:8-9):
// the boxes of the previous frame (track_box), the previous landmark sets and their last displacement.  Between the
// detector and the landmark regressor it matches boxes by IoU and smooths them (judge_boxs :144-189), keeps the top-k by
// area (sort_and_filter :120-142); after the regressor it smooths the landmarks with a One-Euro filter against the
// previous set (GroupTrack.calculate, lk.py:19-56,117-149) and derives the next track boxes from their hulls (:70-81).
// On the host that costs a device->host->device trip of the boxes between the two networks on every frame.  The
// kernels below keep all of it in HBM; every one is a single small workgroup (a stream has at most top_k faces).
//
// Values are STORED as float64, but the arithmetic follows the dtype numpy would be working in (round 3): detector rows and
// network landmarks are float32, `landmarks / [w, h]` (lk.py:39-41) promotes a smoothed landmark set -- and from there the
// hull boxes, the EMA-smoothed track boxes and the next frame's crop arithmetic -- to float64, and np.array([...]) of mixed
// rows is float64 as soon as one row is.  So a frame's boxes are float32 on a detector frame with nothing to smooth against
// and float64 once the stream is running; IoU, EMA, areas and FaceLandmark.preprocess round differently in the two (a crop
// edge lands on the other side of an integer), hence three dtype flags travel with the state: track boxes, current boxes,
// landmark sets.  A float32 value is exactly representable in its float64 slot, so only the operations need the flag.
#pragma once
#include "pf_common.h"

struct TrackState {
    double* d_track_box = nullptr;     // [cap][4] boxes of the previous frame (FaceAna.track_box)
    int* d_n_track = nullptr;          // rows of it
    double* d_judged = nullptr;        // [1024][4] judge_boxs(track_box, detector boxes)
    int* d_n_judged = nullptr;
    double* d_sel = nullptr;           // [top_k][4] sort_and_filter output == boxes_return, fed to the landmark stage
    int* d_n_sel = nullptr;            // (int[1]: doubles as the per-frame count of the landmark stage)
    double* d_lm[2] = {nullptr, nullptr};   // [top_k][98][2] previous / new landmark sets (ping-pong)
    double* d_dx[2] = {nullptr, nullptr};   // [top_k][98][2] previous_dx
    int* d_n_lm[2] = {nullptr, nullptr};    // rows of each
    double* d_hull = nullptr;          // [top_k][4] hull boxes of the new landmark sets (tmp_box)
    float* d_scores = nullptr;         // [top_k][98] scores of the valid faces, compacted like the landmarks
    // dtype flags (1 = float32, 0 = float64), on the device because they depend on which rows matched:
    // [0] track_box  [1] boxes of this frame (judge_boxs output / boxes_return)  [2], [3] landmark sets (ping-pong, like d_lm)
    int* d_f32 = nullptr;
    int cur = 0;                       // which of the ping-pong buffers holds the PREVIOUS sets
    int top_k = 0;
    bool has_track = false;            // track_box is not None
    bool lm_valid = false;             // trace.previous_landmarks_set is not None
    void release() {
        void* ptrs[] = {d_track_box, d_n_track, d_judged, d_n_judged, d_sel, d_n_sel, d_lm[0], d_lm[1], d_dx[0], d_dx[1],
                        d_n_lm[0], d_n_lm[1], d_hull, d_scores, d_f32};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        *this = TrackState();
    }
};

__device__ __forceinline__ double pf_box_iou_f64(const double* a, const double* b) {   // facer.py:152-172, lk.py:58-80
    const double s1 = (a[2] - a[0]) * (a[3] - a[1]);
    const double s2 = (b[2] - b[0]) * (b[3] - b[1]);
    const double x1 = fmax(a[0], b[0]), y1 = fmax(a[1], b[1]);
    const double x2 = fmin(a[2], b[2]), y2 = fmin(a[3], b[3]);
    const double inter = fmax(0.0, x2 - x1) * fmax(0.0, y2 - y1);
    return inter / (s1 + s2 - inter);
}
// the same on float32 scalars (both boxes float32: numpy keeps float32 scalar arithmetic in float32); no FMA contraction
__device__ __forceinline__ double pf_box_iou_f32(const double* a, const double* b) {
    const float a0 = (float)a[0], a1 = (float)a[1], a2 = (float)a[2], a3 = (float)a[3];
    const float b0 = (float)b[0], b1 = (float)b[1], b2 = (float)b[2], b3 = (float)b[3];
    const float s1 = __fmul_rn(__fsub_rn(a2, a0), __fsub_rn(a3, a1));
    const float s2 = __fmul_rn(__fsub_rn(b2, b0), __fsub_rn(b3, b1));
    const float x1 = fmaxf(a0, b0), y1 = fmaxf(a1, b1), x2 = fminf(a2, b2), y2 = fminf(a3, b3);
    const float inter = __fmul_rn(fmaxf(0.f, __fsub_rn(x2, x1)), fmaxf(0.f, __fsub_rn(y2, y1)));   // max(0, f32) stays f32
    return (double)__fdiv_rn(inter, __fsub_rn(__fadd_rn(s1, s2), inter));
}
__device__ __forceinline__ double pf_box_iou(const double* a, int a32, const double* b, int b32) {
    return (a32 && b32) ? pf_box_iou_f32(a, b) : pf_box_iou_f64(a, b);
}
// exponential_smoothing(alpha, x, x_prev) = alpha * x + (1 - alpha) * x_prev on ARRAYS (lk.py:96-97): a python float times a
// float32 array is a float32 product, the sum is float32 only if both terms are
__device__ __forceinline__ double pf_ema(double alpha, double x, int x32, double p, int p32) {
    const double om = 1.0 - alpha;
    if (x32 && p32) return (double)__fadd_rn(__fmul_rn((float)alpha, (float)x), __fmul_rn((float)om, (float)p));
    const double t1 = x32 ? (double)__fmul_rn((float)alpha, (float)x) : alpha * x;
    const double t2 = p32 ? (double)__fmul_rn((float)om, (float)p) : om * p;
    return t1 + t2;
}

// judge_boxs (facer.py:144-189): every current box is matched against the FIRST previous box with IoU > thres and
// EMA-smoothed with it (alpha * now + (1 - alpha) * previous), or passed through.  The result array is float32 only if
// the current rows are and no float64 previous row was mixed in.
struct JudgeArgs {
    const double* prev; const int* n_prev; int has_prev;      // has_prev == 0: previous is None -> pass through
    const int* prev_f32;                                      // dtype flag of prev (device), nullptr = float64
    const float* now_f32; int now_stride;                     // detector rows (float32, stride 16) ...
    const double* now_f64;                                    // ... or rows of 4 stored as float64,
    const int* now_f32_flag;                                  //     whose dtype flag is here (nullptr = float64)
    const int* n_now;
    double* out; int* n_out;
    int* out_f32;                                             // dtype flag of the result
    double iou_thres, alpha;
    int max_now;
};

__global__ __launch_bounds__(256) void track_judge_kernel(JudgeArgs a) {
    __shared__ int s_mixed;                                   // some row was smoothed against a float64 previous row
    if (threadIdx.x == 0) s_mixed = 0;
    __syncthreads();
    const int n = min(*a.n_now, a.max_now);
    const int np = a.has_prev ? *a.n_prev : 0;
    const int n32 = a.now_f64 ? (a.now_f32_flag ? *a.now_f32_flag : 0) : 1;
    const int p32 = a.prev_f32 ? *a.prev_f32 : 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        double b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = a.now_f64 ? a.now_f64[(size_t)i * 4 + c] : (double)a.now_f32[(size_t)i * a.now_stride + c];
        for (int j = 0; j < np; ++j) {
            const double* p = a.prev + (size_t)j * 4;
            if (pf_box_iou(b, n32, p, p32) > a.iou_thres) {
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = pf_ema(a.alpha, b[c], n32, p[c], p32);
                if (!p32) s_mixed = 1;
                break;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) a.out[(size_t)i * 4 + c] = b[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) { *a.n_out = n; *a.out_f32 = (n32 && !s_mixed) ? 1 : 0; }
}

// sort_and_filter (facer.py:120-142): drop area <= min_face, keep the top_k largest (descending; equal areas: the
// later row first, the reversed ascending argsort of the reference).
struct SelectArgs {
    const double* boxes; const int* n; double* out; int* n_out;
    const int* boxes_f32;      // dtype flag of the rows (their areas are float32 products then)
    double min_face; int top_k;
};

__device__ __forceinline__ double pf_box_area(const double* b, int f32) {
    return f32 ? (double)__fmul_rn(__fsub_rn((float)b[2], (float)b[0]), __fsub_rn((float)b[3], (float)b[1])) : (b[2] - b[0]) * (b[3] - b[1]);
}

__global__ __launch_bounds__(64) void track_select_kernel(SelectArgs a) {
    if (threadIdx.x != 0) return;
    const int n = *a.n;
    const int f32 = a.boxes_f32 ? *a.boxes_f32 : 0;
    int npass = 0;
    for (int k = 0; k < n; ++k) npass += pf_box_area(a.boxes + (size_t)k * 4, f32) > a.min_face ? 1 : 0;
    int nsel = 0;
    if (npass <= a.top_k) {
        for (int k = 0; k < n; ++k) {
            const double* b = a.boxes + (size_t)k * 4;
            if (pf_box_area(b, f32) > a.min_face) {
                for (int c = 0; c < 4; ++c) a.out[nsel * 4 + c] = b[c];
                nsel++;
            }
        }
    } else {
        double last_area = 1.0e300;
        int last_k = -1;
        for (int s = 0; s < a.top_k; ++s) {
            double best = -1.0;
            int bk = -1;
            for (int k = n - 1; k >= 0; --k) {
                const double* b = a.boxes + (size_t)k * 4;
                const double ar = pf_box_area(b, f32);
                if (!(ar > a.min_face)) continue;
                if (ar > last_area || (ar == last_area && k >= last_k)) continue;
                if (ar > best) { best = ar; bk = k; }
            }
            if (bk < 0) break;
            for (int c = 0; c < 4; ++c) a.out[nsel * 4 + c] = a.boxes[(size_t)bk * 4 + c];
            nsel++;
            last_area = best;
            last_k = bk;
        }
    }
    *a.n_out = nsel;
}

// GroupTrack.calculate (lk.py:19-56) + OneEuroFilter.__call__ (lk.py:117-149) + the hull boxes of facer.py:70-74.
// One workgroup per face slot of this frame; faces the crop stage rejected (params[slot][0] == 0) are dropped and the
// survivors compacted, like `landmarks[valid]` on the host.
struct GroupTrackArgs {
    const float* kps;         // [top_k][98][2] float32 landmarks of this frame (frame coordinates)
    const float* scores_in;   // [top_k][98]
    const int* crop_params;   // [top_k][8], [0] = valid
    const int* n_sel;         // face slots in use this frame
    const double* prev_lm; const double* prev_dx; const int* n_prev; int prev_valid;
    const int* prev_f32;      // dtype flag of the previous landmark sets
    int* out_f32;             // dtype flag of the new sets: set to 1 before the launch, cleared by any face that was smoothed
    double* out_lm; double* out_dx; int* n_out;
    double* hull;             // [top_k][4]
    float* scores_out;        // [top_k][98]
    double iou_thres, scale_w, scale_h;
    double min_cutoff, beta, d_cutoff;
};

__device__ __forceinline__ void pf_hull_98(const double* pts, int tid, double* s_red, double* box) {
    // min / max over 98 points by the first 128 threads (s_red: 4 x 128 doubles)
    double mnx = 1.0e300, mny = 1.0e300, mxx = -1.0e300, mxy = -1.0e300;
    if (tid < 98) { mnx = mxx = pts[2 * tid]; mny = mxy = pts[2 * tid + 1]; }
    if (tid < 128) { s_red[tid] = mnx; s_red[128 + tid] = mny; s_red[256 + tid] = mxx; s_red[384 + tid] = mxy; }
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (tid < s) {
            s_red[tid] = fmin(s_red[tid], s_red[tid + s]);
            s_red[128 + tid] = fmin(s_red[128 + tid], s_red[128 + tid + s]);
            s_red[256 + tid] = fmax(s_red[256 + tid], s_red[256 + tid + s]);
            s_red[384 + tid] = fmax(s_red[384 + tid], s_red[384 + tid + s]);
        }
        __syncthreads();
    }
    box[0] = s_red[0]; box[1] = s_red[128]; box[2] = s_red[256]; box[3] = s_red[384];
    __syncthreads();
}

__global__ __launch_bounds__(128) void track_group_kernel(GroupTrackArgs a) {
    __shared__ double s_red[512];
    __shared__ double s_now[196];
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int nsel = *a.n_sel;
    if (slot >= nsel || a.crop_params[(size_t)slot * 8] == 0) return;      // uniform per workgroup
    int oi = 0;                                                            // output row = valid slots before this one
    for (int k = 0; k < slot; ++k) oi += a.crop_params[(size_t)k * 8] != 0 ? 1 : 0;
    for (int i = tid; i < 196; i += 128) s_now[i] = (double)a.kps[(size_t)slot * 196 + i];
    __syncthreads();
    double nbox[4];
    pf_hull_98(s_now, tid, s_red, nbox);
    int match = -1;
    if (a.prev_valid) {
        const int np = *a.n_prev;
        const int p32 = *a.prev_f32;
        for (int j = 0; j < np && match < 0; ++j) {
            double pbox[4];
            pf_hull_98(a.prev_lm + (size_t)j * 196, tid, s_red, pbox);
            if (pf_box_iou(nbox, 1, pbox, p32) > a.iou_thres) match = j;      // this frame's landmarks are float32
        }
    }
    if (match >= 0 && tid == 0) *a.out_f32 = 0;                 // `/ scale` made this set float64, hence the whole array
    const double two_pi = 2.0 * 3.141592653589793;
    if (tid < 98) {
        double rx = s_now[2 * tid], ry = s_now[2 * tid + 1], ddx = 0.0, ddy = 0.0;
        if (match >= 0) {
            const double* pl = a.prev_lm + (size_t)match * 196 + 2 * tid;
            const double* pd = a.prev_dx + (size_t)match * 196 + 2 * tid;
            const double x0 = rx / a.scale_w, x1 = ry / a.scale_h;               // now / scale
            const double p0 = pl[0] / a.scale_w, p1 = pl[1] / a.scale_h;         // previous / scale
            const double q0 = pd[0] / a.scale_w, q1 = pd[1] / a.scale_h;         // previous_dx / scale
            const double a_d = (two_pi * a.d_cutoff) / (two_pi * a.d_cutoff + 1.0);
            const double dx = sqrt((x0 - p0) * (x0 - p0) + (x1 - p1) * (x1 - p1));
            const double dxp = sqrt(q0 * q0 + q1 * q1);
            const double dx_hat = a_d * dx + (1.0 - a_d) * dxp;
            const double cutoff = a.min_cutoff + a.beta * fabs(dx_hat);
            double al = (two_pi * cutoff) / (two_pi * cutoff + 1.0);
            if (dx < 0.002) al = 0.01;
            const double f0 = (al * x0 + (1.0 - al) * p0) * a.scale_w;
            const double f1 = (al * x1 + (1.0 - al) * p1) * a.scale_h;
            ddx = pl[0] - f0; ddy = pl[1] - f1;                                  // previous - filtered (lk.py:45)
            rx = f0; ry = f1;
        }
        a.out_lm[(size_t)oi * 196 + 2 * tid] = rx;
        a.out_lm[(size_t)oi * 196 + 2 * tid + 1] = ry;
        a.out_dx[(size_t)oi * 196 + 2 * tid] = ddx;
        a.out_dx[(size_t)oi * 196 + 2 * tid + 1] = ddy;
        a.scores_out[(size_t)oi * 98 + tid] = a.scores_in[(size_t)slot * 98 + tid];
        s_now[2 * tid] = rx;
        s_now[2 * tid + 1] = ry;
    }
    __syncthreads();
    double hb[4];
    pf_hull_98(s_now, tid, s_red, hb);
    if (tid < 4) a.hull[(size_t)oi * 4 + tid] = hb[tid];
}

// rows of the new landmark set = face slots the crop stage accepted
__global__ void track_count_kernel(const int* crop_params, const int* n_sel, int* n_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int nv = 0;
    const int n = *n_sel;
    for (int k = 0; k < n; ++k) nv += crop_params[(size_t)k * 8] != 0 ? 1 : 0;
    *n_out = nv;
}
