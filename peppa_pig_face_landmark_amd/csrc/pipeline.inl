// Pipeline-level entry points of the C ABI (included at the end of engine.cpp):
//   pf_detect          == FaceDetector.__call__       face_detector.py:23-42
//   pf_landmarks       == FaceLandmark.__call__       face_landmark.py:33-64
//   pf_run_frames*     == FaceAna.run()+reset()       facer.py:52-85 / demo.py:83-86, batched over frames
// Everything between the frame bytes and the result arrays stays on the device.
#include <math.h>

namespace {

const int kMaxKeep = 1024;   // rows kept per frame after NMS (s_keep capacity of nms_kernel)
const int kNumPoints = 98;

template <typename T>
int ensure_dev(pf_handle* h, T*& ptr, size_t& have_bytes, size_t need_bytes) {
    if (need_bytes <= have_bytes && ptr) return 0;
    if (h->capturing) PF_FAIL(h, "internal: device scratch would be reallocated inside a graph capture");
    if (ptr) { (void)hipStreamSynchronize(h->stream); (void)hipFree(ptr); }
    ptr = nullptr;
    have_bytes = 0;
    h->alloc_epoch++;            // captured graphs may hold the old pointer
    PF_HIP(h, hipMalloc((void**)&ptr, need_bytes));
    have_bytes = need_bytes;
    return 0;
}

template <typename T>
int realloc_dev(pf_handle* h, T*& ptr, size_t bytes) {
    if (h->capturing) PF_FAIL(h, "internal: device scratch would be reallocated inside a graph capture");
    if (ptr) { (void)hipStreamSynchronize(h->stream); (void)hipFree(ptr); }
    ptr = nullptr;
    h->alloc_epoch++;
    PF_HIP(h, hipMalloc((void**)&ptr, bytes));
    return 0;
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

int ensure_pipeline(pf_handle* h, int frames, int faces, int top_k, int rows) {
    PipelineScratch& s = h->pipe;
    if (frames > s.cap_frames || top_k > s.cap_topk || rows > s.cap_rows) {
        const int F = std::max(frames, s.cap_frames), K = std::max(top_k, s.cap_topk), R = std::max(rows, s.cap_rows);
        const int cap = next_pow2(std::max(R, 2));
        if (realloc_dev(h, s.d_keep_rows, (size_t)F * kMaxKeep * 16 * sizeof(float))) return 1;
        if (realloc_dev(h, s.d_keep_count, (size_t)F * sizeof(int))) return 1;
        if (realloc_dev(h, s.d_sel_boxes, (size_t)F * K * 4 * sizeof(float))) return 1;
        if (realloc_dev(h, s.d_sel_count, (size_t)F * sizeof(int))) return 1;
        if (realloc_dev(h, s.d_cand_count, (size_t)F * sizeof(int))) return 1;
        if (realloc_dev(h, s.d_nms_keys, (size_t)F * cap * sizeof(unsigned long long))) return 1;
        if (realloc_dev(h, s.d_nms_flags, (size_t)F * cap)) return 1;
        s.cap_frames = F; s.cap_topk = K; s.cap_rows = R;
    }
    if (faces > s.cap_faces) {
        if (realloc_dev(h, s.d_crop_params, (size_t)faces * 8 * sizeof(int))) return 1;
        if (realloc_dev(h, s.d_cropf, (size_t)faces * 5 * sizeof(float))) return 1;
        if (realloc_dev(h, s.d_kps, (size_t)faces * kNumPoints * 2 * sizeof(float))) return 1;
        s.cap_faces = faces;
    }
    return 0;
}

struct LetterboxGeom {
    double scale; int rw, rh, top, left; double scale_x, scale_y;
};

// face_detector.py:51-61 in the same double arithmetic Python uses
LetterboxGeom letterbox_geom(int H, int W, int outH, int outW) {
    LetterboxGeom g;
    const double sh = (double)outH / (double)H, sw = (double)outW / (double)W;
    g.scale = sh < sw ? sh : sw;
    g.rw = (int)((double)W * g.scale);
    g.rh = (int)((double)H * g.scale);
    const double dh = (double)(outH - g.rh) / 2.0, dw = (double)(outW - g.rw) / 2.0;
    g.top = (int)nearbyint(dh - 0.1);   // Python round() == round-half-even
    g.left = (int)nearbyint(dw - 0.1);
    g.scale_x = 1.0 / ((double)g.rw / (double)W);   // OpenCV: scale = 1 / inv_scale
    g.scale_y = 1.0 / ((double)g.rh / (double)H);
    return g;
}

// frames (host or device) -> device pointer
int stage_frames(pf_handle* h, const uint8_t* frames, int mem, size_t bytes, const unsigned char** d_out) {
    if (mem == PF_MEM_DEVICE) { *d_out = frames; return 0; }
    if (mem == PF_MEM_RESIDENT) {
        if (!h->pipe.have_cur || (size_t)h->pipe.cur_h * h->pipe.cur_w * 3 != bytes)
            PF_FAIL(h, "no resident frame of this size (call pf_set_frame first)");
        *d_out = h->pipe.d_cur;
        return 0;
    }
    if (ensure_dev(h, h->pipe.d_frames, h->pipe.frames_bytes, bytes)) return 1;
    PF_HIP(h, hipMemcpyAsync(h->pipe.d_frames, frames, bytes, hipMemcpyHostToDevice, h->stream));
    *d_out = h->pipe.d_frames;
    return 0;
}

// letterbox + detector network; leaves decoded rows in the detector program's output buffer
int run_detector_stage(pf_handle* h, const unsigned char* d_frames, int F, int H, int W, int row_stride,
                       const LetterboxGeom& g) {
    Program& det = h->prog[PF_NET_DETECTOR];
    const int oh = det.hdr.in_h, ow = det.hdr.in_w;
    if (ensure_dev(h, h->pipe.d_letterbox, h->pipe.letterbox_bytes, (size_t)F * oh * ow * 3)) return 1;
    LetterboxArgs la{};
    la.frames = d_frames; la.out = h->pipe.d_letterbox;
    la.F = F; la.H = H; la.W = W; la.row_stride = row_stride; la.outH = oh; la.outW = ow;
    la.rw = g.rw; la.rh = g.rh; la.top = g.top; la.left = g.left;
    la.scale_x = g.scale_x; la.scale_y = g.scale_y; la.pad_value = 114;
    {
        ProfScope ps(h, "letterbox");
        PF_LAUNCH(letterbox_kernel, dim3(pf_div_up(oh * ow, 256), F), dim3(256), h->stream, la);
    }
    return run_program(h, PF_NET_DETECTOR, h->pipe.d_letterbox, PF_INPUT_U8_NHWC, F);
}

int run_nms_stage(pf_handle* h, const float* d_rows, int rows, int F, const LetterboxGeom& g,
                  float score_thres, float iou_thres, float min_face, int top_k, bool select,
                  float* sel_boxes_out = nullptr, int* sel_count_out = nullptr) {
    PipelineScratch& s = h->pipe;
    NmsArgs na{};
    // letterbox geometry travels by value in the kernel arguments, so a captured graph carries its own copy
    na.lb_scale = (float)g.scale; na.lb_left = (float)g.left; na.lb_top = (float)g.top;
    na.rows = d_rows; na.keep_rows = s.d_keep_rows; na.keep_count = s.d_keep_count;
    // (the front lane of a pf_batch selects into buffers the landmark lanes read: batch.inl)
    na.sel_boxes = select ? (sel_boxes_out ? sel_boxes_out : s.d_sel_boxes) : nullptr; na.sel_count = sel_count_out ? sel_count_out : s.d_sel_count;
    na.keys = s.d_nms_keys; na.flags = s.d_nms_flags; na.cand_count = s.d_cand_count;
    na.R = rows; na.cap = next_pow2(std::max(s.cap_rows, 2)); na.max_keep = kMaxKeep; na.top_k = top_k;
    na.score_thres = score_thres; na.iou_thres = iou_thres; na.min_face = min_face;
    ProfScope ps(h, "nms");
    PF_HIP(h, hipMemsetAsync(s.d_cand_count, 0, (size_t)F * sizeof(int), h->stream));
    PF_LAUNCH(nms_compact_kernel, dim3(pf_div_up(rows, 1024), F), dim3(1024), h->stream, na);
    PF_LAUNCH(nms_kernel, dim3(F), dim3(256), h->stream, na);
    return 0;
}

// crop boxes -> uint8 crops (FaceLandmark.preprocess)
int run_crop_stage(pf_handle* h, const unsigned char* d_frames, int H, int W, int row_stride,
                   const float* d_boxes, const int* d_counts, int faces, int per_frame, int S, const double* d_boxes64 = nullptr,
                   const int* d_boxes64_f32 = nullptr) {
    PipelineScratch& s = h->pipe;
    if (ensure_dev(h, s.d_crops, s.crops_bytes, (size_t)faces * S * S * 3)) return 1;
    CropParamArgs ca{};
    ca.boxes = d_boxes; ca.boxes64 = d_boxes64; ca.boxes64_f32 = d_boxes64_f32; ca.counts = d_counts; ca.params = s.d_crop_params; ca.cropf = s.d_cropf;
    ca.n = faces; ca.per_frame = per_frame; ca.H = H; ca.W = W;
    ca.min_face = 20.f;                  // FaceLandmark.min_face, face_landmark.py:26
    ca.width_factor = 1 + 2 * 0.2;       // (1 + 2*extend[0]) with extend = [0.2, 0.3], Skps.yml:14
    {
        ProfScope ps(h, "crop_params");
        PF_LAUNCH(crop_params_kernel, dim3(pf_div_up(faces, 64)), dim3(64), h->stream, ca);
    }
    CropResizeArgs ra{};
    ra.frames = d_frames; ra.params = s.d_crop_params; ra.out = s.d_crops;
    ra.n = faces; ra.per_frame = per_frame; ra.H = H; ra.W = W; ra.row_stride = row_stride; ra.S = S;
    {
        ProfScope ps(h, "crop_resize");
        // tiled kernel (source rows of 8 output rows through LDS, 32-bit global accesses) whenever an output row is a
        // whole number of 32-bit words; it falls back to per-pixel loads per workgroup for faces too large for its LDS
        const bool tiled = S <= 256 && (256 % S) == 0 && (S * 3) % 4 == 0;
        if (tiled) PF_LAUNCH(crop_resize_kernel, dim3(pf_div_up(S, PF_CROP_TY), faces), dim3(256), h->stream, ra);
        else PF_LAUNCH(crop_resize_direct_kernel, dim3(pf_div_up(S * S, 256), faces), dim3(256), h->stream, ra);
    }
    return 0;
}

// crops -> landmark program (with back-projection to frame coordinates)
int run_landmark_stage(pf_handle* h, const unsigned char* d_frames, int H, int W, int row_stride,
                       const float* d_boxes, const int* d_counts, int faces, int per_frame, const double* d_boxes64 = nullptr,
                       const int* d_boxes64_f32 = nullptr) {
    Program& lm = h->prog[PF_NET_LANDMARK];
    PipelineScratch& s = h->pipe;
    if (faces > lm.max_batch) PF_FAIL(h, "%d faces exceed the landmark program's max_batch %d", faces, lm.max_batch);
    if (run_crop_stage(h, d_frames, H, W, row_stride, d_boxes, d_counts, faces, per_frame, lm.hdr.in_h, d_boxes64, d_boxes64_f32)) return 1;
    s.d_crop_for_decode = s.d_cropf;
    s.d_kps_for_decode = s.d_kps;
    const int rc = run_program(h, PF_NET_LANDMARK, s.d_crops, PF_INPUT_U8_NHWC, faces);
    s.d_crop_for_decode = nullptr;
    s.d_kps_for_decode = nullptr;
    return rc;
}

}  // namespace

extern "C" {

int pf_detect(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
              float score_thres, float iou_thres, float* boxes, int max_n, int* n_out) {
    if (!h) return 1;
    Program& det = h->prog[PF_NET_DETECTOR];
    if (!det.loaded) PF_FAIL(h, "detector program not loaded");
    if ((!bgr && mem != PF_MEM_RESIDENT) || !boxes || !n_out || height < 1 || width < 1 || row_stride < width * 3) PF_FAIL(h, "pf_detect: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    const int rows = det.bufs[det.hdr.out_buf0].elems_per_item / 16;
    if (ensure_pipeline(h, 1, 0, 1, rows)) return 1;
    const unsigned char* d_frames = nullptr;
    if (stage_frames(h, bgr, mem, (size_t)height * row_stride, &d_frames)) return 1;
    const LetterboxGeom g = letterbox_geom(height, width, det.hdr.in_h, det.hdr.in_w);
    begin_call(h);
    if (run_detector_stage(h, d_frames, 1, height, width, row_stride, g)) return 1;
    if (run_nms_stage(h, (const float*)det.buf_ptr(det.hdr.out_buf0), rows, 1, g, score_thres, iou_thres, 0.f, 1, false)) return 1;
    int n = 0;
    PF_HIP(h, hipMemcpyAsync(&n, h->pipe.d_keep_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (check_numerics(h)) return 1;
    n = std::min(n, max_n);
    if (n > 0) PF_HIP(h, hipMemcpy(boxes, h->pipe.d_keep_rows, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToHost));
    *n_out = n;
    return 0;
}

static int landmarks_impl(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                          const float* boxes, const double* boxes64, int n, float* kps, float* scores, int* valid);

int pf_landmarks(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                 const float* boxes, int n, float* kps, float* scores, int* valid) {
    if (n > 0 && !boxes) { if (h) h->err = "pf_landmarks: boxes is NULL"; return 1; }
    return landmarks_impl(h, bgr, mem, height, width, row_stride, boxes, nullptr, n, kps, scores, valid);
}

int pf_landmarks_f64(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                     const double* boxes, int n, float* kps, float* scores, int* valid) {
    if (n > 0 && !boxes) { if (h) h->err = "pf_landmarks_f64: boxes is NULL"; return 1; }
    return landmarks_impl(h, bgr, mem, height, width, row_stride, nullptr, boxes, n, kps, scores, valid);
}

static int landmarks_impl(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                          const float* boxes, const double* boxes64, int n, float* kps, float* scores, int* valid) {
    if (!h) return 1;
    Program& lm = h->prog[PF_NET_LANDMARK];
    if (!lm.loaded) PF_FAIL(h, "landmark program not loaded");
    if (n < 0 || (!bgr && mem != PF_MEM_RESIDENT) || height < 1 || width < 1 || row_stride < width * 3) PF_FAIL(h, "pf_landmarks: bad arguments");
    if (n == 0) return 0;
    PF_HIP(h, hipSetDevice(h->device));
    if (ensure_pipeline(h, 1, n, n, 2)) return 1;
    const unsigned char* d_frames = nullptr;
    if (stage_frames(h, bgr, mem, (size_t)height * row_stride, &d_frames)) return 1;
    const double* d_b64 = nullptr;
    if (boxes64) {
        if (ensure_dev(h, h->pipe.d_boxes64, h->pipe.boxes64_bytes, (size_t)n * 4 * sizeof(double))) return 1;
        PF_HIP(h, hipMemcpyAsync(h->pipe.d_boxes64, boxes64, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_b64 = h->pipe.d_boxes64;
    } else {
        PF_HIP(h, hipMemcpyAsync(h->pipe.d_sel_boxes, boxes, (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    begin_call(h);
    if (run_landmark_stage(h, d_frames, height, width, row_stride, h->pipe.d_sel_boxes, nullptr, n, n, d_b64)) return 1;
    std::vector<int> params((size_t)n * 8);
    std::vector<float> hk((size_t)n * kNumPoints * 2), hs((size_t)n * kNumPoints);
    PF_HIP(h, hipMemcpyAsync(params.data(), h->pipe.d_crop_params, params.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipMemcpyAsync(hk.data(), h->pipe.d_kps, hk.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipMemcpyAsync(hs.data(), lm.buf_ptr(lm.hdr.out_buf1), hs.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (check_numerics(h)) return 1;
    for (int i = 0; i < n; ++i) {
        const int ok = params[(size_t)i * 8];
        if (valid) valid[i] = ok;
        if (!ok) continue;
        if (kps) memcpy(kps + (size_t)i * kNumPoints * 2, hk.data() + (size_t)i * kNumPoints * 2, kNumPoints * 2 * sizeof(float));
        if (scores) memcpy(scores + (size_t)i * kNumPoints, hs.data() + (size_t)i * kNumPoints, kNumPoints * sizeof(float));
    }
    return 0;
}

// The two halves of a pf_run_frames call.  FRONT: letterbox + detector network + NMS / top-k of F frames -> selected boxes and
// counts (into the handle's own scratch, or into `sel_boxes_out` / `sel_count_out`).  TAIL: crop + landmark network + result
// copies of F frames whose boxes / counts are `d_boxes` / `d_counts`.  enqueue_run_frames is front + tail on one handle; a
// pf_batch (batch.inl) runs ONE front for all frames of a call on its front engine and one tail per lane.
static int enqueue_front(pf_handle* h, const unsigned char* d_frames, int F, int height, int width,
                         const float* det_rows, bool rows_on_device, int rows, float score_thres, float iou_thres,
                         float min_face, int top_k, float* sel_boxes_out, int* sel_count_out) {
    Program& det = h->prog[PF_NET_DETECTOR];
    if (!det.loaded && !det_rows) PF_FAIL(h, "no detector program and no planted rows");
    const int det_nrows = det.loaded ? det.bufs[det.hdr.out_buf0].elems_per_item / 16 : rows;
    if (det_rows && det.loaded && rows != det_nrows) PF_FAIL(h, "planted rows %d != detector rows %d", rows, det_nrows);
    if (ensure_pipeline(h, F, 0, top_k, det_nrows)) return 1;
    const int row_stride = width * 3;
    const int in_h = det.loaded ? det.hdr.in_h : 384, in_w = det.loaded ? det.hdr.in_w : 640;
    const LetterboxGeom g = letterbox_geom(height, width, in_h, in_w);
    if (det.loaded) {
        if (F > det.max_batch) PF_FAIL(h, "%d frames exceed the detector program's max_batch %d", F, det.max_batch);
        if (run_detector_stage(h, d_frames, F, height, width, row_stride, g)) return 1;
    }
    const float* d_rows = det.loaded ? (const float*)det.buf_ptr(det.hdr.out_buf0) : nullptr;
    if (det_rows) {
        if (rows_on_device) {
            d_rows = det_rows;
        } else {
            const size_t bytes = (size_t)F * rows * 16 * sizeof(float);
            if (ensure_dev(h, h->pipe.d_rows_planted, h->pipe.rows_planted_bytes, bytes)) return 1;
            PF_HIP(h, hipMemcpyAsync(h->pipe.d_rows_planted, det_rows, bytes, hipMemcpyHostToDevice, h->stream));
            d_rows = h->pipe.d_rows_planted;
        }
    }
    return run_nms_stage(h, d_rows, det_nrows, F, g, score_thres, iou_thres, min_face, top_k, true, sel_boxes_out, sel_count_out);
}

static int enqueue_tail(pf_handle* h, const unsigned char* d_frames, int F, int height, int width, const float* d_boxes,
                        const int* d_counts, int top_k, int* counts, float* boxes, float* kps, float* scores, int out_mem) {
    Program& lm = h->prog[PF_NET_LANDMARK];
    if (!lm.loaded) PF_FAIL(h, "landmark program not loaded");
    const int faces = F * top_k;
    if (ensure_pipeline(h, 0, faces, 0, 0)) return 1;
    if (run_landmark_stage(h, d_frames, height, width, width * 3, d_boxes, d_counts, faces, top_k)) return 1;
    const hipMemcpyKind kind = out_mem == PF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (counts) PF_HIP(h, hipMemcpyAsync(counts, d_counts, (size_t)F * sizeof(int), kind, h->stream));
    if (boxes) PF_HIP(h, hipMemcpyAsync(boxes, d_boxes, (size_t)faces * 4 * sizeof(float), kind, h->stream));
    if (kps) PF_HIP(h, hipMemcpyAsync(kps, h->pipe.d_kps, (size_t)faces * kNumPoints * 2 * sizeof(float), kind, h->stream));
    if (scores) PF_HIP(h, hipMemcpyAsync(scores, lm.buf_ptr(lm.hdr.out_buf1), (size_t)faces * kNumPoints * sizeof(float), kind, h->stream));
    return 0;
}

static int enqueue_run_frames(pf_handle* h, const uint8_t* frames, int mem, int n_frames, int height, int width,
                              const float* det_rows, int rows, float score_thres, float iou_thres,
                              float min_face, int top_k,
                              int* counts, float* boxes, float* kps, float* scores, int out_mem) {
    Program& det = h->prog[PF_NET_DETECTOR];
    Program& lm = h->prog[PF_NET_LANDMARK];
    if (!lm.loaded) PF_FAIL(h, "landmark program not loaded");
    if (!det.loaded && !det_rows) PF_FAIL(h, "no detector program and no planted rows");
    if (!frames || n_frames < 1 || height < 1 || width < 1 || top_k < 1) PF_FAIL(h, "pf_run_frames: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    const int F = n_frames;
    const int det_nrows = det.loaded ? det.bufs[det.hdr.out_buf0].elems_per_item / 16 : rows;
    if (ensure_pipeline(h, F, F * top_k, top_k, det_nrows)) return 1;      // one (re)allocation round for both halves
    const unsigned char* d_frames = nullptr;
    const bool rows_on_device = (mem & 0xff) == PF_MEM_DEVICE || (mem & PF_MEM_ROWS_DEVICE) != 0;
    mem &= 0xff;
    if (stage_frames(h, frames, mem, (size_t)F * height * width * 3, &d_frames)) return 1;
    if (enqueue_front(h, d_frames, F, height, width, det_rows, rows_on_device, rows, score_thres, iou_thres, min_face, top_k, nullptr, nullptr)) return 1;
    return enqueue_tail(h, d_frames, F, height, width, h->pipe.d_sel_boxes, h->pipe.d_sel_count, top_k, counts, boxes, kps, scores, out_mem);
}

// hipGraph cache of a handle (PF_OPT_HIP_GRAPH): `enqueue` is launched eagerly the first time `key` is seen (which also performs
// every lazy allocation / constant upload), captured into a hipGraph the second time and replayed from then on.
extern "C++" {
template <typename Enqueue>
static int graphed_call(pf_handle* h, GraphKey key, Enqueue&& enqueue) {
    key.epoch = h->alloc_epoch;
    // any (re)allocation since a graph was captured -- scratch growth for a larger call, a program reload -- may have
    // freed memory the graph's kernel arguments point at: drop every graph captured under an older epoch
    if (!h->graphs.empty() && h->graphs.front().key.epoch != h->alloc_epoch) {
        for (auto& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        h->graphs.clear();
    }
    GraphEntry* e = nullptr;
    for (auto& g : h->graphs)
        if (memcmp(&g.key, &key, sizeof(key)) == 0) { e = &g; break; }
    if (!e) {   // first sighting: run eagerly
        if (h->graphs.size() >= 16) {
            for (auto& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
            h->graphs.clear();
        }
        GraphEntry ne{};
        ne.key = key;
        h->graphs.push_back(ne);
        const int rc = enqueue();
        if (h->alloc_epoch != key.epoch) {   // this eager run (re)allocated scratch: older graphs are stale, this entry is not
            for (auto& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
            h->graphs.clear();
            ne.key.epoch = h->alloc_epoch;
            if (!rc) h->graphs.push_back(ne);
        } else if (rc) {
            h->graphs.pop_back();
        }
        return rc;
    }
    if (!e->exec) {
        PF_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        h->capturing = true;
        const int rc = enqueue();
        h->capturing = false;
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return 1; }
        if (ce != hipSuccess || !graph) PF_FAIL(h, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
        const hipError_t ie = hipGraphInstantiate(&e->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) { e->exec = nullptr; PF_FAIL(h, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
    }
    PF_HIP(h, hipGraphLaunch(e->exec, h->stream));
    return 0;
}
}  // extern "C++"

// With PF_OPT_HIP_GRAPH on and everything device resident, the ~170 launches of one call are captured
// into a hipGraph the second time the same (pointers, shapes, thresholds) key is seen and replayed from
// then on: one graph launch instead of ~170 kernel launches (single-frame latency is launch bound).
int pf_run_frames_planted(pf_handle* h, const uint8_t* frames, int mem, int n_frames, int height, int width,
                          const float* det_rows, int rows, float score_thres, float iou_thres,
                          float min_face, int top_k,
                          int* counts, float* boxes, float* kps, float* scores, int out_mem) {
    if (!h) return 1;
    if (out_mem != PF_MEM_HOST && out_mem != PF_MEM_DEVICE && out_mem != PF_MEM_HOST_PINNED) PF_FAIL(h, "pf_run_frames: bad out_mem %d", out_mem);
    // results into page-locked host memory are plain asynchronous copies on the stream: they capture into the graph too
    begin_call(h);
    const bool graphable = h->use_graphs && !h->profiling && mem == PF_MEM_DEVICE &&
                           (out_mem == PF_MEM_DEVICE || out_mem == PF_MEM_HOST_PINNED);
    if (!graphable) {
        if (enqueue_run_frames(h, frames, mem, n_frames, height, width, det_rows, rows, score_thres, iou_thres, min_face,
                               top_k, counts, boxes, kps, scores, out_mem)) return 1;
        if (out_mem == PF_MEM_HOST) {
            PF_HIP(h, hipStreamSynchronize(h->stream));
            return check_numerics(h);
        }
        return 0;
    }
    GraphKey key{};
    key.p[0] = frames; key.p[1] = det_rows; key.p[2] = counts; key.p[3] = boxes; key.p[4] = kps; key.p[5] = scores;
    key.i[0] = n_frames; key.i[1] = height; key.i[2] = width; key.i[3] = rows; key.i[4] = top_k;
    key.f[0] = score_thres; key.f[1] = iou_thres; key.f[2] = min_face;
    return graphed_call(h, key, [&]() {
        return enqueue_run_frames(h, frames, mem, n_frames, height, width, det_rows, rows, score_thres, iou_thres, min_face,
                                  top_k, counts, boxes, kps, scores, out_mem);
    });
}

int pf_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return 1;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocPortable) == hipSuccess ? 0 : 1;
}

int pf_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? 0 : 1; }

int pf_set_option(pf_handle* h, int option, int value) {
    if (!h) return 1;
    if (option == PF_OPT_HIP_GRAPH) { h->use_graphs = value != 0; return 0; }
    if (option == PF_OPT_RANGE_CHECK) {
        if (value < 0) PF_FAIL(h, "PF_OPT_RANGE_CHECK: value must be >= 0");
        if ((value > 0) != (h->range_every > 0)) h->alloc_epoch++;      // captured graphs carry the slot pointers: recapture
        h->range_every = value;                                        // 0 = off, anything else = every forward
        return 0;
    }
    if (option == PF_OPT_JPEG_ENTROPY) {
        if (value < 0 || value > 2) PF_FAIL(h, "PF_OPT_JPEG_ENTROPY: 0 (automatic), 1 (host) or 2 (device)");
        h->jpeg_entropy = value;
        return 0;
    }
    if (option == PF_OPT_JPEG_SYNC_ROUNDS) {
        if (value < 0) PF_FAIL(h, "PF_OPT_JPEG_SYNC_ROUNDS: value must be >= 0");
        h->jpeg_rounds = value;
        return 0;
    }
    PF_FAIL(h, "unknown option %d", option);
}

int pf_set_frame(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                 unsigned long long* abs_diff_sum, int* has_prev) {
    if (!h) return 1;
    if (!bgr || height < 1 || width < 1 || row_stride != width * 3 || mem == PF_MEM_RESIDENT) PF_FAIL(h, "pf_set_frame: bad arguments (packed BGR rows required)");
    PF_HIP(h, hipSetDevice(h->device));
    PipelineScratch& s = h->pipe;
    const size_t bytes = (size_t)height * row_stride;
    // rotate: the current resident frame becomes the previous one
    std::swap(s.d_cur, s.d_prev);
    std::swap(s.cur_bytes, s.prev_bytes);
    s.prev_h = s.cur_h; s.prev_w = s.cur_w; s.have_prev = s.have_cur;
    if (ensure_dev(h, s.d_cur, s.cur_bytes, bytes)) return 1;
    if (!s.d_diff_sum) PF_HIP(h, hipMalloc((void**)&s.d_diff_sum, sizeof(unsigned long long)));
    PF_HIP(h, hipMemcpyAsync(s.d_cur, bgr, bytes, mem == PF_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
    s.cur_h = height; s.cur_w = width; s.have_cur = true;
    const bool comparable = s.have_prev && s.prev_h == height && s.prev_w == width;
    if (has_prev) *has_prev = comparable ? 1 : 0;
    if (abs_diff_sum) *abs_diff_sum = 0;
    if (comparable && abs_diff_sum) {
        PF_HIP(h, hipMemsetAsync(s.d_diff_sum, 0, sizeof(unsigned long long), h->stream));
        AbsDiffArgs a{};
        a.a = s.d_cur; a.b = s.d_prev; a.sum = s.d_diff_sum; a.n = bytes;
        const unsigned blocks = (unsigned)std::min<size_t>(2048, (bytes / 16 + 255) / 256 + 1);
        {
            ProfScope ps(h, "absdiff_sum");
            PF_LAUNCH(absdiff_sum_kernel, dim3(blocks), dim3(256), h->stream, a);
        }
        PF_HIP(h, hipMemcpyAsync(abs_diff_sum, s.d_diff_sum, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    }
    PF_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pf_forget_frames(pf_handle* h) {
    if (!h) return 1;
    h->pipe.have_cur = false;
    h->pipe.have_prev = false;
    return 0;
}

int pf_letterbox(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                 int out_h, int out_w, uint8_t* out_host, float* info3) {
    if (!h) return 1;
    if (!bgr || !out_host || height < 1 || width < 1 || out_h < 1 || out_w < 1 || row_stride < width * 3) PF_FAIL(h, "pf_letterbox: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    const unsigned char* d_frames = nullptr;
    if (stage_frames(h, bgr, mem, (size_t)height * row_stride, &d_frames)) return 1;
    const LetterboxGeom g = letterbox_geom(height, width, out_h, out_w);
    if (ensure_dev(h, h->pipe.d_letterbox, h->pipe.letterbox_bytes, (size_t)out_h * out_w * 3)) return 1;
    LetterboxArgs la{};
    la.frames = d_frames; la.out = h->pipe.d_letterbox;
    la.F = 1; la.H = height; la.W = width; la.row_stride = row_stride; la.outH = out_h; la.outW = out_w;
    la.rw = g.rw; la.rh = g.rh; la.top = g.top; la.left = g.left;
    la.scale_x = g.scale_x; la.scale_y = g.scale_y; la.pad_value = 114;
    PF_LAUNCH(letterbox_kernel, dim3(pf_div_up(out_h * out_w, 256), 1), dim3(256), h->stream, la);
    PF_HIP(h, hipMemcpyAsync(out_host, h->pipe.d_letterbox, (size_t)out_h * out_w * 3, hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (info3) { info3[0] = (float)g.scale; info3[1] = (float)g.left; info3[2] = (float)g.top; }
    return 0;
}

int pf_resize(pf_handle* h, const uint8_t* img, int mem, int height, int width, int row_stride,
              int out_h, int out_w, uint8_t* out_host) {
    if (!h) return 1;
    if (!img || !out_host || height < 1 || width < 1 || out_h < 1 || out_w < 1 || row_stride < width * 3) PF_FAIL(h, "pf_resize: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    const unsigned char* d_img = nullptr;
    if (stage_frames(h, img, mem, (size_t)height * row_stride, &d_img)) return 1;
    if (ensure_dev(h, h->pipe.d_letterbox, h->pipe.letterbox_bytes, (size_t)out_h * out_w * 3)) return 1;
    LetterboxArgs la{};
    la.frames = d_img; la.out = h->pipe.d_letterbox;
    la.F = 1; la.H = height; la.W = width; la.row_stride = row_stride; la.outH = out_h; la.outW = out_w;
    la.rw = out_w; la.rh = out_h; la.top = 0; la.left = 0;
    la.scale_x = 1.0 / ((double)out_w / (double)width);     // OpenCV: scale = 1 / inv_scale
    la.scale_y = 1.0 / ((double)out_h / (double)height);
    la.pad_value = 0; la.keep_order = 1;
    PF_LAUNCH(letterbox_kernel, dim3(pf_div_up(out_h * out_w, 256), 1), dim3(256), h->stream, la);
    PF_HIP(h, hipMemcpyAsync(out_host, h->pipe.d_letterbox, (size_t)out_h * out_w * 3, hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pf_nms_rows(pf_handle* h, const float* rows_host, int n_rows, float scale, float left, float top,
                float score_thres, float iou_thres, float* kept, int max_n, int* n_out) {
    if (!h) return 1;
    if (!rows_host || n_rows < 1 || !kept || !n_out) PF_FAIL(h, "pf_nms_rows: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    if (ensure_pipeline(h, 1, 0, 1, n_rows)) return 1;
    const size_t bytes = (size_t)n_rows * 16 * sizeof(float);
    if (ensure_dev(h, h->pipe.d_rows_planted, h->pipe.rows_planted_bytes, bytes)) return 1;
    PF_HIP(h, hipMemcpyAsync(h->pipe.d_rows_planted, rows_host, bytes, hipMemcpyHostToDevice, h->stream));
    LetterboxGeom g{};
    g.scale = scale; g.left = (int)left; g.top = (int)top;
    if (run_nms_stage(h, h->pipe.d_rows_planted, n_rows, 1, g, score_thres, iou_thres, 0.f, 1, false)) return 1;
    int n = 0;
    PF_HIP(h, hipMemcpyAsync(&n, h->pipe.d_keep_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    n = std::min(n, max_n);
    if (n > 0) PF_HIP(h, hipMemcpy(kept, h->pipe.d_keep_rows, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToHost));
    *n_out = n;
    return 0;
}

static int crop_faces_impl(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                           const float* boxes, const double* boxes64, int n, int out_size, uint8_t* crops_host, int* params_host);

int pf_crop_faces(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                  const float* boxes, int n, int out_size, uint8_t* crops_host, int* params_host) {
    if (!boxes) { if (h) h->err = "pf_crop_faces: boxes is NULL"; return 1; }
    return crop_faces_impl(h, bgr, mem, height, width, row_stride, boxes, nullptr, n, out_size, crops_host, params_host);
}

int pf_crop_faces_f64(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                      const double* boxes, int n, int out_size, uint8_t* crops_host, int* params_host) {
    if (!boxes) { if (h) h->err = "pf_crop_faces_f64: boxes is NULL"; return 1; }
    return crop_faces_impl(h, bgr, mem, height, width, row_stride, nullptr, boxes, n, out_size, crops_host, params_host);
}

static int crop_faces_impl(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                           const float* boxes, const double* boxes64, int n, int out_size, uint8_t* crops_host, int* params_host) {
    if (!h) return 1;
    if (!bgr || n < 1 || out_size < 1 || height < 1 || width < 1 || row_stride < width * 3) PF_FAIL(h, "pf_crop_faces: bad arguments");
    PF_HIP(h, hipSetDevice(h->device));
    if (ensure_pipeline(h, 1, n, n, 2)) return 1;
    const unsigned char* d_frames = nullptr;
    if (stage_frames(h, bgr, mem, (size_t)height * row_stride, &d_frames)) return 1;
    const double* d_b64 = nullptr;
    if (boxes64) {
        if (ensure_dev(h, h->pipe.d_boxes64, h->pipe.boxes64_bytes, (size_t)n * 4 * sizeof(double))) return 1;
        PF_HIP(h, hipMemcpyAsync(h->pipe.d_boxes64, boxes64, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_b64 = h->pipe.d_boxes64;
    } else {
        PF_HIP(h, hipMemcpyAsync(h->pipe.d_sel_boxes, boxes, (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    if (run_crop_stage(h, d_frames, height, width, row_stride, h->pipe.d_sel_boxes, nullptr, n, n, out_size, d_b64)) return 1;
    if (crops_host) PF_HIP(h, hipMemcpyAsync(crops_host, h->pipe.d_crops, (size_t)n * out_size * out_size * 3, hipMemcpyDeviceToHost, h->stream));
    if (params_host) PF_HIP(h, hipMemcpyAsync(params_host, h->pipe.d_crop_params, (size_t)n * 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pf_run_frames(pf_handle* h, const uint8_t* frames, int mem, int n_frames, int height, int width,
                  float score_thres, float iou_thres, float min_face, int top_k,
                  int* counts, float* boxes, float* kps, float* scores, int out_mem) {
    return pf_run_frames_planted(h, frames, mem, n_frames, height, width, nullptr, 0, score_thres, iou_thres,
                                 min_face, top_k, counts, boxes, kps, scores, out_mem);
}

}  // extern "C"
