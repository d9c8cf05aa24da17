// pf_track_frame / pf_track_reset: FaceAna.run() / reset() for one video stream with the tracking state on the device
// (included at the end of engine.cpp; kernels in k_track.h).
namespace {

const int kTrackMaxNow = 1024;   // rows NMS can keep per frame

int ensure_track(pf_handle* h, int top_k) {
    TrackState& t = h->track;
    if (t.top_k >= top_k && t.d_track_box) return 0;
    PF_HIP(h, hipStreamSynchronize(h->stream));
    t.release();
    const size_t lm = (size_t)top_k * 196 * sizeof(double);
    PF_HIP(h, hipMalloc((void**)&t.d_track_box, (size_t)kTrackMaxNow * 4 * sizeof(double)));
    PF_HIP(h, hipMalloc((void**)&t.d_judged, (size_t)kTrackMaxNow * 4 * sizeof(double)));
    PF_HIP(h, hipMalloc((void**)&t.d_sel, (size_t)top_k * 4 * sizeof(double)));
    PF_HIP(h, hipMalloc((void**)&t.d_hull, (size_t)top_k * 4 * sizeof(double)));
    PF_HIP(h, hipMalloc((void**)&t.d_scores, (size_t)top_k * 98 * sizeof(float)));
    for (int k = 0; k < 2; ++k) {
        PF_HIP(h, hipMalloc((void**)&t.d_lm[k], lm));
        PF_HIP(h, hipMalloc((void**)&t.d_dx[k], lm));
        PF_HIP(h, hipMalloc((void**)&t.d_n_lm[k], sizeof(int)));
    }
    PF_HIP(h, hipMalloc((void**)&t.d_n_track, sizeof(int)));
    PF_HIP(h, hipMalloc((void**)&t.d_n_judged, sizeof(int)));
    PF_HIP(h, hipMalloc((void**)&t.d_n_sel, sizeof(int)));
    PF_HIP(h, hipMalloc((void**)&t.d_f32, 4 * sizeof(int)));
    PF_HIP(h, hipMemset(t.d_f32, 0, 4 * sizeof(int)));
    t.top_k = top_k;
    t.has_track = t.lm_valid = false;
    t.cur = 0;
    return 0;
}

}  // namespace

extern "C" {

int pf_track_reset(pf_handle* h) {
    if (!h) return 1;
    h->track.has_track = false;
    h->track.lm_valid = false;
    return pf_forget_frames(h);
}

static int track_frame_impl(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                            const float* planted_rows, int planted_n,
                            float score_thres, float nms_iou_thres, float min_face, int top_k,
                            float track_iou_thres, float smooth_box, float diff_thres,
                            int* n_out, double* boxes, double* kps, float* scores, int* detector_ran);

int pf_track_frame(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                   float score_thres, float nms_iou_thres, float min_face, int top_k,
                   float track_iou_thres, float smooth_box, float diff_thres, int reserved,
                   int* n_out, double* boxes, double* kps, float* scores, int* detector_ran) {
    (void)reserved;
    return track_frame_impl(h, bgr, mem, height, width, row_stride, nullptr, 0, score_thres, nms_iou_thres, min_face, top_k,
                            track_iou_thres, smooth_box, diff_thres, n_out, boxes, kps, scores, detector_ran);
}

int pf_track_frame_planted(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                           const float* det_rows, int rows, float score_thres, float nms_iou_thres, float min_face, int top_k,
                           float track_iou_thres, float smooth_box, float diff_thres,
                           int* n_out, double* boxes, double* kps, float* scores, int* detector_ran) {
    if (!det_rows || rows < 1) { if (h) h->err = "pf_track_frame_planted: no rows"; return 1; }
    return track_frame_impl(h, bgr, mem, height, width, row_stride, det_rows, rows, score_thres, nms_iou_thres, min_face, top_k,
                            track_iou_thres, smooth_box, diff_thres, n_out, boxes, kps, scores, detector_ran);
}

static int track_frame_impl(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                            const float* planted_rows, int planted_n,
                            float score_thres, float nms_iou_thres, float min_face, int top_k,
                            float track_iou_thres, float smooth_box, float diff_thres,
                            int* n_out, double* boxes, double* kps, float* scores, int* detector_ran) {
    if (!h) return 1;
    Program& det = h->prog[PF_NET_DETECTOR];
    Program& lm = h->prog[PF_NET_LANDMARK];
    if (!det.loaded || !lm.loaded) PF_FAIL(h, "pf_track_frame: detector and landmark programs must be loaded");
    if (!bgr || !n_out || top_k < 1 || top_k > lm.max_batch) PF_FAIL(h, "pf_track_frame: bad arguments (top_k %d, landmark max_batch %d)", top_k, lm.max_batch);
    PF_HIP(h, hipSetDevice(h->device));
    if (ensure_track(h, top_k)) return 1;
    TrackState& t = h->track;
    // 1. frame upload + frame-difference gate (facer.py:55-63,98-118): one 8-byte read-back decides whether the detector runs
    unsigned long long diff_sum = 0;
    int has_prev = 0;
    if (pf_set_frame(h, bgr, mem, height, width, row_stride, &diff_sum, &has_prev)) return 1;
    const double diff = has_prev ? (double)diff_sum / (double)height / (double)width / 3.0 : 0.0;
    const bool run_det = !has_prev || !t.has_track || diff > (double)diff_thres;
    if (detector_ran) *detector_ran = run_det ? 1 : 0;
    const unsigned char* d_frame = h->pipe.d_cur;
    const int rows = det.bufs[det.hdr.out_buf0].elems_per_item / 16;
    if (ensure_pipeline(h, 1, top_k, top_k, rows)) return 1;
    begin_call(h);
    const double* d_boxes_in = t.d_track_box;      // boxes that enter sort_and_filter
    const int* d_n_in = t.d_n_track;
    const int* d_in_f32 = t.d_f32 + 0;             // ... and their dtype flag (k_track.h): track_box's, unless the detector runs
    if (run_det) {
        const LetterboxGeom g = letterbox_geom(height, width, det.hdr.in_h, det.hdr.in_w);
        if (run_detector_stage(h, d_frame, 1, height, width, row_stride, g)) return 1;
        const float* d_rows = (const float*)det.buf_ptr(det.hdr.out_buf0);
        if (planted_rows) {   // planted-candidate protocol: the network ran, its rows are replaced
            if (planted_n != rows) PF_FAIL(h, "pf_track_frame_planted: %d rows, the detector produces %d", planted_n, rows);
            const size_t bytes = (size_t)rows * 16 * sizeof(float);
            if (ensure_dev(h, h->pipe.d_rows_planted, h->pipe.rows_planted_bytes, bytes)) return 1;
            PF_HIP(h, hipMemcpyAsync(h->pipe.d_rows_planted, planted_rows, bytes, hipMemcpyHostToDevice, h->stream));
            d_rows = h->pipe.d_rows_planted;
        }
        if (run_nms_stage(h, d_rows, rows, 1, g, score_thres, nms_iou_thres, 0.f, 1, false)) return 1;
        JudgeArgs ja{};
        ja.prev = t.d_track_box; ja.n_prev = t.d_n_track; ja.has_prev = t.has_track ? 1 : 0; ja.prev_f32 = t.d_f32 + 0;
        ja.now_f32 = h->pipe.d_keep_rows; ja.now_stride = 16; ja.now_f64 = nullptr; ja.now_f32_flag = nullptr; ja.n_now = h->pipe.d_keep_count;
        ja.out = t.d_judged; ja.n_out = t.d_n_judged; ja.out_f32 = t.d_f32 + 1;
        ja.iou_thres = track_iou_thres; ja.alpha = smooth_box; ja.max_now = kTrackMaxNow;
        PF_LAUNCH(track_judge_kernel, dim3(1), dim3(256), h->stream, ja);
        t.lm_valid = false;                        // trace.previous_landmarks_set = None (facer.py:60)
        d_boxes_in = t.d_judged;
        d_n_in = t.d_n_judged;
        d_in_f32 = t.d_f32 + 1;
    }
    // 2. sort_and_filter -> boxes_return (float64 rows, stay on the device)
    SelectArgs sa{};
    sa.boxes = d_boxes_in; sa.n = d_n_in; sa.out = t.d_sel; sa.n_out = t.d_n_sel; sa.boxes_f32 = d_in_f32; sa.min_face = min_face; sa.top_k = top_k;
    PF_LAUNCH(track_select_kernel, dim3(1), dim3(64), h->stream, sa);
    // 3. landmark stage on the selected boxes (count read on the device: slots >= n_sel are skipped)
    if (run_landmark_stage(h, d_frame, height, width, row_stride, h->pipe.d_sel_boxes, t.d_n_sel, top_k, top_k, t.d_sel, d_in_f32)) return 1;
    // 4. One-Euro smoothing against the previous sets + hull boxes
    const int nxt = t.cur ^ 1;
    PF_LAUNCH(track_count_kernel, dim3(1), dim3(64), h->stream, (const int*)h->pipe.d_crop_params, (const int*)t.d_n_sel, t.d_n_lm[nxt]);
    GroupTrackArgs ga{};
    ga.kps = h->pipe.d_kps; ga.scores_in = (const float*)lm.buf_ptr(lm.hdr.out_buf1); ga.crop_params = h->pipe.d_crop_params;
    ga.n_sel = t.d_n_sel;
    ga.prev_lm = t.d_lm[t.cur]; ga.prev_dx = t.d_dx[t.cur]; ga.n_prev = t.d_n_lm[t.cur]; ga.prev_valid = t.lm_valid ? 1 : 0;
    ga.prev_f32 = t.d_f32 + 2 + t.cur; ga.out_f32 = t.d_f32 + 2 + nxt;
    static const int one = 1;     // the new sets are float32 (the network's landmarks) unless a face is smoothed against the previous ones
    PF_HIP(h, hipMemcpyAsync(t.d_f32 + 2 + nxt, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
    ga.out_lm = t.d_lm[nxt]; ga.out_dx = t.d_dx[nxt]; ga.n_out = t.d_n_lm[nxt];
    ga.hull = t.d_hull; ga.scores_out = t.d_scores;
    ga.iou_thres = track_iou_thres; ga.scale_w = (double)width; ga.scale_h = (double)height;
    ga.min_cutoff = 0.15; ga.beta = 0.8; ga.d_cutoff = 1.0;          // OneEuroFilter defaults, lk.py:100-101
    PF_LAUNCH(track_group_kernel, dim3(top_k), dim3(128), h->stream, ga);
    // 5. track_box = judge_boxs(boxes_return, hull boxes) (facer.py:70-81)
    JudgeArgs jb{};
    jb.prev = t.d_sel; jb.n_prev = t.d_n_sel; jb.has_prev = 1; jb.prev_f32 = d_in_f32;
    jb.now_f32 = nullptr; jb.now_stride = 4; jb.now_f64 = t.d_hull; jb.now_f32_flag = t.d_f32 + 2 + nxt; jb.n_now = t.d_n_lm[nxt];
    jb.out = t.d_track_box; jb.n_out = t.d_n_track; jb.out_f32 = t.d_f32 + 0;
    jb.iou_thres = track_iou_thres; jb.alpha = smooth_box; jb.max_now = top_k;
    PF_LAUNCH(track_judge_kernel, dim3(1), dim3(256), h->stream, jb);
    t.cur = nxt;
    t.lm_valid = true;
    t.has_track = true;
    // 6. results of this frame: the only device->host traffic of the call besides the 8-byte gate
    int n = 0;
    PF_HIP(h, hipMemcpyAsync(&n, t.d_n_track, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PF_HIP(h, hipStreamSynchronize(h->stream));
    if (check_numerics(h)) {
        // The range guard poisoned this frame's outputs with NaN, and steps 4-5 above have already folded them into the
        // stream's state (track boxes, landmark sets, One-Euro filters).  Drop all of it: the caller reloads the network as
        // exact f32 and sends the frame again, and that call must find a stream with no track and no previous frame -- so
        // the detector runs -- instead of a zero frame difference over NaN track boxes.
        const std::string why = h->err;
        (void)pf_track_reset(h);
        h->err = why;
        return 1;
    }
    n = std::min(n, top_k);
    *n_out = n;
    if (n > 0) {
        if (boxes) PF_HIP(h, hipMemcpyAsync(boxes, t.d_track_box, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (kps) PF_HIP(h, hipMemcpyAsync(kps, t.d_lm[t.cur], (size_t)n * 196 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (scores) PF_HIP(h, hipMemcpyAsync(scores, t.d_scores, (size_t)n * 98 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        PF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

}  // extern "C"
