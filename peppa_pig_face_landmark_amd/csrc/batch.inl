// Multi-lane batch runner of the C ABI (included at the end of engine.cpp): pf_batch_*.
//
// FaceAna.run() is a per-frame call (facer.py:52-85) and the reference never wrote a batch path (face_landmark.py:119:
// "TODO batched").  A pf_batch owns `lanes` engines -- one HIP stream, one activation arena, one graph cache each -- on ONE device
// and gives every call's frames to the lanes as contiguous slices: the launches of a lane are asynchronous, so the many small
// kernels of one lane's detector overlap the large landmark kernels of the others (profiles/r04_run1_lane_trace_3lanes.md:
// two kernels in flight 40 % of the time with three lanes).  This is the configuration bench.py measures; it used to be
// bench-side Python over three Engine objects.
#include <memory>

struct pf_batch {
    int device = 0;
    std::vector<pf_handle*> lane;
    std::string err;
    // page-locked staging for out_mem == PF_MEM_HOST (pageable user buffers): results of all lanes land here asynchronously,
    // one synchronisation, then plain memcpy -- a pageable device-to-host copy would serialise the lanes at enqueue time
    char* h_stage = nullptr;
    size_t stage_bytes = 0;
};

#define PF_BFAIL(b, ...)                                  \
    do {                                                  \
        char _b[640];                                     \
        snprintf(_b, sizeof(_b), __VA_ARGS__);            \
        (b)->err = _b;                                    \
        return 1;                                         \
    } while (0)

extern "C" {

int pf_batch_create(int device_id, int lanes, pf_batch** out) {
    if (!out) return 1;
    *out = nullptr;
    if (lanes < 1 || lanes > 32) { g_create_error = "pf_batch_create: lanes must be in [1, 32]"; return 1; }
    pf_batch* b = new pf_batch();
    b->device = device_id;
    for (int i = 0; i < lanes; ++i) {
        pf_handle* h = nullptr;
        if (pf_create(device_id, &h)) {
            for (pf_handle* q : b->lane) pf_destroy(q);
            delete b;
            return 1;                 // g_create_error is set by pf_create
        }
        b->lane.push_back(h);
    }
    *out = b;
    return 0;
}

void pf_batch_destroy(pf_batch* b) {
    if (!b) return;
    for (pf_handle* h : b->lane) pf_destroy(h);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    delete b;
}

const char* pf_batch_last_error(pf_batch* b) { return b ? b->err.c_str() : g_create_error.c_str(); }

int pf_batch_lanes(pf_batch* b) { return b ? (int)b->lane.size() : 0; }

pf_handle* pf_batch_lane(pf_batch* b, int lane) {
    return (b && lane >= 0 && lane < (int)b->lane.size()) ? b->lane[lane] : nullptr;
}

int pf_batch_load_program(pf_batch* b, int slot, const void* blob, size_t bytes, int max_batch_per_lane) {
    if (!b) return 1;
    for (size_t i = 0; i < b->lane.size(); ++i)
        if (pf_load_program(b->lane[i], slot, blob, bytes, max_batch_per_lane))
            PF_BFAIL(b, "lane %zu: %s", i, pf_last_error(b->lane[i]));
    return 0;
}

int pf_batch_set_option(pf_batch* b, int option, int value) {
    if (!b) return 1;
    for (size_t i = 0; i < b->lane.size(); ++i)
        if (pf_set_option(b->lane[i], option, value)) PF_BFAIL(b, "lane %zu: %s", i, pf_last_error(b->lane[i]));
    return 0;
}

int pf_batch_sync(pf_batch* b) {
    if (!b) return 1;
    int rc = 0;
    for (size_t i = 0; i < b->lane.size(); ++i)
        if (pf_sync(b->lane[i]) && !rc) {       // every lane is synchronised even if an earlier one failed
            char buf[640];
            snprintf(buf, sizeof(buf), "lane %zu: %s", i, pf_last_error(b->lane[i]));
            b->err = buf;
            rc = 1;
        }
    return rc;
}

int pf_batch_run_frames(pf_batch* b, const uint8_t* frames, int mem, int n_frames, int height, int width,
                        const float* det_rows, int rows, float score_thres, float iou_thres, float min_face, int top_k,
                        int* counts, float* boxes, float* kps, float* scores, int out_mem) {
    if (!b) return 1;
    if (!frames || n_frames < 1 || height < 1 || width < 1 || top_k < 1) PF_BFAIL(b, "pf_batch_run_frames: bad arguments");
    if (out_mem != PF_MEM_HOST && out_mem != PF_MEM_DEVICE && out_mem != PF_MEM_HOST_PINNED) PF_BFAIL(b, "pf_batch_run_frames: bad out_mem %d", out_mem);
    const int L = (int)b->lane.size();
    const int per = (n_frames + L - 1) / L;
    const size_t frame_bytes = (size_t)height * width * 3;
    const size_t n_box = (size_t)top_k * 4, n_kps = (size_t)top_k * kNumPoints * 2, n_sc = (size_t)top_k * kNumPoints;
    // pageable host outputs: all lanes deliver into page-locked staging first
    int* s_counts = counts; float* s_boxes = boxes; float* s_kps = kps; float* s_scores = scores;
    int lane_out = out_mem;
    if (out_mem == PF_MEM_HOST) {
        const size_t need = (size_t)n_frames * (sizeof(int) + (n_box + n_kps + n_sc) * sizeof(float));
        if (need > b->stage_bytes) {
            if (b->h_stage) (void)hipHostFree(b->h_stage);
            b->h_stage = nullptr; b->stage_bytes = 0;
            if (hipSetDevice(b->device) != hipSuccess || hipHostMalloc((void**)&b->h_stage, need, hipHostMallocPortable) != hipSuccess)
                PF_BFAIL(b, "pf_batch_run_frames: cannot allocate %zu bytes of page-locked staging", need);
            b->stage_bytes = need;
        }
        char* q = b->h_stage;
        s_boxes = (float*)q; q += (size_t)n_frames * n_box * sizeof(float);
        s_kps = (float*)q; q += (size_t)n_frames * n_kps * sizeof(float);
        s_scores = (float*)q; q += (size_t)n_frames * n_sc * sizeof(float);
        s_counts = (int*)q;
        lane_out = PF_MEM_HOST_PINNED;
    }
    // every lane's share is checked BEFORE anything is enqueued: a lane that refuses its slice must not leave the lanes in front of
    // it writing into buffers the caller frees when the call fails
    for (int i = 0; i < L; ++i) {
        const int f0 = i * per, nf = std::min(per, n_frames - f0);
        if (nf <= 0) break;
        const Program& det = b->lane[i]->prog[PF_NET_DETECTOR];
        const Program& lm = b->lane[i]->prog[PF_NET_LANDMARK];
        if (!lm.loaded) PF_BFAIL(b, "lane %d: landmark program not loaded", i);
        if (!det.loaded && !det_rows) PF_BFAIL(b, "lane %d: no detector program and no planted rows", i);
        if (nf * top_k > lm.max_batch) PF_BFAIL(b, "lane %d: %d faces exceed the landmark program's max_batch %d", i, nf * top_k, lm.max_batch);
        if (det.loaded && nf > det.max_batch) PF_BFAIL(b, "lane %d: %d frames exceed the detector program's max_batch %d", i, nf, det.max_batch);
    }
    for (int i = 0; i < L; ++i) {
        const int f0 = i * per, nf = std::min(per, n_frames - f0);
        if (nf <= 0) break;
        if (pf_run_frames_planted(b->lane[i], frames + (size_t)f0 * frame_bytes, mem, nf, height, width,
                                  det_rows ? det_rows + (size_t)f0 * rows * 16 : nullptr, rows, score_thres, iou_thres, min_face, top_k,
                                  counts ? s_counts + f0 : nullptr, boxes ? s_boxes + (size_t)f0 * n_box : nullptr,
                                  kps ? s_kps + (size_t)f0 * n_kps : nullptr, scores ? s_scores + (size_t)f0 * n_sc : nullptr, lane_out)) {
            // the lanes already launched keep writing into the caller's buffers (or the staging): drain them before reporting, and
            // keep the FIRST error text
            char first[640];
            snprintf(first, sizeof(first), "lane %d: %s", i, pf_last_error(b->lane[i]));
            for (int j = 0; j < i; ++j) (void)pf_sync(b->lane[j]);
            b->err = first;
            return 1;
        }
    }
    if (out_mem != PF_MEM_HOST) return 0;
    if (pf_batch_sync(b)) return 1;
    if (counts) memcpy(counts, s_counts, (size_t)n_frames * sizeof(int));
    if (boxes) memcpy(boxes, s_boxes, (size_t)n_frames * n_box * sizeof(float));
    if (kps) memcpy(kps, s_kps, (size_t)n_frames * n_kps * sizeof(float));
    if (scores) memcpy(scores, s_scores, (size_t)n_frames * n_sc * sizeof(float));
    return 0;
}

}  // extern "C"
