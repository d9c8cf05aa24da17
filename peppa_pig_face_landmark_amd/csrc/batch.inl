// Multi-lane batch runner of the C ABI (included at the end of engine.cpp): pf_batch_*.
//
// FaceAna.run() is a per-frame call (facer.py:52-85) and the reference never wrote a batch path (face_landmark.py:119:
// "TODO batched").  A pf_batch owns `lanes` engines -- one HIP stream, one activation arena, one graph cache each -- on ONE device
// and gives every call's frames to the lanes as contiguous slices: the launches of a lane are asynchronous, so the many small
// kernels of one lane's detector overlap the large landmark kernels of the others (profiles/r04_run1_lane_trace_3lanes.md:
// two kernels in flight 40 % of the time with three lanes).  This is the configuration bench.py measures; it used to be
// bench-side Python over three Engine objects.
//
// Round 6, the FRONT engine: the detector's ~25 launches per call sit at one to three workgroups per CU on a lane's 32 frames, and a
// 35 us stride-2 unit queued behind another lane's hero conv took 0.66 ms (profiles/r05_run59_lane_trace_3lanes.md).  With device-
// resident frames a call now runs letterbox + detector + NMS / top-k ONCE, on all of its frames, on a fourth engine (`front`, own
// stream and arena): 2.05 ms for 96 frames against 3 x 0.89 ms for three slices of 32 (profiles/r06_run1_*: the launches are
// latency-bound, three times the frames cost 2.3 x), and the lanes run crop + landmark network + result copies of their slices
// behind an event.  The selected boxes are double-buffered by call parity, so the front half of call n + 1 overlaps the lanes'
// tails of call n; the front half of call n + 2 waits for the tails of call n (the readers of the buffer it overwrites).
// Host-resident frames keep the per-lane path (every lane stages and detects its own slice: the copies overlap).
#include <memory>

struct pf_batch {
    int device = 0;
    std::vector<pf_handle*> lane;
    std::string err;
    // page-locked staging for out_mem == PF_MEM_HOST (pageable user buffers): results of all lanes land here asynchronously,
    // one synchronisation, then plain memcpy -- a pageable device-to-host copy would serialise the lanes at enqueue time
    char* h_stage = nullptr;
    size_t stage_bytes = 0;
    // front engine (see the header comment): detector + NMS of a whole call; selected boxes / counts double-buffered by call parity
    pf_handle* front = nullptr;
    int front_mode = 1;                      // PF_OPT_BATCH_FRONT
    float* d_sel_boxes[2] = {nullptr, nullptr};
    int* d_sel_count[2] = {nullptr, nullptr};
    size_t sel_cap = 0;                      // capacity of each buffer pair in faces (frames x top_k) ...
    size_t sel_cap_frames = 0;               // ... and frames
    unsigned long long front_calls = 0;
    hipEvent_t ev_front[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_lane[2];      // lane i's tail of the last call of that parity
    bool ev_lane_live[2] = {false, false};
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
    bool ok = pf_create(device_id, &b->front) == 0 && hipSetDevice(device_id) == hipSuccess;
    for (int par = 0; ok && par < 2; ++par) {
        ok = hipEventCreateWithFlags(&b->ev_front[par], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < lanes; ++i) {
            hipEvent_t e = nullptr;
            ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            if (ok) b->ev_lane[par].push_back(e);
        }
    }
    if (!ok) {
        if (g_create_error.empty()) g_create_error = "pf_batch_create: cannot create the front engine's events";
        pf_batch_destroy(b);
        return 1;
    }
    *out = b;
    return 0;
}

void pf_batch_destroy(pf_batch* b) {
    if (!b) return;
    for (pf_handle* h : b->lane) (void)pf_sync(h);
    if (b->front) (void)pf_sync(b->front);
    for (int par = 0; par < 2; ++par) {
        if (b->ev_front[par]) (void)hipEventDestroy(b->ev_front[par]);
        for (hipEvent_t e : b->ev_lane[par]) (void)hipEventDestroy(e);
        if (b->d_sel_boxes[par]) (void)hipFree(b->d_sel_boxes[par]);
        if (b->d_sel_count[par]) (void)hipFree(b->d_sel_count[par]);
    }
    for (pf_handle* h : b->lane) pf_destroy(h);
    if (b->front) pf_destroy(b->front);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    delete b;
}

pf_handle* pf_batch_front(pf_batch* b) { return b ? b->front : nullptr; }

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
    // the front engine runs the detector on ALL frames of a call (lanes x the per-lane bound); the lanes keep their copy for the
    // per-lane path (host-resident frames, PF_OPT_BATCH_FRONT = 0, pf_batch_lane() users)
    if (slot == PF_NET_DETECTOR && pf_load_program(b->front, slot, blob, bytes, max_batch_per_lane * (int)b->lane.size()))
        PF_BFAIL(b, "front engine: %s", pf_last_error(b->front));
    return 0;
}

int pf_batch_set_option(pf_batch* b, int option, int value) {
    if (!b) return 1;
    if (option == PF_OPT_BATCH_FRONT) { b->front_mode = value ? 1 : 0; return 0; }
    for (size_t i = 0; i < b->lane.size(); ++i)
        if (pf_set_option(b->lane[i], option, value)) PF_BFAIL(b, "lane %zu: %s", i, pf_last_error(b->lane[i]));
    if (pf_set_option(b->front, option, value)) PF_BFAIL(b, "front engine: %s", pf_last_error(b->front));
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
    if (b->front && pf_sync(b->front) && !rc) {
        b->err = std::string("front engine: ") + pf_last_error(b->front);
        rc = 1;
    }
    return rc;
}

// One call in FRONT mode (device-resident frames): detector + NMS of all frames on the front engine, then one tail per lane.
static int batch_run_front(pf_batch* b, const uint8_t* frames, int n_frames, int height, int width, const float* det_rows, int rows,
                           float score_thres, float iou_thres, float min_face, int top_k, int* counts, float* boxes, float* kps,
                           float* scores, int lane_out) {
    const int L = (int)b->lane.size();
    const int per = (n_frames + L - 1) / L;
    const size_t frame_bytes = (size_t)height * width * 3;
    const size_t n_box = (size_t)top_k * 4, n_kps = (size_t)top_k * kNumPoints * 2, n_sc = (size_t)top_k * kNumPoints;
    pf_handle* fr = b->front;
    if (hipSetDevice(b->device) != hipSuccess) PF_BFAIL(b, "pf_batch_run_frames: hipSetDevice(%d) failed", b->device);
    const size_t faces = (size_t)n_frames * top_k;
    if (faces > b->sel_cap || (size_t)n_frames > b->sel_cap_frames) {      // (re)allocation: nothing may be in flight on the old buffers
        if (pf_batch_sync(b)) return 1;
        const size_t cf = std::max(faces, b->sel_cap), cn = std::max((size_t)n_frames, b->sel_cap_frames);
        for (int par = 0; par < 2; ++par) {
            if (b->d_sel_boxes[par]) (void)hipFree(b->d_sel_boxes[par]);
            if (b->d_sel_count[par]) (void)hipFree(b->d_sel_count[par]);
            b->d_sel_boxes[par] = nullptr; b->d_sel_count[par] = nullptr;
            if (hipMalloc((void**)&b->d_sel_boxes[par], cf * 4 * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&b->d_sel_count[par], cn * sizeof(int)) != hipSuccess) {
                b->sel_cap = b->sel_cap_frames = 0;
                PF_BFAIL(b, "pf_batch_run_frames: cannot allocate the selected-box buffers");
            }
        }
        b->sel_cap = cf; b->sel_cap_frames = cn;
        // graphs captured over the old pointers carry them in their keys and are never matched again
    }
    const int par = (int)(b->front_calls++ & 1);
    float* sel_boxes = b->d_sel_boxes[par];
    int* sel_count = b->d_sel_count[par];
    // the front half overwrites buffers the tails of the call before last read
    if (b->ev_lane_live[par])
        for (int i = 0; i < L; ++i)
            if (hipStreamWaitEvent(fr->stream, b->ev_lane[par][i], 0) != hipSuccess) PF_BFAIL(b, "pf_batch_run_frames: hipStreamWaitEvent failed");
    begin_call(fr);
    {
        auto enq = [&]() {
            return enqueue_front(fr, frames, n_frames, height, width, det_rows, true, rows, score_thres, iou_thres, min_face, top_k, sel_boxes, sel_count);
        };
        int rc;
        if (fr->use_graphs && !fr->profiling) {
            GraphKey key{};
            key.p[0] = frames; key.p[1] = det_rows; key.p[2] = sel_boxes; key.p[3] = sel_count;
            key.i[0] = n_frames; key.i[1] = height; key.i[2] = width; key.i[3] = rows; key.i[4] = top_k; key.i[5] = 1;
            key.f[0] = score_thres; key.f[1] = iou_thres; key.f[2] = min_face;
            rc = graphed_call(fr, key, enq);
        } else {
            rc = enq();
        }
        if (rc) {
            b->err = std::string("front engine: ") + pf_last_error(fr);
            (void)pf_sync(fr);
            return 1;
        }
    }
    if (hipEventRecord(b->ev_front[par], fr->stream) != hipSuccess) PF_BFAIL(b, "pf_batch_run_frames: hipEventRecord failed");
    for (int i = 0; i < L; ++i) {
        const int f0 = i * per, nf = std::min(per, n_frames - f0);
        if (nf <= 0) break;
        pf_handle* h = b->lane[i];
        int rc = hipStreamWaitEvent(h->stream, b->ev_front[par], 0) != hipSuccess;
        if (rc) h->err = "hipStreamWaitEvent failed";
        begin_call(h);
        if (!rc) {
            const uint8_t* fp = frames + (size_t)f0 * frame_bytes;
            int* oc = counts ? counts + f0 : nullptr;
            float* ob = boxes ? boxes + (size_t)f0 * n_box : nullptr;
            float* ok = kps ? kps + (size_t)f0 * n_kps : nullptr;
            float* os = scores ? scores + (size_t)f0 * n_sc : nullptr;
            auto enq = [&]() {
                return enqueue_tail(h, fp, nf, height, width, sel_boxes + (size_t)f0 * n_box, sel_count + f0, top_k, oc, ob, ok, os, lane_out);
            };
            if (h->use_graphs && !h->profiling) {
                GraphKey key{};
                key.p[0] = fp; key.p[1] = sel_boxes + (size_t)f0 * n_box; key.p[2] = oc; key.p[3] = ob; key.p[4] = ok; key.p[5] = os;
                key.i[0] = nf; key.i[1] = height; key.i[2] = width; key.i[3] = lane_out; key.i[4] = top_k; key.i[5] = 2;
                rc = graphed_call(h, key, enq);
            } else {
                rc = enq();
            }
        }
        if (!rc && hipEventRecord(b->ev_lane[par][i], h->stream) != hipSuccess) { h->err = "hipEventRecord failed"; rc = 1; }
        if (rc) {
            // what is already queued -- the front half, the lanes in front, this lane's partial work -- keeps writing into the
            // caller's buffers: drain before reporting, keep the FIRST error text
            char first[640];
            snprintf(first, sizeof(first), "lane %d: %s", i, pf_last_error(h));
            (void)pf_sync(fr);
            for (int j = 0; j <= i; ++j) (void)pf_sync(b->lane[j]);
            b->ev_lane_live[par] = false;
            b->err = first;
            return 1;
        }
    }
    b->ev_lane_live[par] = true;
    return 0;
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
    const Program& fdet = b->front->prog[PF_NET_DETECTOR];
    const bool front = b->front_mode && (mem & 0xff) == PF_MEM_DEVICE && (fdet.loaded ? n_frames <= fdet.max_batch : det_rows != nullptr);
    for (int i = 0; i < L; ++i) {
        const int f0 = i * per, nf = std::min(per, n_frames - f0);
        if (nf <= 0) break;
        const Program& det = b->lane[i]->prog[PF_NET_DETECTOR];
        const Program& lm = b->lane[i]->prog[PF_NET_LANDMARK];
        if (!lm.loaded) PF_BFAIL(b, "lane %d: landmark program not loaded", i);
        if (nf * top_k > lm.max_batch) PF_BFAIL(b, "lane %d: %d faces exceed the landmark program's max_batch %d", i, nf * top_k, lm.max_batch);
        if (front) continue;
        if (!det.loaded && !det_rows) PF_BFAIL(b, "lane %d: no detector program and no planted rows", i);
        if (det.loaded && nf > det.max_batch) PF_BFAIL(b, "lane %d: %d frames exceed the detector program's max_batch %d", i, nf, det.max_batch);
    }
    if (front) {
        if (batch_run_front(b, frames, n_frames, height, width, det_rows, rows, score_thres, iou_thres, min_face, top_k,
                            counts ? s_counts : nullptr, boxes ? s_boxes : nullptr, kps ? s_kps : nullptr, scores ? s_scores : nullptr, lane_out)) return 1;
    } else
    for (int i = 0; i < L; ++i) {
        const int f0 = i * per, nf = std::min(per, n_frames - f0);
        if (nf <= 0) break;
        if (pf_run_frames_planted(b->lane[i], frames + (size_t)f0 * frame_bytes, mem, nf, height, width,
                                  det_rows ? det_rows + (size_t)f0 * rows * 16 : nullptr, rows, score_thres, iou_thres, min_face, top_k,
                                  counts ? s_counts + f0 : nullptr, boxes ? s_boxes + (size_t)f0 * n_box : nullptr,
                                  kps ? s_kps + (size_t)f0 * n_kps : nullptr, scores ? s_scores + (size_t)f0 * n_sc : nullptr, lane_out)) {
            // the lanes already launched -- and the failing lane itself, which may have queued copies into counts / boxes / kps before
            // a later call failed -- keep writing into the caller's buffers (or the staging): drain them before reporting, and keep
            // the FIRST error text (pf_sync overwrites the lane's message)
            char first[640];
            snprintf(first, sizeof(first), "lane %d: %s", i, pf_last_error(b->lane[i]));
            for (int j = 0; j <= i; ++j) (void)pf_sync(b->lane[j]);
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
