/* peppa_hip.h -- C ABI of the MI355X-native FaceAna hot path (libpeppa_hip.so).
 *
 * Drop-in boundary for the compute the reference delegates to onnxruntime / OpenCV / numpy
 * (paths relative to the reference checkout 610265158/Peppa_Pig_Face_Landmark):
 *
 *   reference seam                                              entry point here
 *   ----------------------------------------------------------  -------------------------------
 *   ONNXEngine.__init__(onnx_f)      onnx_model_base.py:7-14     pf_create + pf_load_program
 *   ONNXEngine.__call__(data) kps    onnx_model_base.py:17-27,   pf_landmark_forward
 *                                    face_landmark.py:48
 *   ONNXEngine.__call__(data) det    face_detector.py:29-31      pf_detector_forward
 *   FaceDetector.__call__(image)     face_detector.py:23-42      pf_detect
 *   FaceLandmark.__call__(img,boxes) face_landmark.py:33-64      pf_landmarks
 *   FaceAna.run per frame (no track) facer.py:52-85, demo.py:83-86  pf_run_frames
 *
 * Conventions: every entry returns 0 on success, non-zero on failure (message via
 * pf_last_error).  Plain pointers and sizes only.  `mem` says where caller buffers live:
 * PF_MEM_HOST (numpy / malloc) or PF_MEM_DEVICE (HBM of the handle's device).  One handle = one
 * device + one HIP stream; calls on a handle are serialised by the caller; handles on different
 * devices are independent.  The library never falls back to the CPU.
 */
#ifndef PEPPA_HIP_H
#define PEPPA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pf_handle pf_handle;

enum { PF_MEM_HOST = 0, PF_MEM_DEVICE = 1, PF_MEM_RESIDENT = 2,     /* 2: the frame stored by pf_set_frame (bgr may be NULL) */
       PF_MEM_HOST_PINNED = 3 };  /* out_mem of pf_run_frames* only: page-locked host result buffers (pf_host_alloc); the call
                                   * returns after enqueueing the device->host copies, results are complete after pf_sync() */
/* pf_run_frames_planted only: OR into `mem` when the frames are in host memory but the planted detector rows
 * (a test / benchmark instrument, 968 KB per frame) already live on the device */
enum { PF_MEM_ROWS_DEVICE = 0x100 };
enum { PF_NET_LANDMARK = 0, PF_NET_DETECTOR = 1, PF_NET_SLOTS = 4 };
enum { PF_INPUT_U8_NHWC = 0, PF_INPUT_F32_NCHW = 1 };
enum { PF_DTYPE_F16 = 0, PF_DTYPE_F32 = 1, PF_DTYPE_F32_SPLIT = 2 };  /* 2: f32 tensors, 3 x f16-MFMA split-precision convs */

/* library / build identification: "peppa-hip <version> gfx950" (or "... simt-emu" for the
 * CPU test build of the same sources, which only tests/ may load) */
const char* pf_version(void);

/* Create an engine bound to HIP device `device_id` (one stream, no networks loaded yet). */
int pf_create(int device_id, pf_handle** out);
void pf_destroy(pf_handle* h);
/* Last error message of the handle (or of pf_create when h == NULL). */
const char* pf_last_error(pf_handle* h);
/* Block until all work queued on the handle's stream has finished. */
int pf_sync(pf_handle* h);

/* Load a packed network program (built by peppa_pig_face_landmark_amd.graph) into `slot`
 * and size its activation arena for up to `max_batch` items.  Replaces
 * rt.InferenceSession(onnx_f) -- onnx_model_base.py:14. */
int pf_load_program(pf_handle* h, int slot, const void* blob, size_t bytes, int max_batch);

/* Landmark regressor forward == session.run of kps_student.onnx (face_landmark.py:48):
 * input  B crops, either uint8 NHWC [B][S][S][3] (channel order as given by the caller) or
 *        float32 NCHW [B][3][S][S] already divided by 255 (face_landmark.py:44-47);
 * output loc_fix [B][196] (x0,y0,...,x97,y97 normalised to the crop, model.py:549-552) and
 *        score [B][98] (raw heat-map maximum, model.py:522). */
int pf_landmark_forward(pf_handle* h, const void* input, int input_kind, int mem, int batch,
                        float* loc_fix, float* score, int out_mem);

/* Detector forward == session.run of yolov5n-0.5.onnx (face_detector.py:29-31):
 * input float32 NCHW [1][3][384][640] RGB/255 or uint8 NHWC letterboxed RGB;
 * output [rows][16] decoded rows (cx,cy,w,h,obj,10 landmark coords,cls), rows = 15120 at 384x640. */
int pf_detector_forward(pf_handle* h, const void* input, int input_kind, int mem, int batch,
                        float* rows_out, int out_mem);

/* Debug / parity tap: copy activation tensor `tensor_id` of the program in `slot` (as left by
 * the last forward of `batch` items) to host float32 NHWC [batch][H][W][C]. */
int pf_read_tensor(pf_handle* h, int slot, int tensor_id, int batch, float* out_host, size_t out_elems);

/* FaceDetector.__call__ (face_detector.py:23-42): BGR uint8 frame -> kept detections
 * [n][16] in frame coordinates (cols 0:4 xyxy, col 4 score), at most max_n rows, in keep order. */
int pf_detect(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
              float score_thres, float iou_thres, float* boxes, int max_n, int* n_out);

/* FaceLandmark.__call__ (face_landmark.py:33-64): for each of n boxes (xyxy, float32 [n][4])
 * crop (zero pad, square 1.4*w box), resize to the network size, run the regressor, map the
 * landmarks back to frame coordinates.  kps [n][98][2], scores [n][98], valid[n] = 0 for boxes
 * the reference rejects (w or h <= 20 px, face_landmark.py:76-77) -- those rows are left untouched. */
int pf_landmarks(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                 const float* boxes, int n, float* kps, float* scores, int* valid);
/* Same for float64 box rows.  On tracked video frames FaceAna hands FaceLandmark a float64 array (track_box comes out of
 * the float64 EMA / One-Euro arithmetic, facer.py:66-81, lk.py:19-56), and then every step of preprocess
 * (face_landmark.py:74-93: widths, `bbox += add`, `// 2`, the store-back, astype(int32)) is float64 arithmetic: rounding
 * the boxes to float32 first can move a floor division across an integer and shift the crop by one pixel. */
int pf_landmarks_f64(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                     const double* boxes, int n, float* kps, float* scores, int* valid);

/* Batched FaceAna.run()+reset() (facer.py:52-85 without tracking, demo.py:83-86) over F frames
 * of identical size resident in `frames` ([F][H][W][3] BGR): detect -> NMS -> drop area <=
 * min_face, keep top_k by area (facer.py:120-142) -> landmarks.  Outputs (host or device per
 * out_mem): counts[F], boxes [F][top_k][4], kps [F][top_k][98][2], scores [F][top_k][98]. */
int pf_run_frames(pf_handle* h, const uint8_t* frames, int mem, int n_frames, int height, int width,
                  float score_thres, float iou_thres, float min_face, int top_k,
                  int* counts, float* boxes, float* kps, float* scores, int out_mem);

/* Same stages as pf_run_frames but detections are supplied by the caller (planted-candidate
 * protocol): det_rows is the decoded detector output [F][rows][16] in letterboxed
 * coordinates; everything downstream (xywh2xyxy, NMS, un-letterbox, top-k, landmarks) runs on
 * the device.  det_rows lives where `frames` lives (host or device, per `mem`).  When a detector program
 * is loaded, letterbox + detector network + decode still run (their cost stays in the call) and only their
 * output is replaced by det_rows.  det_rows == NULL means "use the detector's own rows" (== pf_run_frames).
 * Device-side outputs (out_mem == PF_MEM_DEVICE) are complete after pf_sync(). */
int pf_run_frames_planted(pf_handle* h, const uint8_t* frames, int mem, int n_frames, int height, int width,
                          const float* det_rows, int rows, float score_thres, float iou_thres,
                          float min_face, int top_k,
                          int* counts, float* boxes, float* kps, float* scores, int out_mem);

/* Stage-level entry points (each is one reference function on its own; the parity tests check
 * them bit-for-bit against the numpy/OpenCV restatement):
 *  pf_letterbox   == FaceDetector.preprocess up to the float conversion (face_detector.py:45-63):
 *                    BGR frame -> RGB uint8 [out_h][out_w][3] letterboxed with 114; info = scale,left,top
 *  pf_nms_rows    == xywh2xyxy + py_nms + scale_coords (face_detector.py:31-37,73-136) on decoded
 *                    rows [R][16] (host); kept rows (xyxy in frame coords) in keep order
 *  pf_crop_faces  == FaceLandmark.preprocess (face_landmark.py:66-104): crops uint8 [n][S][S][3] and
 *                    params [n][8] = valid, add, x0, y0, xs, ys, w_crop, h_crop */
int pf_letterbox(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                 int out_h, int out_w, uint8_t* out_host, float* info3);
/* cv2.resize(img, (out_w, out_h)) with the default INTER_LINEAR on a uint8 HxWx3 image (channel order untouched), the same
 * fixed-point arithmetic as the letterbox / crop kernels.  Used by the WFLW evaluation harness (tools/eval_wflw.py), whose
 * reference counterpart resizes an aspect-changing crop (TRAIN/face_landmark/tools/eval_WFLW.py:123). */
int pf_resize(pf_handle* h, const uint8_t* img, int mem, int height, int width, int row_stride,
              int out_h, int out_w, uint8_t* out_host);
int pf_nms_rows(pf_handle* h, const float* rows_host, int n_rows, float scale, float left, float top,
                float score_thres, float iou_thres, float* kept, int max_n, int* n_out);
int pf_crop_faces(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                  const float* boxes, int n, int out_size, uint8_t* crops_host, int* params_host);
int pf_crop_faces_f64(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                      const double* boxes, int n, int out_size, uint8_t* crops_host, int* params_host);   /* float64 rows, see pf_landmarks_f64 */

/* Video mode (FaceAna.run on a stream, facer.py:52-85): upload the frame ONCE, keep it resident for the
 * following pf_detect / pf_landmarks calls (pass mem = PF_MEM_RESIDENT), and evaluate the frame-difference
 * gate of FaceAna.diff_frames (facer.py:98-118) on the device against the previous resident frame:
 * *abs_diff_sum = sum |prev - cur| over all H*W*3 bytes (the caller divides by H*W*3 and compares with 5),
 * *has_prev = 0 when there is no previous frame of the same shape.  row_stride must equal 3*width.
 * pf_forget_frames == FaceAna.reset() for the resident frames (facer.py:200-208). */
int pf_set_frame(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                 unsigned long long* abs_diff_sum, int* has_prev);
int pf_forget_frames(pf_handle* h);

/* Frame ingest: cv2.imread(path) for JPEG files (demo.py:76) without the host-side decode and the
 * upload of a 6 MB frame.  Single-scan baseline files -- what cameras and cv2.imwrite produce -- are entropy-decoded on the
 * DEVICE: the host strips the byte stuffing, the Huffman stream (~0.1 byte per pixel) crosses PCIe and is decoded as
 * self-synchronising 1024-bit sub-sequences (or one thread per restart interval when the file carries restart markers, see
 * below).  Other files (several scans, table ids above 1, tiny images) are Huffman-decoded on the host and their 16-bit
 * coefficient records go up through page-locked memory.  Dequantisation, inverse DCT, chroma upsampling and YCbCr->BGR run as
 * kernels either way, bit-identical with libjpeg(-turbo)'s default decoder (JDCT_ISLOW, fancy upsampling), i.e. with what
 * cv2.imread returns.  A stream whose sub-sequences do not settle within the queued rounds is decoded again on the host.  The frame (packed BGR, 3*width bytes per row) stays in device memory owned by the handle until
 * the decode after the next one; *d_bgr is that pointer -- hand it to pf_set_frame / pf_detect / pf_landmarks / pf_run_frames /
 * pf_track_frame with mem = PF_MEM_DEVICE.  bgr_host (may be NULL): host copy for drawing, height*width*3 bytes (sizes from
 * pf_jpeg_info).  Supported: 8-bit baseline / extended-sequential Huffman JPEG, greyscale or YCbCr 4:4:4 / 4:2:2 / 4:2:0,
 * restart markers; progressive, arithmetic-coded, 12-bit, CMYK / Adobe-RGB files and other sampling grids are refused with an
 * error (decode those with the host library).  subsampling: 0 grey, 444, 422 or 420. */
int pf_jpeg_info(const uint8_t* jpeg, size_t bytes, int* height, int* width, int* components, int* subsampling);
int pf_decode_jpeg(pf_handle* h, const uint8_t* jpeg, size_t bytes, int* height, int* width, const uint8_t** d_bgr,
                   uint8_t* bgr_host);
/* n files of one size and sampling -> [n][height][width][3] in device memory (*d_frames, same ownership), ready for
 * pf_run_frames(mem = PF_MEM_DEVICE): the files' Huffman streams are decoded on `threads` host threads (one file per task), the
 * device stages run once over the whole batch.  Files that carry restart markers (one interleaved scan, table ids 0 / 1) in batches
 * of >= 4096 restart intervals skip the host Huffman loop: one device thread per interval decodes the stream, and only the
 * compressed scan crosses PCIe; files WITHOUT restart markers always take the device's sub-sequence decoder
 * (pf_set_option(PF_OPT_JPEG_ENTROPY, 1 | 2) overrides either choice).  If that decoder does not settle, this asynchronous call cannot decode
 * again by itself: the next synchronising call on the handle fails with "did not synchronise" and the batch has to be resubmitted
 * after pf_set_option(PF_OPT_JPEG_ENTROPY, 1) -- the Python shim's Engine.run_jpeg_files() does that and repeats the pipeline
 * call that consumed the frames (ordinary photographs settle in the first rounds; PF_OPT_JPEG_SYNC_ROUNDS queues fewer
 * rounds than the default 10 -- that can only make the decoder give up sooner, i.e. fall back or report, never return different
 * pixels).  Asynchronous like pf_run_frames: the frames are valid in the order of the
 * handle's stream (pf_run_frames on the same handle just works; pf_sync before another stream reads them).  Two buffer sets
 * alternate, so the pointer of call k stays valid until call k + 2 and the host work of call k + 1 overlaps the pipeline still
 * running on the frames of call k. */
int pf_decode_jpeg_batch(pf_handle* h, int n, const uint8_t* const* jpegs, const size_t* sizes, int threads, int* height,
                         int* width, const uint8_t** d_frames);

/* FaceAna.run(image) / reset() for ONE video stream with the tracking state on the device.  The
 * reference keeps track_box, the previous landmark sets and their displacement on the host and walks the boxes through
 * numpy between the two networks (judge_boxs facer.py:144-189, sort_and_filter :120-142, GroupTrack.calculate + the
 * One-Euro filter core/smoother/lk.py:19-56,117-149, hull boxes facer.py:70-81).  Here all of it is device kernels over
 * device state (float64, as the reference computes it under its pinned numpy): per frame the host reads back 8 bytes (the
 * frame-difference sum that gates the detector, facer.py:98-118) and the results.  One handle = one stream; shard streams,
 * never one stream, across handles / GPUs.  Outputs: *n_out faces (<= top_k); boxes [n][4] = the new track boxes, kps
 * [n][98][2] = the smoothed landmarks (both float64), scores [n][98]; *detector_ran says whether the gate ran the detector.
 * track_iou_thres / smooth_box = Skps.yml Trace.iou_thres / Trace.smooth_box, diff_thres = 5 in the reference. */
int pf_track_frame(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                   float score_thres, float nms_iou_thres, float min_face, int top_k,
                   float track_iou_thres, float smooth_box, float diff_thres, int reserved,
                   int* n_out, double* boxes, double* kps, float* scores, int* detector_ran);
int pf_track_reset(pf_handle* h);
/* Same with the decoded detector rows supplied by the caller whenever the gate runs the detector (planted-candidate
 * protocol of pf_run_frames_planted: the network still runs, its output is replaced).  Test / benchmark instrument. */
int pf_track_frame_planted(pf_handle* h, const uint8_t* bgr, int mem, int height, int width, int row_stride,
                           const float* det_rows, int rows, float score_thres, float nms_iou_thres, float min_face, int top_k,
                           float track_iou_thres, float smooth_box, float diff_thres,
                           int* n_out, double* boxes, double* kps, float* scores, int* detector_ran);

/* Frame ingest (replaces the pageable numpy arrays cv2.imread / VideoCapture.read hand to
 * FaceAna.run, demo.py:13-17,76): page-locked host memory for frame batches (decode straight into it) and for
 * results.  Frames passed with mem = PF_MEM_HOST from such a buffer are copied host->device asynchronously on the
 * handle's stream, so with two handles (two streams) one batch's PCIe transfer overlaps the other batch's kernels;
 * from pageable memory the same call still works but the copy serialises.  Not tied to a handle. */
int pf_host_alloc(size_t bytes, void** out);
int pf_host_free(void* p);

/* Multi-GPU: frames shard across ranks -- frame f is processed by rank f mod R, each rank owning one GPU
 * and one handle -- and nothing on the data path crosses GPUs.  The one exchange is the start-up broadcast of the packed
 * network programs from rank 0 over RCCL / xGMI (the reference has no inference-side parallelism at all,
 * face_landmark.py:40-48; its only collectives are the DDP training all-reduces, net_work.py:30,131-137).
 *  pf_comm_unique_id     rank 0: a fresh 128-byte RCCL unique id (ncclGetUniqueId); the caller hands it to the
 *                        other ranks by whatever side channel launched them (env, file, torch.distributed store)
 *  pf_broadcast_weights  collective over all `world` ranks: creates the communicator on first use
 *                        (ncclCommInitRank on the handle's device), broadcasts rank 0's `blob` (*bytes long; other
 *                        ranks pass a buffer of `capacity` bytes and get the size back in *bytes) HBM-to-HBM with
 *                        ncclBroadcast on the handle's stream, then loads it into `slot` exactly like
 *                        pf_load_program on every rank.  *bcast_ms = device time of the payload broadcast.
 *                        world == 1 is valid (degenerates to pf_load_program through the same code path).
 * librccl is bound at the first call (dlopen); a missing library fails that call, not the engine. */
#define PF_COMM_ID_BYTES 128
int pf_comm_unique_id(void* id_out, size_t id_bytes);
int pf_rccl_version(int* version);
int pf_broadcast_weights(pf_handle* h, const void* rccl_unique_id, int rank, int world, int slot,
                         void* blob, size_t capacity, size_t* bytes, int max_batch, float* bcast_ms);
int pf_comm_destroy(pf_handle* h);

/* ---- multi-lane batch runner ---------------------------------------------------------------------------------------------
 * FaceAna.run() is a per-frame call (Skps/core/api/facer.py:52-85) and the reference never wrote a batch path
 * (face_landmark.py:119).  A pf_batch owns `lanes` engines -- one HIP stream, one activation arena and one graph cache
 * each -- on ONE device and hands every call's frames to them as contiguous slices (lane i: frames [i * ceil(F / lanes), ...)),
 * so the small kernels of one lane's detector overlap the large landmark kernels of the others.  This is the configuration
 * bench.py measures (three lanes of 32 frames).  Semantics and argument meaning of pf_batch_run_frames are those of
 * pf_run_frames_planted (det_rows == NULL: the detector's own rows); outputs cover all n_frames in frame order.  With
 * out_mem == PF_MEM_DEVICE / PF_MEM_HOST_PINNED the call returns after enqueueing (results complete after pf_batch_sync);
 * with PF_MEM_HOST it synchronises.  pf_batch_lane() exposes a lane's handle (owned by the batch) for pf_profile_* and
 * the stage-level entry points; max_batch_per_lane of pf_batch_load_program bounds the frames (detector slot) / faces
 * (landmark slot) of ONE lane's slice; the front engine's detector copy is sized for lanes x that many frames. */
typedef struct pf_batch pf_batch;
int pf_batch_create(int device_id, int lanes, pf_batch** out);
void pf_batch_destroy(pf_batch* b);
const char* pf_batch_last_error(pf_batch* b);
int pf_batch_lanes(pf_batch* b);
pf_handle* pf_batch_lane(pf_batch* b, int lane);
/* the engine that runs the detector + NMS half of a call for all lanes (PF_OPT_BATCH_FRONT; owned by the batch) -- for pf_profile_* */
pf_handle* pf_batch_front(pf_batch* b);
int pf_batch_load_program(pf_batch* b, int slot, const void* blob, size_t bytes, int max_batch_per_lane);
int pf_batch_set_option(pf_batch* b, int option, int value);
int pf_batch_sync(pf_batch* b);
int pf_batch_run_frames(pf_batch* b, const uint8_t* frames, int mem, int n_frames, int height, int width,
                        const float* det_rows, int rows, float score_thres, float iou_thres, float min_face, int top_k,
                        int* counts, float* boxes, float* kps, float* scores, int out_mem);

/* Engine options.  PF_OPT_HIP_GRAPH = 1: pf_run_frames* calls whose buffers all live on the device are captured
 * into a hipGraph per distinct (pointers, shapes, thresholds) and replayed (launch-latency bound small batches). */
enum { PF_OPT_HIP_GRAPH = 1,
       /* Range guard of the f32s (split-precision) programs.  Their convolutions write each f32 activation as f16 hi + f16 lo:
        * |x| >= 65504 overflows, and a tensor whose largest magnitude is below 2^-10 loses its low halves to f16 subnormals.
        * On EVERY forward call (value != 0, the default; 0 switches the guard off) each kernel that splits keeps max |x| of
        * what it splits and a one-workgroup kernel at the end of the forward judges the maxima -- part of the captured graphs
        * too.  A tensor outside [2^-10, 6e4] (or holding a NaN) sets that call's outputs to NaN and makes the next
        * synchronising entry point (pf_sync, any call with host outputs) fail with a message naming the op -- never silent
        * inf or garbage.  The remedy is to load the program with PF_DTYPE_F32 (exact f32 MFMA), which the Python facade does
        * on its own. */
       PF_OPT_RANGE_CHECK = 2,
       /* Where pf_decode_jpeg* decodes the Huffman stream: 0 = automatic (device for files without restart markers and for
        * batches that offer >= 4096 restart intervals, host threads otherwise), 1 = host, 2 = device. */
       PF_OPT_JPEG_ENTROPY = 3,
       /* Synchronisation rounds queued by the self-synchronising sub-sequence decoder (1 .. 10; 0 = all 10).  Fewer rounds save
        * empty launches on ordinary photographs; a stream that needs more is decoded again on the host (synchronous call) or
        * reported at the next synchronisation (pf_decode_jpeg_batch) -- never passed on. */
       PF_OPT_JPEG_SYNC_ROUNDS = 4,
       /* pf_batch_set_option only.  1 (default): a pf_batch_run_frames call on device-resident frames runs letterbox + detector +
        * NMS / top-k ONCE, on all of its frames, on the batch's front engine, and the lanes run crop + landmarks of their slices
        * behind it (the detector's launches are latency-bound: 96 frames cost 2.3 x what 32 do).  0: every lane detects its own
        * slice (the only path for host-resident frames). */
       PF_OPT_BATCH_FRONT = 5 };
int pf_set_option(pf_handle* h, int option, int value);

/* Per-kernel device time of the last call, accumulated with HIP events on the handle's stream
 * when profiling is enabled.  names: '\n'-separated kernel tags; ms: same order. */
int pf_profile_enable(pf_handle* h, int on);
int pf_profile_fetch(pf_handle* h, char* names, size_t names_cap, float* ms, int* counts, int cap, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* PEPPA_HIP_H */
