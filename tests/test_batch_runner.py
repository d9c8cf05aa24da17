"""The multi-lane batch runner (pf_batch_* of the C ABI, _native.BatchEngine, FrameBatchRunner): the configuration bench.py
measures, as a product feature.  CPU tier: slicing / ragged splits / host staging on the SIMT emulator, bit-identical with one
engine.  GPU tier: EXACTLY the bench shape -- 96 frames of 1080p x 8 planted faces, three lanes, hipGraph replay, results in
page-locked host memory -- on the oracle-calibrated weights, a sample of faces checked against the oracle chain."""
import numpy as np
import pytest

from oracle import prepost as pp
from peppa_pig_face_landmark_amd import _native
from peppa_pig_face_landmark_amd.graph.detector import build_detector_program
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows
from tests import helpers


def _small_inputs(F, seed=40):
    frames, rows_all = [], []
    for f in range(F):
        frame, boxes = make_frame(270, 480, 4, seed=seed + f, face_w=300 + 20 * f, face_h=400)
        boxes[:, 2] += np.arange(4) * 6
        frames.append(frame)
        rows_all.append(plant_rows(boxes, (270, 480), n_rows=1260, input_hw=(384, 640), per_box=6, seed=f))
    return np.stack(frames), np.stack(rows_all)


@pytest.mark.parametrize("F,lanes", [(5, 3), (2, 3), (4, 2)])
def test_batch_equals_single_engine_emu(emu_library, student_weights, F, lanes):
    """Ragged splits (5 frames over 3 lanes = 2 + 2 + 1; 2 frames over 3 lanes = an idle lane), pageable host outputs through the
    page-locked staging: every array bit-identical with ONE engine on the same frames."""
    S, top_k = 64, 3
    blob, _ = build_student_program(student_weights, S, "f32")
    frames, rows = _small_inputs(F)
    one = _native.Engine(0, emu_library)
    one.load_program(0, blob, F * top_k)
    ref = one.run_frames(frames, 0.5, 0.3, 100.0, top_k, planted_rows=rows)
    one.close()
    be = _native.BatchEngine(0, lanes, emu_library)
    assert be.lanes == lanes
    be.load_program(0, blob, ((F + lanes - 1) // lanes) * top_k)
    got = be.run_frames(frames, 0.5, 0.3, 100.0, top_k, planted_rows=rows)
    got2 = be.run_frames(frames, 0.5, 0.3, 100.0, top_k, planted_rows=rows)       # staging reused
    be.close()
    assert ref[0].tolist() == [top_k] * F
    for r, g, g2 in zip(ref, got, got2):
        assert np.array_equal(r, g) and np.array_equal(r, g2)


@pytest.mark.parametrize("F,lanes,with_det", [(5, 3, False), (4, 2, True), (2, 3, False)])
def test_batch_front_engine_equals_per_lane_path_emu(emu_library, student_weights, detector_weights, F, lanes, with_det):
    """Round 6: with device-resident frames a call runs letterbox + detector + NMS ONCE on the front engine and the lanes run crop +
    landmarks of their slices behind an event (PF_OPT_BATCH_FRONT, default on).  On the emulator device memory is host memory, so
    numpy buffers stand in: every array bit-identical with the per-lane path (option 0) and with one engine, over two calls (the
    selected boxes are double-buffered by call parity), ragged splits, with and without a detector program behind the planted rows."""
    S, top_k = 64, 3
    blob, _ = build_student_program(student_weights, S, "f32")
    frames, rows = _small_inputs(F)
    frames, rows = np.ascontiguousarray(frames), np.ascontiguousarray(rows, np.float32)
    det_blob = None
    if with_det:
        # the detector's own rows are computed (and replaced by the planted ones, as in bench.py): the planted rows must have the
        # program's row count, 3 anchors x (12 x 20 + 6 x 10 + 3 x 5) = 945 at 96 x 160.  Boxes planted past row 945 are lost: the
        # single-engine reference below sees the same rows.
        det_blob = build_detector_program(detector_weights, (96, 160), "f32")[0]
        rows = np.ascontiguousarray(rows[:, :945])
    one = _native.Engine(0, emu_library)
    one.load_program(0, blob, F * top_k)
    if det_blob is not None:
        one.load_program(_native.PF_NET_DETECTOR, det_blob, F)       # (the letterbox geometry follows the detector's input size)
    ref = one.run_frames(frames, 0.5, 0.3, 100.0, top_k, planted_rows=rows)
    one.close()
    per = (F + lanes - 1) // lanes

    def run(front_mode):
        be = _native.BatchEngine(0, lanes, emu_library)
        be.set_option(_native.PF_OPT_BATCH_FRONT, front_mode)
        be.load_program(0, blob, per * top_k)
        if det_blob is not None:
            be.load_program(_native.PF_NET_DETECTOR, det_blob, per)
        outs = []
        for rep in range(3):            # parity 0, 1, 0: the third call waits for the events of the first
            counts = np.zeros((F,), np.int32)
            boxes = np.zeros((F, top_k, 4), np.float32)
            kps = np.zeros((F, top_k, 98, 2), np.float32)
            scores = np.zeros((F, top_k, 98), np.float32)
            be.run_frames_device(frames.ctypes.data, F, 270, 480, 0.5, 0.3, 100.0, top_k, d_planted=rows.ctypes.data, rows=rows.shape[1],
                                 d_counts=counts.ctypes.data, d_boxes=boxes.ctypes.data, d_kps=kps.ctypes.data, d_scores=scores.ctypes.data)
            be.sync()
            outs.append((counts, boxes.reshape(F, top_k, 4), kps, scores))
        be.close()
        return outs

    assert int(ref[0].sum()) > 0
    for mode in (1, 0):
        for got in run(mode):
            for r, g in zip(ref, got):
                # rows of a frame beyond its count are undefined (run_frames' contract): compare what is defined
                for f in range(F):
                    n = int(ref[0][f])
                    assert np.array_equal(np.asarray(r[f])[:n] if r.ndim > 1 else r[f], np.asarray(g[f])[:n] if g.ndim > 1 else g[f]), (mode, f)


def test_batch_errors_name_the_lane_emu(emu_library, student_weights):
    be = _native.BatchEngine(0, 2, emu_library)
    frames, rows = _small_inputs(2)
    with pytest.raises(_native.PeppaHipError, match="lane 0"):          # nothing loaded
        be.run_frames(frames, 0.5, 0.3, 100.0, 3, planted_rows=rows)
    blob, _ = build_student_program(student_weights, 64, "f32")
    be.load_program(0, blob, 3)                                         # one frame of 3 faces per lane
    with pytest.raises(_native.PeppaHipError, match="exceed"):
        be.run_frames(np.concatenate([frames, frames]), 0.5, 0.3, 100.0, 3, planted_rows=np.concatenate([rows, rows]))
    with pytest.raises(_native.PeppaHipError):
        _native.BatchEngine(0, 0, emu_library)
    be.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [2, 3])
def test_bench_shape_96_frames_lanes_graph_pinned_matches_oracle(hip_library, student_weights, detector_weights, lanes):
    """bench.py's timed program -- pf_batch_run_frames on 96 x 1080p x 8 planted faces, two lanes (the round-6 default; three = rounds
    4-5) behind the front engine, graph replay, results to page-locked host buffers -- with the ORACLE's weights; boxes bit-exact
    against numpy NMS / top-k for every frame, landmarks of 40 faces (spread over all lanes) within 1e-3 of the crop against the
    oracle chain."""
    F, K, H, W = 96, 8, 1080, 1920
    be = _native.BatchEngine(0, lanes, hip_library)
    be.set_option(_native.PF_OPT_HIP_GRAPH, 1)
    be.load_program(_native.PF_NET_LANDMARK, build_student_program(student_weights, 256, "f32s")[0], F // lanes * K)
    be.load_program(_native.PF_NET_DETECTOR, build_detector_program(detector_weights, (384, 640), "f32s")[0], F // lanes)
    base = [make_frame(H, W, K, seed=7 + i) for i in range(4)]
    rng = np.random.default_rng(3)
    frames = be.pinned_empty((F, H, W, 3), np.uint8)
    rows = np.empty((F, 15120, 16), np.float32)
    for f in range(F):
        nz = rng.integers(-3, 4, (H, W, 3), dtype=np.int16)
        frames[f] = np.clip(base[f % 4][0].astype(np.int16) + nz, 0, 255).astype(np.uint8)
        rows[f] = plant_rows(base[f % 4][1], (H, W), 15120, (384, 640), 24, seed=7 + f % 4)
    import ctypes as C
    import torch
    d_frames = torch.from_numpy(np.asarray(frames)).cuda()
    d_rows = torch.from_numpy(rows).cuda()
    h_counts = be.pinned_empty((F,), np.int32)
    h_boxes = be.pinned_empty((F, K, 4), np.float32)
    h_kps = be.pinned_empty((F, K, 98, 2), np.float32)
    h_scores = be.pinned_empty((F, K, 98), np.float32)
    for rep in range(3):        # eager, capture, replay
        for a in (h_counts, h_boxes, h_kps, h_scores):
            a[...] = 0
        be.run_frames_device(d_frames.data_ptr(), F, H, W, 0.5, 0.3, 1600.0, K, d_planted=d_rows.data_ptr(), rows=15120,
                             d_counts=h_counts.ctypes.data, d_boxes=h_boxes.ctypes.data, d_kps=h_kps.ctypes.data,
                             d_scores=h_scores.ctypes.data, out_mem=_native.PF_MEM_HOST_PINNED)
        be.sync()
    assert h_counts.tolist() == [K] * F
    _, info = pp.detector_preprocess_u8(frames[0], (384, 640))
    info = [np.float32(info[0]), info[1], info[2]]
    for f in range(F):
        kept = pp.detector_postprocess(rows[f], info, 0.3, 0.5)
        assert np.array_equal(h_boxes[f], pp.sort_and_filter(kept, 1600.0, K)[:, :4]), f
    worst, n_safe, n_moved, worst_score = 0.0, 0, 0, 0.0
    for f in list(range(0, F, 12)) + [31, 63]:           # lanes 0, 1 and 2 (frames 0-31 / 32-63 / 64-95)
        for k in range(0, K, 2):
            ci = pp.landmark_crop_box(h_boxes[f, k], H, W)
            crop = pp.landmark_crop(np.asarray(frames[f]), ci, (256, 256))
            oloc, oscore, taps = helpers.oracle_student(student_weights, crop[None])
            ref = pp.landmark_backproject(oloc[0], ci)
            margin = helpers.heat_margins(taps)[0]
            safe = margin > 2e-3
            err = np.abs(h_kps[f, k] - ref).max(1) / max(ci.w_crop, ci.h_crop)
            serr = np.abs(h_scores[f, k] - oscore[0])
            # a landmark may only leave the 1e-3 bound by picking the other of two near-equal heat-map cells: the oracle's
            # top-1 / top-2 margin must then be of the size of the heat-map error itself (logits of range ~ 20)
            moved = err > 1e-3
            assert (margin[moved] < 0.05).all(), (f, k, margin[moved], err[moved])
            n_safe += int(safe.sum())
            n_moved += int((moved & safe).sum())
            worst = max(worst, float(err[safe & ~moved].max()))
            worst_score = max(worst_score, float(serr[~moved].max()))
    assert n_moved <= 0.005 * n_safe + 1, (n_moved, n_safe)           # same flip bound as tests/test_gpu_landmark.py
    assert worst < 1e-3
    assert worst_score < 2e-2                                          # heat-map logits, range ~ 20: 1e-3 relative
    print("bench-shape parity: 40 faces, %d landmarks, %d near-tie flips, worst landmark error %.2e of the crop, worst score error %.2e"
          % (n_safe, n_moved, worst, worst_score))
    del C
    be.close()


@pytest.mark.gpu
def test_frame_batch_runner_facade(hip_library, student_weights, detector_weights):
    """FrameBatchRunner.run(frames) returns per frame what a fresh FaceAna.run(frame) returns (device path, no tracking)."""
    from peppa_pig_face_landmark_amd.core.api.batch_runner import FrameBatchRunner
    from peppa_pig_face_landmark_amd.core.api.facer import FaceAna, get_cfg
    cfg = get_cfg()
    cfg["Skps"]["Detect"]["topk"] = 4
    w = {"detector": detector_weights, "keypoints": student_weights}
    r = FrameBatchRunner(cfg=cfg, weights=w, lanes=2, frames_per_lane=2, library=hip_library)
    frames = [make_frame(1080, 1920, 8, seed=7 + i)[0] for i in range(3)]     # seed 7: the frame test_detect_chain_is_self_consistent finds faces on
    got = r.run(frames)
    r.close()
    fa = FaceAna(cfg=cfg, weights=w, library=hip_library)
    n_faces = 0
    for f in range(3):
        fa.reset()
        want = fa.run(frames[f])
        assert len(got[f]) == len(want)
        for g, w_ in zip(got[f], want):          # same kernels, batch-invariant arithmetic: values, not lengths
            assert np.array_equal(g["kps"], w_["kps"]), (f, float(np.abs(g["kps"] - w_["kps"]).max()))
            assert np.array_equal(g["scores"], w_["scores"]), (f, float(np.abs(g["scores"] - w_["scores"]).max()))
            assert np.array_equal(np.asarray(g["box"], np.float32), np.asarray(w_["box"], np.float32)), (f, g["box"], w_["box"])
            n_faces += 1
    assert n_faces >= 1, "the random-weight detector found no face on any frame: nothing was compared"
    fa.engine.close()


def test_run_jpeg_files_recovers_with_the_host_decoder_emu(emu_library, student_weights):
    """Engine.run_jpeg_files = pf_decode_jpeg_batch + pf_run_frames (demo.py:76 over a batch of files).  A Huffman stream the device's
    sub-sequence decoder cannot synchronise in the rounds it was given (forced here: one round, quality-100 noise) is reported by the
    C ABI at the next synchronising call; the wrapper decodes the batch again on the host path and runs the frames again -- results
    equal to the same pipeline fed with libjpeg's pixels."""
    import io
    from PIL import Image
    S, top_k = 64, 3
    frames, rows = _small_inputs(2)
    rng = np.random.default_rng(5)
    noisy = np.clip(frames.astype(np.int16) + rng.integers(-60, 60, frames.shape), 0, 255).astype(np.uint8)
    files = []
    for f in noisy:
        buf = io.BytesIO()
        Image.fromarray(f[..., ::-1]).save(buf, "JPEG", quality=100, subsampling=2)
        files.append(buf.getvalue())
    decoded = np.stack([np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))[..., ::-1] for d in files])
    eng = _native.Engine(0, emu_library)
    eng.load_program(0, build_student_program(student_weights, S, "f32")[0], 2 * top_k)
    ref = eng.run_frames(decoded, 0.5, 0.3, 100.0, top_k, planted_rows=rows)
    eng.set_option(_native.PF_OPT_JPEG_SYNC_ROUNDS, 1)
    eng.decode_jpeg_batch(files, threads=2)
    with pytest.raises(_native.PeppaHipError, match="did not synchronise"):      # what the bare ABI does
        eng.sync()
    got = eng.run_jpeg_files(files, 0.5, 0.3, 100.0, top_k, threads=2, planted_rows=rows)
    eng.close()
    assert ref[0].tolist() == [top_k, top_k]
    for r, g in zip(ref, got):
        assert np.array_equal(r, g)
