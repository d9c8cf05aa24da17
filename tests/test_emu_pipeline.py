"""Pre/post-processing kernels and the fused pipeline on the CPU SIMT emulator vs the numpy oracle
(bit-exact for the integer / byte work: letterbox pixels, crop boxes, crop pixels, NMS keep lists)."""
import numpy as np
import pytest

from oracle import prepost as pp
from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from tests import helpers
from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows


@pytest.mark.parametrize("hw,out_hw", [((270, 480), (96, 160)), ((300, 200), (96, 160)), ((192, 320), (96, 160)),
                                       ((101, 333), (64, 96))])
def test_letterbox_bit_exact(emu_engine, hw, out_hw):
    frame, _ = make_frame(hw[0], hw[1], 2, seed=3)
    got, info = emu_engine.letterbox(frame, out_hw)
    ref, rinfo = pp.detector_preprocess_u8(frame, out_hw)
    assert np.array_equal(got, ref)
    assert info[0] == np.float32(rinfo[0]) and info[1] == rinfo[1] and info[2] == rinfo[2]


def test_nms_rows_matches_py_nms(emu_engine):
    rng = np.random.default_rng(0)
    for trial in range(3):
        n = 600
        rows = np.zeros((n, 16), np.float32)
        rows[:, 0] = rng.uniform(50, 590, n)
        rows[:, 1] = rng.uniform(50, 330, n)
        rows[:, 2:4] = rng.uniform(10, 120, (n, 2))
        rows[:, 4] = rng.permutation(np.linspace(0.01, 0.99, n)).astype(np.float32)
        rows[:, 5:] = rng.uniform(0, 1, (n, 11))
        if trial == 2:
            rows[5, 2:4] = 0.0   # zero-area box: IoU is 0/0 = NaN against itself -> must be handled like numpy
        info = [np.float32(1.0 / 3.0), 0, 12]
        ref = pp.detector_postprocess(rows, info, 0.3, 0.5)
        got = emu_engine.nms_rows(rows, info[0], info[1], info[2], 0.5, 0.3)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)


def test_crop_faces_bit_exact(emu_engine):
    frame, boxes = make_frame(270, 480, 4, seed=5)
    extra = np.array([[-30.0, -20.0, 60.0, 70.0],       # sticks out of the frame (zero padding visible)
                      [400.0, 200.0, 479.5, 269.0],     # bottom-right corner
                      [100.0, 100.0, 115.0, 140.0],     # too narrow: rejected (w <= 20)
                      [200.0, 50.0, 296.0, 140.0]], np.float32)  # 2*floor(0.7*96)=134 -> generic resize
    allb = np.concatenate([boxes, extra], 0)
    S = 64
    crops, params = emu_engine.crop_faces(frame, allb, S)
    for i, b in enumerate(allb):
        ci = pp.landmark_crop_box(b, frame.shape[0], frame.shape[1])
        assert bool(params[i, 0]) == ci.valid, i
        if not ci.valid:
            continue
        assert (params[i, 1], params[i, 2], params[i, 3], params[i, 6], params[i, 7]) == \
               (ci.add, ci.x0, ci.y0, ci.w_crop, ci.h_crop), i
        ref = pp.landmark_crop(frame, ci, (S, S))
        assert np.array_equal(crops[i], ref), i


def test_crop_exact_2x_uses_box_average(emu_engine):
    frame, _ = make_frame(270, 480, 1, seed=9)
    w = 92.0   # 2*floor(0.7*92) = 128 = 2*64
    box = np.array([[150.0, 80.0, 150.0 + w, 180.0]], np.float32)
    crops, params = emu_engine.crop_faces(frame, box, 64)
    ci = pp.landmark_crop_box(box[0], 270, 480)
    assert ci.w_crop == 128 and ci.h_crop == 128
    assert np.array_equal(crops[0], pp.landmark_crop(frame, ci, (64, 64)))


def test_landmarks_stage_matches_reference_chain(emu_engine, student_weights):
    """pf_landmarks == FaceLandmark.__call__: crop -> /255 -> net -> back-projection."""
    S = 64
    frame, boxes = make_frame(270, 480, 3, seed=11)
    blob, _ = build_student_program(student_weights, S, "f32")
    emu_engine.load_program(0, blob, 4)
    bad = np.array([[10.0, 10.0, 25.0, 60.0]], np.float32)
    kps, scores, valid = emu_engine.landmarks(frame, np.concatenate([boxes, bad], 0))
    assert valid.tolist() == [True, True, True, False]
    for i, b in enumerate(boxes):
        ci = pp.landmark_crop_box(b, 270, 480)
        crop = pp.landmark_crop(frame, ci, (S, S))
        oloc, oscore, taps = helpers.oracle_student(student_weights, crop[None])
        ref = pp.landmark_backproject(oloc[0], ci)
        safe = helpers.heat_margins(taps)[0] > 1e-3
        assert np.abs(kps[i] - ref)[safe].max() < 1e-3 * max(ci.w_crop, ci.h_crop)  # north-star 1e-3 (normalised)
        assert np.abs(scores[i] - oscore[0])[safe].max() < 2e-3


def test_run_frames_planted_end_to_end(emu_engine, student_weights):
    """FaceAna.run()+reset() semantics on planted detections: NMS must keep exactly the planted
    boxes, top-k by area, then landmarks for each."""
    S, F, top_k = 64, 2, 3
    blob, _ = build_student_program(student_weights, S, "f32")
    emu_engine.load_program(0, blob, F * top_k)
    frames, rows_all, refs = [], [], []
    for f in range(F):
        frame, boxes = make_frame(270, 480, 4, seed=20 + f, face_w=300 + 40 * f, face_h=400)
        boxes[:, 2] += np.arange(4) * 6     # distinct areas so top-k is unambiguous
        rows = plant_rows(boxes, (270, 480), n_rows=1260, input_hw=(384, 640), per_box=6, seed=f)
        frames.append(frame)
        rows_all.append(rows)
        _, info = pp.detector_preprocess_u8(frame, (384, 640))
        kept = pp.detector_postprocess(rows, [np.float32(info[0]), info[1], info[2]], 0.3, 0.5)
        assert kept.shape[0] == 4
        refs.append(pp.sort_and_filter(kept, 100.0, top_k))
    counts, boxes_out, kps, scores = emu_engine.run_frames(np.stack(frames), 0.5, 0.3, 100.0, top_k,
                                                            planted_rows=np.stack(rows_all))
    assert counts.tolist() == [top_k, top_k]
    for f in range(F):
        assert np.array_equal(boxes_out[f], refs[f][:, :4])
        for k in range(top_k):
            ci = pp.landmark_crop_box(refs[f][k], 270, 480)
            crop = pp.landmark_crop(frames[f], ci, (S, S))
            oloc, oscore, taps = helpers.oracle_student(student_weights, crop[None])
            ref = pp.landmark_backproject(oloc[0], ci)
            safe = helpers.heat_margins(taps)[0] > 1e-3
            assert np.abs(kps[f, k] - ref)[safe].max() < 1e-3 * max(ci.w_crop, ci.h_crop)  # north-star 1e-3 (normalised)


def test_run_frames_empty_and_ragged(emu_engine, student_weights):
    """Edge cases of the batched pipeline: a frame with no detection above the threshold, a frame with
    fewer faces than top_k, a frame whose only candidate is too small (area <= min_face)."""
    S, top_k = 64, 3
    blob, _ = build_student_program(student_weights, S, "f32")
    emu_engine.load_program(0, blob, 3 * top_k)
    frames, rows_all = [], []
    frame, boxes = make_frame(270, 480, 2, seed=31, face_w=300, face_h=400)
    for case in range(3):
        frames.append(frame)
        rows = plant_rows(boxes, (270, 480), n_rows=600, input_hw=(384, 640), per_box=4, seed=case)
        if case == 0:
            rows[:, 4] = np.minimum(rows[:, 4], 0.3)            # nothing above the score threshold
        if case == 2:
            rows[:, 4] = np.minimum(rows[:, 4], 0.3)
            rows[7, :5] = (100.0, 100.0, 12.0, 14.0, 0.9)        # one tiny box: dropped by min_face
        rows_all.append(rows)
    counts, bout, kps, scores = emu_engine.run_frames(np.stack(frames), 0.5, 0.3, 1600.0, top_k,
                                                      planted_rows=np.stack(rows_all))
    assert counts.tolist() == [0, 2, 0]
    _, info = pp.detector_preprocess_u8(frame, (384, 640))
    kept = pp.detector_postprocess(rows_all[1], [np.float32(info[0]), info[1], info[2]], 0.3, 0.5)
    ref = pp.sort_and_filter(kept, 1600.0, top_k)
    assert np.array_equal(bout[1, :2], ref[:, :4])
    assert np.isfinite(kps[1, :2]).all()


def test_landmarks_no_boxes_and_detect_rejects_bad_args(emu_engine, student_weights):
    blob, _ = build_student_program(student_weights, 64, "f32")
    emu_engine.load_program(0, blob, 2)
    frame, _ = make_frame(120, 160, 1, seed=2)
    kps, scores, valid = emu_engine.landmarks(frame, np.zeros((0, 4), np.float32))
    assert kps.shape == (0, 98, 2) and valid.shape == (0,)
    from peppa_pig_face_landmark_amd._native import PeppaHipError
    with pytest.raises(PeppaHipError):                         # more faces than the program was sized for
        emu_engine.landmarks(frame, np.tile(np.array([[10, 10, 90, 100]], np.float32), (3, 1)))
    with pytest.raises(PeppaHipError):                         # detector program not loaded
        emu_engine.detect(frame, 0.5, 0.3)


def test_resident_frame_and_diff_gate(emu_engine, student_weights):
    """pf_set_frame: exact |prev - cur| sum (FaceAna.diff_frames, facer.py:111-115) and resident-frame reuse."""
    rng = np.random.default_rng(0)
    f0 = rng.integers(0, 256, (123, 211, 3), dtype=np.uint8)
    f1 = np.clip(f0.astype(np.int16) + rng.integers(-9, 10, f0.shape), 0, 255).astype(np.uint8)
    assert emu_engine.set_frame(f0) is None
    d = emu_engine.set_frame(f1)
    ref = np.abs(f0.astype(np.int64) - f1.astype(np.int64)).sum() / 123 / 211 / 3.0
    assert d == ref
    assert emu_engine.set_frame(f1) == 0.0
    assert emu_engine.set_frame(rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)) is None   # shape change
    emu_engine.forget_frames()
    assert emu_engine.set_frame(f0) is None
    # resident frame feeds the landmark stage exactly like a host frame
    blob, _ = build_student_program(student_weights, 64, "f32")
    emu_engine.load_program(0, blob, 2)
    frame, boxes = make_frame(270, 480, 2, seed=4, face_w=300, face_h=400)
    a = emu_engine.landmarks(frame, boxes)
    emu_engine.set_frame(frame)
    b = emu_engine.landmarks(None, boxes)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_pinned_host_buffer_roundtrip(emu_engine):
    """pf_host_alloc / pf_host_free (frame ingest seam): the buffer is writable, readable and owned by the engine."""
    a = emu_engine.pinned_empty((3, 5, 7, 3), np.uint8)
    a[...] = np.arange(a.size, dtype=np.uint32).reshape(a.shape) % 251
    assert a.shape == (3, 5, 7, 3) and int(a[2, 4, 6, 2]) == (a.size - 1) % 251
    b = emu_engine.pinned_empty((16,), np.float32)
    b[:] = 1.5
    assert float(b.sum()) == 24.0


def _f64_boxes(n=300, seed=9):
    """float64 box rows as FaceAna.track_box holds them on tracked frames, including rows whose float32 rounding crosses
    a floor-division boundary (the case ADVICE r1 pointed at: the crop would shift by one pixel)."""
    rng = np.random.default_rng(seed)
    x1, y1 = rng.uniform(0, 380, n), rng.uniform(0, 180, n)
    bw, bh = rng.uniform(15, 220, n), rng.uniform(15, 220, n)
    b = np.stack([x1, y1, x1 + bw, y1 + bh], 1)
    # (x1 + x2 + 2 add) just below an even integer: float32 rounding of the row pushes it over
    k = np.arange(0, n, 7)
    b[k, 0] = np.floor(b[k, 0])
    b[k, 2] = np.floor(b[k, 2]) + 1.0 - 1e-9
    return b


def check_crop_faces_f64(eng):
    frame, _ = make_frame(270, 480, 2, seed=6)
    boxes = _f64_boxes()
    S = 64
    crops, params = eng.crop_faces(frame, boxes, S)
    differs_from_f32 = 0
    for i, b in enumerate(boxes):
        ci = pp.landmark_crop_box(b, 270, 480)                          # float64 row -> float64 arithmetic
        assert bool(params[i, 0]) == ci.valid, i
        if not ci.valid or ci.x0 < 0 or ci.y0 < 0:
            continue
        assert (params[i, 1], params[i, 2], params[i, 3], params[i, 6], params[i, 7]) == (ci.add, ci.x0, ci.y0, ci.w_crop, ci.h_crop), i
        assert np.array_equal(crops[i], pp.landmark_crop(frame, ci, (S, S))), i
        c32 = pp.landmark_crop_box(b.astype(np.float32), 270, 480)
        differs_from_f32 += (c32.x0, c32.y0, c32.w_crop, c32.add) != (ci.x0, ci.y0, ci.w_crop, ci.add)
    assert differs_from_f32 > 0, "the test boxes never distinguish float64 from float32 box arithmetic"


def test_crop_faces_float64_rows_bit_exact(emu_engine):
    check_crop_faces_f64(emu_engine)


def test_crop_very_large_face_takes_the_per_pixel_path(emu_engine):
    """A face whose source rows do not fit the tiled kernel's LDS budget (decided per workgroup on the device) and a
    small one in the same call: both bit-exact."""
    frame, _ = make_frame(540, 960, 1, seed=8)
    boxes = np.array([[100.0, 40.0, 820.0, 500.0], [300.0, 200.0, 380.0, 290.0]], np.float32)
    crops, params = emu_engine.crop_faces(frame, boxes, 64)
    for i, b in enumerate(boxes):
        ci = pp.landmark_crop_box(b, 540, 960)
        assert ci.valid and bool(params[i, 0])
        assert np.array_equal(crops[i], pp.landmark_crop(frame, ci, (64, 64))), i
