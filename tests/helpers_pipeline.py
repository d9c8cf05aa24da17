"""Pipeline-level smoke check used by __graft_entry__.smoke() (oracle = checker)."""
import numpy as np

from oracle import prepost as pp
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows
from tests import helpers


def smoke_pipeline(eng, student_weights):
    S, top_k = 128, 2
    blob, _ = build_student_program(student_weights, S, "f32")
    eng.load_program(0, blob, top_k)
    frame, boxes = make_frame(540, 960, 2, seed=3)
    rows = plant_rows(boxes, (540, 960), 15120, (384, 640), 8, seed=3)
    counts, bout, kps, scores = eng.run_frames(frame[None], 0.5, 0.3, 1600.0, top_k, planted_rows=rows[None])
    assert counts.tolist() == [2], counts
    _, info = pp.detector_preprocess_u8(frame, (384, 640))
    kept = pp.detector_postprocess(rows, [np.float32(info[0]), info[1], info[2]], 0.3, 0.5)
    ref_boxes = pp.sort_and_filter(kept, 1600.0, top_k)
    assert np.array_equal(bout[0], ref_boxes[:, :4])
    worst = 0.0
    for k in range(top_k):
        ci = pp.landmark_crop_box(ref_boxes[k], 540, 960)
        crop = pp.landmark_crop(frame, ci, (S, S))
        oloc, _, taps = helpers.oracle_student(student_weights, crop[None])
        ref = pp.landmark_backproject(oloc[0], ci)
        safe = helpers.heat_margins(taps)[0] > 2e-3
        worst = max(worst, float(np.abs(kps[0, k] - ref)[safe].max() / max(ci.w_crop, ci.h_crop)))
    assert worst < 1e-3, worst
    print("[smoke] frame -> NMS -> crop -> Student@128 -> landmarks OK: max normalised error %.2e" % worst)
