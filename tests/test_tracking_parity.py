"""Frame-to-frame logic of FaceAna (diff gate, judge_boxs, sort_and_filter, One-Euro landmark smoothing, float64 track
boxes) against the REFERENCE's own facer.py / lk.py executed from source (oracle.ref_import.reference_faceana; only the
two network sessions are oracle callables).  Detections are planted rows on both sides, so the comparison covers exactly
the host/device logic between the networks.  Runs in the CPU tier on the SIMT-emulator build; GPU-marked twin below."""
import numpy as np
import pytest
import torch

from oracle import landmark_net as ln
from oracle import ref_import as ri
from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows

S = 64


class _PlantedDetector:
    """Stands where FaceDetector stands in the facade: planted decoded rows through the engine's NMS (pf_nms_rows)."""

    def __init__(self, engine, rows_for, hw):
        self.engine, self.rows_for, self.hw = engine, rows_for, hw

    def __call__(self, image):
        from oracle import prepost as pp
        s, rw, rh, t, b, l, r = pp.letterbox_geometry(self.hw[0], self.hw[1], (384, 640))
        return self.engine.nms_rows(self.rows_for(), np.float32(s), l, t, 0.5, 0.3)


def _video():
    """5 frames: f0, f0 again (static: detector skipped, float64 track boxes feed the landmark stage), a shifted scene
    (detector runs, IoU-matched boxes are EMA-smoothed), the same again, and a frame with one face fewer."""
    f0, b0 = make_frame(270, 480, 3, seed=11, face_w=330, face_h=430)
    f1 = np.roll(f0, 6, axis=1)
    b1 = b0 + np.float32([6, 0, 6, 0])
    f2, b2 = make_frame(270, 480, 2, seed=12, face_w=330, face_h=430)
    frames = [f0, f0, f1, f1, f2]
    boxes = [b0, b0, b1, b1, b2]
    rows = [plant_rows(b, (270, 480), 15120, (384, 640), 6, seed=3 + i) for i, b in enumerate(boxes)]
    return frames, rows


def _compare(make_facer, student_weights):
    frames, rows = _video()
    W = ln.to_torch(student_weights)
    state = {"i": 0}

    def det_model(x):
        return [rows[state["i"]][None]]

    def lmk_model(x):
        with torch.no_grad():
            loc, score = ln.student_forward(W, torch.from_numpy(np.ascontiguousarray(x)))[:2]
        return loc.numpy(), score.numpy()

    ref = ri.reference_faceana(det_model, lmk_model, top_k=5, min_face=1600, kps_input=(S, S, 3))
    facer = make_facer()
    if facer.device_tracking:
        facer._planted_rows = lambda: rows[state["i"]]
    else:
        facer.face_detector = _PlantedDetector(facer.engine, lambda: rows[state["i"]], (270, 480))
    try:
        for i, fr in enumerate(frames):
            state["i"] = i
            r = ref.run(fr.copy())
            g = facer.run(fr.copy())
            assert len(r) == len(g) and len(r) >= 2, (i, len(r), len(g))
            for a, b in zip(r, g):
                if not facer.device_tracking:     # the host facade follows numpy's promotion; the device state is always float64
                    assert np.asarray(b["box"]).dtype == np.asarray(a["box"]).dtype, i
                # north-star bound: 1e-3 of the crop size (crops here are >= 300 px); synthetic weights put some landmarks --
                # and the hull boxes made from them -- thousands of pixels out, hence relative to the magnitude beyond that
                for key in ("box", "kps"):
                    x, y = np.asarray(a[key], np.float64), np.asarray(b[key], np.float64)
                    assert (np.abs(x - y) / np.maximum(300.0, np.abs(x))).max() < 1e-3, (i, key)
                assert np.abs(a["scores"] - b["scores"]).max() < 5e-3, i
        # (under numpy 1.23 -- the reference's pin -- track_box turns float64 after the first frame; under numpy >= 2 it
        # stays float32.  The facade follows whatever numpy does, as asserted per frame above; the float64 crop path of
        # the engine is pinned by tests/test_emu_pipeline.py::test_crop_faces_float64_rows_bit_exact.)
        assert facer.device_tracking or facer.track_box.dtype == ref.track_box.dtype
    finally:
        facer.engine.close()


def _make_facer(library, student_weights, detector_weights, device_tracking=False):
    from Skps import FaceAna
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    cfg = get_cfg()
    cfg["Skps"]["Engine"]["device_tracking"] = device_tracking
    # the detector net's own rows are replaced by planted ones; the device-tracking path checks the row count, so it runs
    # the real 384 x 640 input there
    cfg["Skps"]["Detect"]["input_shape"] = [384, 640, 3] if device_tracking else [96, 160, 3]
    cfg["Skps"]["Keypoints"]["input_shape"] = [S, S, 3]
    cfg["Skps"]["Engine"]["dtype"] = "f32"
    return FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": student_weights}, library=library)


@pytest.mark.skipif(not ri.available(), reason="reference checkout not present (GPU box)")
def test_faceana_video_matches_reference_source_emulator(emu_library, student_weights, detector_weights):
    _compare(lambda: _make_facer(emu_library, student_weights, detector_weights), student_weights)


@pytest.mark.skipif(not ri.available(), reason="reference checkout not present (GPU box)")
def test_device_tracking_matches_reference_source_emulator(emu_library, student_weights, detector_weights):
    """Same video through pf_track_frame: track boxes, landmark sets and One-Euro state never leave the device."""
    _compare(lambda: _make_facer(emu_library, student_weights, detector_weights, device_tracking=True), student_weights)


@pytest.mark.gpu
@pytest.mark.skipif(not ri.available(), reason="reference checkout not present (GPU box)")
def test_faceana_video_matches_reference_source_gpu(hip_library, student_weights, detector_weights):
    _compare(lambda: _make_facer(hip_library, student_weights, detector_weights), student_weights)


def _device_vs_host(library, student_weights, detector_weights):
    """No reference needed (runs on the GPU box): pf_track_frame against the host-side facade logic on the same engine."""
    frames, rows = _video()
    state = {"i": 0}
    host = _make_facer(library, student_weights, detector_weights, device_tracking=False)
    host.face_detector = _PlantedDetector(host.engine, lambda: rows[state["i"]], (270, 480))
    dev = _make_facer(library, student_weights, detector_weights, device_tracking=True)
    dev._planted_rows = lambda: rows[state["i"]]
    try:
        for i, fr in enumerate(frames):
            state["i"] = i
            r, g = host.run(fr.copy()), dev.run(fr.copy())
            assert len(r) == len(g) and len(r) >= 2, (i, len(r), len(g))
            for a, b in zip(r, g):
                for key in ("box", "kps"):
                    x, y = np.asarray(a[key], np.float64), np.asarray(b[key], np.float64)
                    assert (np.abs(x - y) / np.maximum(300.0, np.abs(x))).max() < 1e-3, (i, key)
                assert np.abs(a["scores"] - b["scores"]).max() < 5e-3, i
        dev.reset()
        host.reset()
        state["i"] = 0
        assert len(dev.run(frames[0].copy())) == len(host.run(frames[0].copy()))      # reset() starts a fresh stream on both
    finally:
        host.engine.close()
        dev.engine.close()


def test_device_tracking_equals_host_facade_emulator(emu_library, student_weights, detector_weights):
    _device_vs_host(emu_library, student_weights, detector_weights)


@pytest.mark.gpu
def test_device_tracking_equals_host_facade_gpu(hip_library, student_weights, detector_weights):
    _device_vs_host(hip_library, student_weights, detector_weights)
