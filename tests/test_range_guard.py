"""f32s hardening (round-1 verdict): the split-precision convolutions write activations as f16 hi + f16 lo, so data-
dependent activation ranges outside f16's reach must never produce silent inf / garbage.  The engine measures max |x| of
every split op's input on checked calls (PF_OPT_RANGE_CHECK) and fails loudly; the Python facade then reloads the network
with exact-f32 convolutions.  Weights are rescaled here so activations reach ~1e5 and ~1e-6."""
import numpy as np
import pytest

from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd import _native
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from tests import helpers


def _scaled(student_weights, factor):
    """Scale one BatchNorm's output so that everything a split-precision conv reads grows / shrinks by `factor`:
    factor > 1: decoder.aspp.project.1 (input of both DecoderBlocks; ReLU is positively homogeneous and the downstream
    BatchNorms keep their fixed statistics); factor < 1: decoder.upsampler2.conv1.1, whose output is the ONLY input of the
    hero conv (a tensor that mixes tiny and O(1) channels is not an underflow: the tiny part is below f32 resolution of
    the sum either way)."""
    w = dict(student_weights)
    bn = "decoder.aspp.project.1" if factor > 1 else "decoder.upsampler2.conv1.1"
    for k in ("weight", "bias"):
        w[f"{bn}.{k}"] = (np.asarray(w[f"{bn}.{k}"], np.float64) * factor).astype(np.float32)
    return w


def _run_guard_cases(make_engine, student_weights, size):
    crops = sw.smooth_blob_images(2, size, seed=606)
    for factor, side in ((3.0e5, "above"), (1.0e-7, "below")):
        w = _scaled(student_weights, factor)
        eng = make_engine()
        try:
            blob, _ = build_student_program(w, size, "f32s")
            eng.load_program(0, blob, 2)
            with pytest.raises(_native.PeppaHipError, match="activation range check failed.*" + side):
                eng.landmark_forward(crops)
            # the exact-f32 program of the SAME weights is the remedy and matches the oracle
            blob, _ = build_student_program(w, size, "f32")
            eng.load_program(0, blob, 2)
            loc, score = eng.landmark_forward(crops)
            oloc, oscore, taps = helpers.oracle_student(w, crops)
            margins = helpers.heat_margins(taps)
            safe = margins > 2e-3 * max(1.0, float(np.abs(taps["hm"].numpy()).max()))
            # the offsets scale with the activations and cancel (f32 noise either way), so the scaled runs are judged on
            # the heat-map maxima: relative 1e-3
            assert np.isfinite(loc).all() and np.isfinite(score).all()
            if safe.any():
                assert np.abs(score - oscore)[safe].max() < 1e-3 * float(np.abs(taps["hm"].numpy()[:, :98]).max())
        finally:
            eng.close()
    # in-range weights: checked calls pass and change nothing
    eng = make_engine()
    try:
        eng.set_option(_native.PF_OPT_RANGE_CHECK, 1)          # check EVERY call
        blob, _ = build_student_program(student_weights, size, "f32s")
        eng.load_program(0, blob, 2)
        a = eng.landmark_forward(crops)
        eng.set_option(_native.PF_OPT_RANGE_CHECK, 0)
        b = eng.landmark_forward(crops)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.isfinite(a[0]).all()
    finally:
        eng.close()


def test_range_guard_emulator(emu_library, student_weights):
    from peppa_pig_face_landmark_amd._native import Engine
    _run_guard_cases(lambda: Engine(0, emu_library), student_weights, 64)


@pytest.mark.gpu
def test_range_guard_gpu(hip_library, student_weights):
    from peppa_pig_face_landmark_amd._native import Engine
    _run_guard_cases(lambda: Engine(0, hip_library), student_weights, 256)


def _facade_fallback(library, student_weights, size):
    from peppa_pig_face_landmark_amd.core.api.hip_model_base import HIPEngine
    w = _scaled(student_weights, 3.0e5)
    m = HIPEngine(w, "keypoints", [size, size, 3], dtype="f32s", max_batch=2, library=library)
    try:
        crops = sw.smooth_blob_images(2, size, seed=607)
        loc, score = m(crops)                       # range guard trips -> reload as f32 -> answer
        assert m.dtype == "f32" and np.isfinite(loc).all()
        oloc, oscore, taps = helpers.oracle_student(w, crops)
        safe = helpers.heat_margins(taps) > 2e-3 * max(1.0, float(np.abs(taps["hm"].numpy()).max()))
        if safe.any():
            assert np.abs(score - oscore)[safe].max() < 1e-3 * float(np.abs(taps["hm"].numpy()[:, :98]).max())
    finally:
        m.engine.close()


def test_facade_falls_back_to_f32_emulator(emu_library, student_weights):
    _facade_fallback(emu_library, student_weights, 64)


@pytest.mark.gpu
def test_facade_falls_back_to_f32_gpu(hip_library, student_weights):
    _facade_fallback(hip_library, student_weights, 256)


def _fused_hrnet_guard(make_engine):
    """PF_OP_CHAIN / PF_OP_BLOCK split intermediates that never reach HBM: an overflow INSIDE the op (block input in range,
    conv1's output beyond f16) must be reported by the op itself."""
    from peppa_pig_face_landmark_amd.graph import ir
    for c, hw, fused in ((72, 16, "chain"), (18, 16, "block")):
        for boost, fails in ((1.0, False), (4.0e5, True)):
            rng = np.random.default_rng(77 + c)
            pb = ir.ProgramBuilder("f32s", 2 * hw, 2 * hw)
            f0 = pb.stem(rng.normal(0, 0.6, (16, 3, 3, 3)), rng.normal(0, 0.1, 16), "relu")
            x = pb.conv(f0, rng.normal(0, 0.35, (c, 16, 1, 1)), rng.normal(0, 0.2, c), "none")
            std = np.sqrt(2.0 / (9 * c))
            w1, b1 = rng.normal(0, std, (c, c, 3, 3)) * boost, rng.normal(0, 0.05, c)      # conv1's output reaches ~1e5
            w2, b2 = rng.normal(0, std, (c, c, 3, 3)) / boost, rng.normal(0, 0.05, c)
            y = pb.basic_chain(x, [(w1, b1, w2, b2)]) if fused == "chain" else pb.basic_block(x, w1, b1, w2, b2)
            # the outputs must be activation buffers of the right size; a 1x1 conv to 4 channels gives (hw*hw*4) floats -- use
            # plain F32 buffers instead and keep y alive through a consumer
            pb.conv(y, rng.normal(0, 0.1, (16, c, 1, 1)), np.zeros(16), "none")
            blob = pb.finish([pb.buffer(196, ir.ELEM_F32, "loc"), pb.buffer(98, ir.ELEM_F32, "score")])
            eng = make_engine()
            try:
                eng.load_program(0, blob, 2)
                crops = rng.integers(0, 256, (2, 2 * hw, 2 * hw, 3), dtype=np.uint8)
                if fails:
                    with pytest.raises(_native.PeppaHipError, match="activation range check failed.*above"):
                        eng.landmark_forward(crops)
                else:
                    eng.landmark_forward(crops)
            finally:
                eng.close()


def test_fused_hrnet_ops_report_internal_overflow_emulator(emu_library):
    from peppa_pig_face_landmark_amd._native import Engine
    _fused_hrnet_guard(lambda: Engine(0, emu_library))


@pytest.mark.gpu
def test_fused_hrnet_ops_report_internal_overflow_gpu(hip_library):
    from peppa_pig_face_landmark_amd._native import Engine
    _fused_hrnet_guard(lambda: Engine(0, hip_library))


def _tracking_fallback(library, student_weights, detector_weights):
    """pf_track_frame under the range guard (round-2 advisor finding): the failing frame has already been folded into the
    stream's device state (NaN track boxes, has_track) when the guard reports it; the engine must drop that state, so the
    facade's retry -- landmark network reloaded as exact f32 -- runs the detector again and answers like an f32 engine does."""
    from tests.test_tracking_parity import _make_facer
    from tests.tracking_video import video
    frames, rows = video()
    w = _scaled(student_weights, 3.0e5)
    state = {"i": 0}

    def run(dtype):
        from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
        from Skps import FaceAna
        cfg = get_cfg()
        cfg["Skps"]["Engine"]["device_tracking"] = True
        cfg["Skps"]["Detect"]["input_shape"] = [384, 640, 3]
        cfg["Skps"]["Keypoints"]["input_shape"] = [64, 64, 3]
        cfg["Skps"]["Engine"]["dtype"] = dtype
        f = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": w}, library=library)
        f._planted_rows = lambda: rows[state["i"]]
        out = []
        try:
            for i in range(3):
                state["i"] = i
                out.append(f.run(frames[i].copy()))
            return out, f.face_landmark.model.dtype, f.face_detector.model.dtype
        finally:
            f.engine.close()

    want, _, _ = run("f32")
    got, lm_dtype, det_dtype = run("f32s")
    assert lm_dtype == "f32" and det_dtype == "f32s"           # the network that failed was reloaded, the other one was not
    for a, b in zip(want, got):
        assert len(a) == len(b) >= 2
        for x, y in zip(a, b):
            assert np.isfinite(np.asarray(y["box"], np.float64)).all() and np.isfinite(np.asarray(y["kps"], np.float64)).all()
            assert np.array_equal(np.asarray(x["box"], np.float64), np.asarray(y["box"], np.float64))
            assert np.array_equal(np.asarray(x["kps"], np.float64), np.asarray(y["kps"], np.float64))


def test_tracking_state_rolled_back_on_guard_failure_emulator(emu_library, student_weights, detector_weights):
    _tracking_fallback(emu_library, student_weights, detector_weights)


@pytest.mark.gpu
def test_tracking_state_rolled_back_on_guard_failure_gpu(hip_library, student_weights, detector_weights):
    _tracking_fallback(hip_library, student_weights, detector_weights)


def _guard_fires_on_any_call(make_engine, student_weights, size):
    """Always-on guard (round-2 verdict: the every-256th-call schedule left 255 calls unguarded): the SECOND call of a healthy
    program is fed an input that drives the activations past f16's range -- it must fail like a first call would, and the
    third call, healthy again, must answer exactly like the first."""
    eng = make_engine()
    try:
        blob, _ = build_student_program(student_weights, size, "f32s")
        eng.load_program(0, blob, 2)
        crops = sw.smooth_blob_images(2, size, seed=608)
        x = np.ascontiguousarray(crops.astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0))
        a = eng.landmark_forward(x)
        assert np.isfinite(a[0]).all()
        with pytest.raises(_native.PeppaHipError, match="activation range check failed.*above"):
            eng.landmark_forward(x * np.float32(3.0e6))
        b = eng.landmark_forward(x)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        eng.close()


def test_guard_fires_on_any_call_emulator(emu_library, student_weights):
    from peppa_pig_face_landmark_amd._native import Engine
    _guard_fires_on_any_call(lambda: Engine(0, emu_library), student_weights, 64)


@pytest.mark.gpu
def test_guard_fires_on_any_call_gpu(hip_library, student_weights):
    from peppa_pig_face_landmark_amd._native import Engine
    _guard_fires_on_any_call(lambda: Engine(0, hip_library), student_weights, 256)
