"""A race net for the hand-counted LDS-DMA rings (round 5; VERDICT r04 "What's weak" 2).

Every hot kernel of the f32s programs orders its global -> LDS rings with ``s_waitcnt vmcnt(N)`` in front of a raw
``s_barrier`` (``pf_wait_vm_barrier<N>``: k_hero.h, k_sepup.h, k_chain.h, k_hrb.h, the unrolled pointwise GEMM).
An N that is one too large lets a wave read a stage whose bytes have not landed: a stale lo plane is ~1e-4 relative, inside
every tolerance of the parity tests, and depends on what else runs on the chip.  Two nets that a tolerance cannot hide from:

  * the bench shape (96 x 1080p x 8 planted faces, hipGraph replay) on 2 / 3 / 6 lanes, repeated, against ONE engine on ONE
    stream (nothing else on the chip): every box, landmark and score of all 768 faces ``np.array_equal``;
  * the production library against ``libpeppa_hip_strict.so`` (the same sources with -DPF_STRICT_WAITS=1: every partial wait
    drains to vmcnt(0)) on the Student, Teacher and detector programs: outputs bit for bit.
"""
import numpy as np
import pytest

from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd import _native
from peppa_pig_face_landmark_amd.graph.detector import build_detector_program
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program
from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows


@pytest.fixture(scope="module")
def strict_library():
    from peppa_pig_face_landmark_amd import build
    return build.build_hip(flavour="strict", verbose=False)


def _bench_inputs(F, K, H, W):
    base = [make_frame(H, W, K, seed=7 + i) for i in range(4)]
    rng = np.random.default_rng(3)
    frames = np.empty((F, H, W, 3), np.uint8)
    rows = np.empty((F, 15120, 16), np.float32)
    for f in range(F):
        nz = rng.integers(-3, 4, (H, W, 3), dtype=np.int16)
        frames[f] = np.clip(base[f % 4][0].astype(np.int16) + nz, 0, 255).astype(np.uint8)
        rows[f] = plant_rows(base[f % 4][1], (H, W), 15120, (384, 640), 24, seed=7 + f % 4)
    return frames, rows


@pytest.mark.gpu
def test_multi_lane_graph_replay_equals_one_engine_bit_for_bit(hip_library, student_weights, detector_weights):
    F, K, H, W = 96, 8, 1080, 1920
    import torch
    frames, rows = _bench_inputs(F, K, H, W)
    d_frames = torch.from_numpy(frames).cuda()
    d_rows = torch.from_numpy(rows).cuda()
    lm_blob = build_student_program(student_weights, 256, "f32s")[0]
    det_blob = build_detector_program(detector_weights, (384, 640), "f32s")[0]

    def outputs(host):
        return [host.pinned_empty((F,), np.int32), host.pinned_empty((F, K, 4), np.float32),
                host.pinned_empty((F, K, 98, 2), np.float32), host.pinned_empty((F, K, 98), np.float32)]

    # reference: ONE engine, ONE stream, eager launches, all 96 frames in one call -- nothing else runs on the chip
    one = _native.Engine(0, hip_library)
    one.load_program(_native.PF_NET_LANDMARK, lm_blob, F * K)
    one.load_program(_native.PF_NET_DETECTOR, det_blob, F)
    ref = outputs(one)
    for a in ref:
        a[...] = 0
    one.run_frames_device(d_frames.data_ptr(), F, H, W, 0.5, 0.3, 1600.0, K, d_planted=d_rows.data_ptr(), rows=15120,
                          d_counts=ref[0].ctypes.data, d_boxes=ref[1].ctypes.data, d_kps=ref[2].ctypes.data,
                          d_scores=ref[3].ctypes.data, out_mem=_native.PF_MEM_HOST_PINNED)
    one.sync()
    ref = [np.array(a) for a in ref]
    one.close()
    assert ref[0].tolist() == [K] * F
    assert np.isfinite(ref[2]).all() and np.isfinite(ref[3]).all()

    runs = 0
    for lanes, reps in ((3, 5), (2, 4), (6, 4)):
        be = _native.BatchEngine(0, lanes, hip_library)
        be.set_option(_native.PF_OPT_HIP_GRAPH, 1)
        per = (F + lanes - 1) // lanes
        be.load_program(_native.PF_NET_LANDMARK, lm_blob, per * K)
        be.load_program(_native.PF_NET_DETECTOR, det_blob, per)
        got = outputs(be)
        for rep in range(reps):          # eager, capture, replays
            for a in got:
                a[...] = 0
            be.run_frames_device(d_frames.data_ptr(), F, H, W, 0.5, 0.3, 1600.0, K, d_planted=d_rows.data_ptr(), rows=15120,
                                 d_counts=got[0].ctypes.data, d_boxes=got[1].ctypes.data, d_kps=got[2].ctypes.data,
                                 d_scores=got[3].ctypes.data, out_mem=_native.PF_MEM_HOST_PINNED)
            be.sync()
            for name, r, g in zip(("counts", "boxes", "landmarks", "scores"), ref, got):
                if not np.array_equal(r, g):
                    bad = np.argwhere(np.asarray(r != g).reshape(F, -1).any(1)).ravel()
                    diff = np.abs(np.asarray(g, np.float64) - np.asarray(r, np.float64)).max()
                    raise AssertionError("%d lanes, run %d: %s differ from the one-engine run on %d frames (first %s), worst |diff| %.3e"
                                         % (lanes, rep, name, bad.size, bad[:8].tolist(), diff))
            runs += 1
        be.close()
    assert runs >= 10
    print("race net: %d multi-lane runs (2 / 3 / 6 lanes, graph replay) bit-identical with one engine on all %d faces" % (runs, F * K))


def _both(hip_library, strict_library, fn):
    outs = []
    for lib in (hip_library, strict_library):
        eng = _native.Engine(0, lib)
        try:
            outs.append(fn(eng))
        finally:
            eng.close()
    return outs


@pytest.mark.gpu
def test_production_equals_strict_waits_student(hip_library, strict_library, student_weights):
    blob = build_student_program(student_weights, 256, "f32s")[0]
    crops = sw.smooth_blob_images(256, 256, seed=911)

    def run(eng):
        eng.load_program(0, blob, 256)
        res = [eng.landmark_forward(crops) for _ in range(3)]
        for r in res[1:]:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1])     # run to run
        return res[0]

    (loc_p, sc_p), (loc_s, sc_s) = _both(hip_library, strict_library, run)
    assert np.isfinite(loc_p).all() and np.isfinite(sc_p).all()
    assert np.array_equal(loc_p, loc_s), "landmarks: production != strict waits on %d values" % int((loc_p != loc_s).sum())
    assert np.array_equal(sc_p, sc_s), "scores: production != strict waits, worst %.3e" % float(np.abs(sc_p - sc_s).max())


@pytest.mark.gpu
def test_production_equals_strict_waits_teacher(hip_library, strict_library):
    weights = sw.teacher_weights()
    blob = build_teacher_program(weights, 256, "f32s")[0]
    crops = sw.smooth_blob_images(64, 256, seed=912)

    def run(eng):
        eng.load_program(0, blob, 64)
        return eng.landmark_forward(crops)

    (loc_p, sc_p), (loc_s, sc_s) = _both(hip_library, strict_library, run)
    assert np.isfinite(loc_p).all() and np.isfinite(sc_p).all()
    assert np.array_equal(loc_p, loc_s), "landmarks: production != strict waits on %d values" % int((loc_p != loc_s).sum())
    assert np.array_equal(sc_p, sc_s), "scores: production != strict waits, worst %.3e" % float(np.abs(sc_p - sc_s).max())


@pytest.mark.gpu
def test_production_equals_strict_waits_detector(hip_library, strict_library, detector_weights):
    blob = build_detector_program(detector_weights, (384, 640), "f32s")[0]
    rng = np.random.default_rng(17)
    frames = np.stack([make_frame(384, 640, 3, seed=60 + i, face_w=90, face_h=120)[0] for i in range(4)])
    x = np.concatenate([np.clip(frames.astype(np.int16) + rng.integers(-4, 5, frames.shape), 0, 255).astype(np.uint8) for _ in range(8)])

    def run(eng):
        eng.load_program(_native.PF_NET_DETECTOR, blob, x.shape[0])
        return eng.detector_forward(x)

    rows_p, rows_s = _both(hip_library, strict_library, run)
    assert rows_p.shape == (32, 15120, 16) and np.isfinite(rows_p).all()
    assert np.array_equal(rows_p, rows_s), "detector rows: production != strict waits, worst %.3e" % float(np.abs(rows_p - rows_s).max())
