"""Full-size parity of the detector, the pre/post kernels and the fused pipeline on a real MI355X
(through the C ABI) against the oracle.  Byte / index work is bit-exact; float work within the
north-star tolerance (1e-3 of the crop size for landmarks)."""
import numpy as np
import pytest
import torch

from oracle import detector_net as dn
from oracle import prepost as pp
from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd.graph.detector import build_detector_program
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame1080():
    return make_frame(1080, 1920, 8, seed=7)


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_detector_384x640_matches_oracle(gpu_engine, detector_weights, dtype):
    blob, info = build_detector_program(detector_weights, (384, 640), dtype, keep_all=True)
    assert info["rows"] == 15120                       # face_detector.py:31
    gpu_engine.load_program(1, blob, 2)
    img = sw.smooth_blob_images(2, 640, seed=9)[:, :384]
    rows = gpu_engine.detector_forward(img, 15120)
    W = {k: torch.from_numpy(v) for k, v in detector_weights.items()}
    x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    with torch.no_grad():
        ref = dn.detector_forward(W, x, taps).numpy()
    for name, tid in info["tensors"].items():
        if name in taps:
            r = taps[name].permute(0, 2, 3, 1).numpy()
            g = gpu_engine.read_tensor(1, tid, 2, r.shape[1:])
            assert np.abs(g - r).max() / (np.abs(r).max() + 1e-9) < 2e-4, name
    assert np.abs(rows - ref).max() / np.abs(ref).max() < 2e-4


@pytest.mark.parametrize("hw", [(1080, 1920), (2160, 3840), (720, 1280), (273, 410), (768, 1280)])
def test_letterbox_bit_exact(gpu_engine, hw):
    frame, _ = make_frame(hw[0], hw[1], 4, seed=13)
    got, info = gpu_engine.letterbox(frame, (384, 640))
    ref, rinfo = pp.detector_preprocess_u8(frame, (384, 640))
    assert np.array_equal(got, ref)
    assert (info[0], info[1], info[2]) == (np.float32(rinfo[0]), rinfo[1], rinfo[2])


def test_nms_15120_rows_bit_exact(gpu_engine, frame1080):
    frame, boxes = frame1080
    rows = plant_rows(boxes, (1080, 1920), 15120, (384, 640), 24, seed=7)
    _, info = pp.detector_preprocess_u8(frame, (384, 640))
    info = [np.float32(info[0]), info[1], info[2]]
    ref = pp.detector_postprocess(rows, info, 0.3, 0.5)
    assert ref.shape[0] == 8                           # exactly the planted faces survive
    got = gpu_engine.nms_rows(rows, info[0], info[1], info[2], 0.5, 0.3)
    assert np.array_equal(got, ref)
    # dense case: thousands of overlapping candidates above threshold
    rng = np.random.default_rng(1)
    dense = rows.copy()
    dense[:, 4] = rng.permutation(np.linspace(0.2, 0.999, 15120)).astype(np.float32)
    ref = pp.detector_postprocess(dense, info, 0.3, 0.5)
    got = gpu_engine.nms_rows(dense, info[0], info[1], info[2], 0.5, 0.3, max_n=1024)
    assert np.array_equal(got, ref[:1024])


def test_crops_256_bit_exact(gpu_engine, frame1080):
    frame, boxes = frame1080
    extra = np.array([[-40.0, -30.0, 160.0, 220.0], [1700.0, 900.0, 1919.0, 1079.0], [300.0, 300.0, 318.0, 380.0],
                      [600.0, 200.0, 600.0 + 366.0, 640.0]], np.float32)   # last: 2*floor(0.7*366)=512 -> exact 2x path
    allb = np.concatenate([boxes, extra], 0)
    crops, params = gpu_engine.crop_faces(frame, allb, 256)
    for i, b in enumerate(allb):
        ci = pp.landmark_crop_box(b, 1080, 1920)
        assert bool(params[i, 0]) == ci.valid
        if ci.valid:
            assert (params[i, 1], params[i, 2], params[i, 3], params[i, 6], params[i, 7]) == \
                   (ci.add, ci.x0, ci.y0, ci.w_crop, ci.h_crop)
            assert np.array_equal(crops[i], pp.landmark_crop(frame, ci, (256, 256))), i


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_run_frames_1080p_x8_planted(gpu_engine, student_weights, detector_weights, frame1080, dtype):
    """BASELINE configs[2] shape: 1080p frames x 8 faces, detector net running, planted detections."""
    F, K = 2, 8
    blob, _ = build_student_program(student_weights, 256, dtype)
    gpu_engine.load_program(0, blob, F * K)
    blob, _ = build_detector_program(detector_weights, (384, 640), dtype)
    gpu_engine.load_program(1, blob, F)
    frames, rows_all = [], []
    for f in range(F):
        frame, boxes = make_frame(1080, 1920, K, seed=7 + f)
        frames.append(frame)
        rows_all.append(plant_rows(boxes, (1080, 1920), 15120, (384, 640), 24, seed=7 + f))
    counts, bout, kps, scores = gpu_engine.run_frames(np.stack(frames), 0.5, 0.3, 1600.0, K,
                                                      planted_rows=np.stack(rows_all))
    assert counts.tolist() == [K] * F
    worst = 0.0
    for f in range(F):
        _, info = pp.detector_preprocess_u8(frames[f], (384, 640))
        kept = pp.detector_postprocess(rows_all[f], [np.float32(info[0]), info[1], info[2]], 0.3, 0.5)
        ref_boxes = pp.sort_and_filter(kept, 1600.0, K)
        assert np.array_equal(bout[f], ref_boxes[:, :4])
        for k in range(K):
            ci = pp.landmark_crop_box(ref_boxes[k], 1080, 1920)
            crop = pp.landmark_crop(frames[f], ci, (256, 256))
            oloc, oscore, taps = helpers.oracle_student(student_weights, crop[None])
            ref = pp.landmark_backproject(oloc[0], ci)
            safe = helpers.heat_margins(taps)[0] > 2e-3
            worst = max(worst, float(np.abs(kps[f, k] - ref)[safe].max() / max(ci.w_crop, ci.h_crop)))
    print("pipeline: worst normalised landmark error %.2e" % worst)
    assert worst < 1e-3


def test_detect_chain_is_self_consistent(gpu_engine, detector_weights, frame1080):
    """pf_detect (letterbox -> net -> decode -> NMS -> scale_coords) equals the numpy post-processing
    applied stage by stage (pf_letterbox -> pf_detector_forward -> pf_nms_rows)."""
    frame, _ = frame1080
    blob, _ = build_detector_program(detector_weights, (384, 640), "f32")
    gpu_engine.load_program(1, blob, 1)
    got = gpu_engine.detect(frame, 0.5, 0.3, max_n=1024)
    lb, info = gpu_engine.letterbox(frame, (384, 640))
    rows = gpu_engine.detector_forward(lb[None], 15120)[0]
    # py_nms leaves the order of equal scores unspecified (np.argsort, face_detector.py:106) and a
    # random-weight detector saturates many scores to exactly 1.0f, so the numpy oracle cannot be the
    # checker here; test_nms_15120_rows_bit_exact pins the NMS kernel against it on distinct scores,
    # this test pins the plumbing of the fused call against the separately-tested stages.
    ref = gpu_engine.nms_rows(rows, info[0], info[1], info[2], 0.5, 0.3, max_n=1024)
    assert got.shape[0] > 0 and np.array_equal(got, ref)


def test_faceana_facade(hip_library, student_weights, detector_weights, frame1080):
    """Skps.FaceAna()/run()/reset() surface (facer.py:25-208) on the HIP engine."""
    from Skps import FaceAna
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    cfg = get_cfg()
    cfg["Skps"]["Detect"]["topk"] = 8
    facer = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": student_weights}, library=hip_library)
    frame, _ = frame1080
    res = facer.run(frame)
    assert isinstance(res, list)
    for r in res:
        assert r["box"].shape == (4,) and r["kps"].shape == (98, 2) and r["scores"].shape == (98,)
    res2 = facer.run(frame)           # static frame: detector skipped, tracked boxes reused
    assert len(res2) == len(res)
    facer.reset()
    assert facer.track_box is None and facer.previous_image is None
    facer.engine.close()


def test_config1_plumbing_student128_single_face(hip_library, student_weights, detector_weights):
    """BASELINE configs[0]: one 1080p frame, one face, Student@128 through FaceAna.run()
    (Keypoints.input_shape is honoured exactly like face_landmark.py:29,97-98)."""
    from Skps import FaceAna
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    cfg = get_cfg()
    cfg["Skps"]["Keypoints"]["input_shape"] = [128, 128, 3]
    facer = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": student_weights}, library=hip_library)
    frame, boxes = make_frame(1080, 1920, 1, seed=5)
    # the detector has random weights, so drive the landmark stage with the known box (plumbing check)
    lm, states = facer.face_landmark(frame, boxes)
    assert lm.shape == (1, 98, 2) and states.shape == (1, 98)
    ci = pp.landmark_crop_box(boxes[0], 1080, 1920)
    crop = pp.landmark_crop(frame, ci, (128, 128))
    oloc, oscore, taps = helpers.oracle_student(student_weights, crop[None])
    ref = pp.landmark_backproject(oloc[0], ci)
    safe = helpers.heat_margins(taps)[0] > 2e-3
    assert np.abs(lm[0] - ref)[safe].max() < 1e-3 * max(ci.w_crop, ci.h_crop)
    res = facer.run(frame)
    assert isinstance(res, list)
    facer.engine.close()


def test_frame_diff_gate_1080p(gpu_engine, frame1080):
    """K13 / SURVEY 8f N1: FaceAna.diff_frames (facer.py:98-118) on the device, exact integer sum."""
    frame, _ = frame1080
    rng = np.random.default_rng(1)
    other = np.clip(frame.astype(np.int16) + rng.integers(-12, 13, frame.shape), 0, 255).astype(np.uint8)
    assert gpu_engine.set_frame(frame) is None
    d = gpu_engine.set_frame(other)
    ref = np.abs(frame.astype(np.int64) - other.astype(np.int64)).sum() / 1080 / 1920 / 3.0
    assert d == ref
    assert gpu_engine.set_frame(other) == 0.0


def test_hip_graph_replay_equals_eager(gpu_engine, student_weights, frame1080):
    """PF_OPT_HIP_GRAPH: the captured graph of a device-resident pf_run_frames call reproduces the eager
    results bit for bit, also after the inputs change in place."""
    from peppa_pig_face_landmark_amd import _native
    F, K = 2, 8
    blob, _ = build_student_program(student_weights, 256, "f32s")
    gpu_engine.load_program(0, blob, F * K)
    dev = torch.device("cuda", 0)
    frames_np, rows_np = [], []
    for f in range(F):
        fr, boxes = make_frame(1080, 1920, K, seed=50 + f)
        frames_np.append(fr)
        rows_np.append(plant_rows(boxes, (1080, 1920), 15120, (384, 640), 24, seed=50 + f))
    frames = torch.from_numpy(np.stack(frames_np)).to(dev)
    rows = torch.from_numpy(np.stack(rows_np)).to(dev)
    outs = [torch.zeros(F, dtype=torch.int32, device=dev), torch.zeros(F * K, 4, device=dev),
            torch.zeros(F * K, 98, 2, device=dev), torch.zeros(F * K, 98, device=dev)]

    def run():
        gpu_engine.run_frames_device(frames.data_ptr(), F, 1080, 1920, 0.5, 0.3, 1600.0, K, d_planted=rows.data_ptr(),
                                     rows=15120, d_counts=outs[0].data_ptr(), d_boxes=outs[1].data_ptr(),
                                     d_kps=outs[2].data_ptr(), d_scores=outs[3].data_ptr())
        gpu_engine.sync()
        return [o.clone() for o in outs]

    eager = run()
    gpu_engine.set_option(_native.PF_OPT_HIP_GRAPH, 1)
    for _ in range(3):          # eager (first sighting), capture + launch, replay
        got = run()
        assert all(torch.equal(a, b) for a, b in zip(eager, got))
    frames.copy_(torch.flip(frames, dims=[0]))      # new content, same buffers -> replay must see it
    rows.copy_(torch.flip(rows, dims=[0]))
    replay = run()
    gpu_engine.set_option(_native.PF_OPT_HIP_GRAPH, 0)
    again = run()
    assert all(torch.equal(a, b) for a, b in zip(again, replay))
    assert int(replay[0].sum()) == F * K


def test_pinned_host_frames_equal_device_frames(gpu_engine, student_weights, frame1080):
    """SURVEY 8f N2 (frame ingest): a batch handed over in pf_host_alloc (page-locked) host memory and copied inside
    the call gives bit-identical results to the same batch already resident in HBM."""
    F, K = 2, 8
    blob, _ = build_student_program(student_weights, 256, "f32s")
    gpu_engine.load_program(0, blob, F * K)
    dev = torch.device("cuda", 0)
    frames_np, rows_np = [], []
    for f in range(F):
        fr, boxes = make_frame(1080, 1920, K, seed=70 + f)
        frames_np.append(fr)
        rows_np.append(plant_rows(boxes, (1080, 1920), 15120, (384, 640), 24, seed=70 + f))
    frames = torch.from_numpy(np.stack(frames_np)).to(dev)
    rows = torch.from_numpy(np.stack(rows_np)).to(dev)
    outs = [torch.zeros(F, dtype=torch.int32, device=dev), torch.zeros(F * K, 4, device=dev),
            torch.zeros(F * K, 98, 2, device=dev), torch.zeros(F * K, 98, device=dev)]
    gpu_engine.run_frames_device(frames.data_ptr(), F, 1080, 1920, 0.5, 0.3, 1600.0, K, d_planted=rows.data_ptr(),
                                 rows=15120, d_counts=outs[0].data_ptr(), d_boxes=outs[1].data_ptr(),
                                 d_kps=outs[2].data_ptr(), d_scores=outs[3].data_ptr())
    gpu_engine.sync()
    ref = [o.clone() for o in outs]
    assert ref[0].tolist() == [K] * F
    pinned = gpu_engine.pinned_empty((F, 1080, 1920, 3), np.uint8)
    pinned[...] = np.stack(frames_np)
    for o in outs:
        o.zero_()
    gpu_engine.run_frames_host_async(pinned, rows.data_ptr(), 15120, 0.5, 0.3, 1600.0, K, outs[0].data_ptr(),
                                     outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr())
    gpu_engine.sync()
    assert all(torch.equal(a, b) for a, b in zip(ref, outs))


def test_c5_teacher_4k_32_faces(gpu_engine):
    """BASELINE configs[4] / SURVEY 8d C5 shape: one 2160x3840 frame, 32 faces on an 8 x 4 grid (w = 200, h = 260),
    Teacher@256 (HRNet-W18): planted detections -> NMS -> top-32 -> crops -> Teacher -> landmarks, against the oracle
    chain on the same frame (boxes bit-exact, landmarks within the north-star tolerance)."""
    from oracle import landmark_net as ln
    from oracle import teacher_net as tn
    from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program
    from peppa_pig_face_landmark_amd.synth import make_frame_grid
    H, W, K = 2160, 3840, 32
    frame, boxes = make_frame_grid(H, W, 8, 4, seed=5)     # distinct areas: top-k order is unambiguous
    rows = plant_rows(boxes, (H, W), 15120, (384, 640), 24, seed=5)
    weights = sw.teacher_weights()
    blob, _ = build_teacher_program(weights, 256, "f32s")
    gpu_engine.load_program(0, blob, K)
    counts, bout, kps, scores = gpu_engine.run_frames(frame[None], 0.5, 0.3, 1600.0, K, planted_rows=rows[None])
    assert counts.tolist() == [K]
    _, info = pp.detector_preprocess_u8(frame, (384, 640))
    kept = pp.detector_postprocess(rows, [np.float32(info[0]), info[1], info[2]], 0.3, 0.5)
    ref_boxes = pp.sort_and_filter(kept, 1600.0, K)
    assert ref_boxes.shape[0] == K and np.array_equal(bout[0], ref_boxes[:, :4])
    assert np.isfinite(kps).all() and np.isfinite(scores).all()
    Wt = ln.to_torch(weights)
    worst = 0.0
    for k in (0, 13, 31):
        ci = pp.landmark_crop_box(ref_boxes[k], H, W)
        crop = pp.landmark_crop(frame, ci, (256, 256))
        x = torch.from_numpy(crop[None].astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
        taps = {}
        with torch.no_grad():
            oloc, _ = tn.teacher_forward(Wt, x, taps)
        ref = pp.landmark_backproject(oloc.numpy()[0], ci)
        safe = helpers.heat_margins(taps)[0] > 2e-3
        worst = max(worst, float(np.abs(kps[0, k] - ref)[safe].max() / max(ci.w_crop, ci.h_crop)))
    print("C5 Teacher 4K x 32: worst normalised landmark error %.2e" % worst)
    assert worst < 1e-3


def test_crop_faces_float64_rows_bit_exact_gpu(gpu_engine):
    """Tracked frames hand FaceLandmark float64 boxes (facer.py:66-81): pf_crop_faces_f64 / pf_landmarks_f64."""
    from tests.test_emu_pipeline import check_crop_faces_f64
    check_crop_faces_f64(gpu_engine)


def test_faceana_facade_with_teacher(hip_library, detector_weights):
    """Keypoints.model: teacher -- the FaceLandmark facade runs TeacherNet (model.py:302-345) instead of the Student."""
    import torch
    from Skps import FaceAna
    from oracle import landmark_net as ln
    from oracle import teacher_net as tn
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    tw = sw.teacher_weights()
    cfg = get_cfg()
    cfg["Skps"]["Keypoints"]["model"] = "teacher"
    facer = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": tw}, library=hip_library)
    frame, boxes = make_frame(1080, 1920, 2, seed=9)
    lm, states = facer.face_landmark(frame, boxes)
    assert lm.shape == (2, 98, 2)
    for i in range(2):
        ci = pp.landmark_crop_box(boxes[i], 1080, 1920)
        crop = pp.landmark_crop(frame, ci, (256, 256))
        with torch.no_grad():
            taps = {}
            oloc, _ = tn.teacher_forward(ln.to_torch(tw), torch.from_numpy(pp.landmark_input(crop)), taps)
        ref = pp.landmark_backproject(oloc[0].numpy(), ci)
        safe = helpers.heat_margins(taps)[0] > 2e-3
        assert np.abs(lm[i] - ref)[safe].max() < 1e-3 * max(ci.w_crop, ci.h_crop)
    facer.engine.close()
