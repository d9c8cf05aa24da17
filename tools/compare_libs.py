"""Bit-for-bit comparison of two builds of the library on the GPU (run on the GPU box): the Student on 256 crops, the Teacher on 64,
the detector on 32 frames -- every output ``np.array_equal``.  usage: python tools/compare_libs.py <libA.so> <libB.so>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import synth_weights as sw                                  # noqa: E402  (weights and inputs only: test infrastructure)
from peppa_pig_face_landmark_amd import _native                          # noqa: E402
from peppa_pig_face_landmark_amd.graph.detector import build_detector_program   # noqa: E402
from peppa_pig_face_landmark_amd.graph.student import build_student_program     # noqa: E402
from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program     # noqa: E402
from peppa_pig_face_landmark_amd.synth import make_frame                  # noqa: E402
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
stu = build_student_program(sw.student_weights(), 256, "f32s")[0]
tea = build_teacher_program(sw.teacher_weights(), 256, "f32s")[0]
det = build_detector_program(sw.detector_weights(), (384, 640), "f32s")[0]
crops = sw.smooth_blob_images(256, 256, seed=911)
frames = np.stack([make_frame(384, 640, 3, seed=60 + i, face_w=90, face_h=120)[0] for i in range(32)])
outs = []
for lib in libs:
    eng = _native.Engine(0, lib)
    eng.load_program(_native.PF_NET_LANDMARK, stu, 256)
    a = eng.landmark_forward(crops)
    eng.load_program(_native.PF_NET_LANDMARK, tea, 64)
    b = eng.landmark_forward(crops[:64])
    eng.load_program(_native.PF_NET_DETECTOR, det, 32)
    c = eng.detector_forward(frames)
    outs.append((a[0], a[1], b[0], b[1], c))
    eng.close()
ok = True
for name, x, y in zip(("student landmarks", "student scores", "teacher landmarks", "teacher scores", "detector rows"), outs[0], outs[1]):
    same = np.array_equal(x, y)
    ok &= same
    print("%-18s %s  %s  finite %s" % (name, x.shape, "IDENTICAL" if same else "DIFFERENT: %d values, worst %.3e" % (int((x != y).sum()), float(np.abs(x.astype(np.float64) - y).max())), bool(np.isfinite(x).all())))
sys.exit(0 if ok else 1)
