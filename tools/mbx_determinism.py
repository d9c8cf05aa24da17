"""GPU check of the block kernels (csrc/k_mbx.h): run to run determinism and agreement with the layer-wise path, per library flavour.
usage: python tools/mbx_determinism.py lib1.so [lib2.so ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from peppa_pig_face_landmark_amd._native import Engine
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.graph.random_init import random_student_weights
from bench_support import synthetic_crops
w = random_student_weights(0)
crops = synthetic_crops(256, 256, 11)
for lib in sys.argv[1:]:
    eng = Engine(0, os.path.abspath(lib))
    outs = {}
    for mbx in (False, True):
        blob, _ = build_student_program(w, 256, "f32s", fuse_mbx=mbx)
        eng.load_program(0, blob, 256)
        outs[mbx] = [eng.landmark_forward(crops) for _ in range(12)]
    ref = outs[False][0]
    det_old = all(np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]) for r in outs[False])
    new = outs[True]
    det_new = [bool(np.array_equal(r[0], new[0][0]) and np.array_equal(r[1], new[0][1])) for r in new]
    nbad = [int((r[1] != new[0][1]).any(1).sum()) for r in new]
    print("%-40s layer-wise deterministic %s | mbx runs equal run 0: %s faces differing %s | score diff mbx-old max %.3e q99 %.3e" % (
        os.path.basename(lib), det_old, det_new, nbad, float(np.abs(new[0][1] - ref[1]).max()), float(np.quantile(np.abs(new[0][1] - ref[1]), 0.99))), flush=True)
    eng.close()
