"""GPU diagnosis of the block kernels (csrc/k_mbx.h): per-block agreement with the layer-wise path (5 faces) and run-to-run agreement
per block at 256 faces (keep_all programs)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from peppa_pig_face_landmark_amd._native import Engine
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.graph.random_init import random_student_weights
from bench_support import synthetic_crops
from tests import helpers
lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else None
w = random_student_weights(0)
names = ["encoder.blocks.%d.%d.out" % (s, b) for s, n in ((3, 4), (4, 2), (5, 3)) for b in range(n)]
C = {"3": 80, "4": 112, "5": 160}
def run(eng, mbx, crops, reps):
    blob, info = build_student_program(w, 256, "f32s", keep_all=True, fuse_mbx=mbx)
    eng.load_program(0, blob, crops.shape[0])
    out = []
    for _ in range(reps):
        eng.landmark_forward(crops)
        out.append({n: helpers.read_engine_tensor(eng, 0, info, n, crops.shape[0], (16, 16, C[n.split(".")[2]]), 4) for n in names})
    return out
eng = Engine(0, lib)
c5 = synthetic_crops(5, 256, 3)
old = run(eng, False, c5, 1)[0]
new = run(eng, True, c5, 2)
for n in names:
    print("5 faces  %-26s rel diff vs layer-wise %.3e   run0==run1 %s" % (n, np.abs(new[0][n] - old[n]).max() / np.abs(old[n]).max(), np.array_equal(new[0][n], new[1][n])))
c = synthetic_crops(256, 256, 11)
runs = run(eng, True, c, 5)
for n in names:
    bad = [int((r[n] != runs[0][n]).reshape(256, -1).any(1).sum()) for r in runs[1:]]
    worst = max(float(np.abs(r[n] - runs[0][n]).max()) for r in runs[1:])
    print("256 faces %-26s faces differing from run 0: %s  worst |diff| %.3e" % (n, bad, worst))
eng.close()
