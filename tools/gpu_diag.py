"""GPU diagnostic: per-layer error table of the HIP landmark regressor vs the oracle (test tool)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from oracle import synth_weights as sw  # noqa: E402
from peppa_pig_face_landmark_amd._native import Engine  # noqa: E402
from peppa_pig_face_landmark_amd.graph.student import build_student_program  # noqa: E402
from tests import helpers  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    out = {}
    w = sw.student_weights()
    eng = Engine(0)
    out["version"] = eng.version()
    for dtype, ve in (("f32", 4), ("f16", 8)):
        B = 2
        blob, info = build_student_program(w, size, dtype, keep_all=True, debug_full_hm=True)
        eng.load_program(0, blob, B)
        crops = sw.smooth_blob_images(B, size, seed=77)
        loc, score = eng.landmark_forward(crops)
        oloc, oscore, taps = helpers.oracle_student(w, crops)
        rows = []
        for name in info["tensors"]:
            if name not in taps:
                continue
            ref = helpers.tap_nhwc(taps, name)
            got = helpers.read_engine_tensor(eng, 0, info, name, B, ref.shape[1:], ve)
            rows.append((name, float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)), bool(np.isfinite(got).all())))
        d = np.abs(loc - oloc).reshape(B, 98, 2).max(2)
        out[dtype] = {"layers": rows, "loc_max_err": float(d.max()), "loc_median_err": float(np.median(d)),
                      "score_max_err": float(np.abs(score - oscore).max()),
                      "frac_landmarks_within_1e-3": float((d < 1e-3).mean())}
        print(dtype, "loc max err", d.max(), "median", np.median(d), "within 1e-3:", (d < 1e-3).mean())
        for r in rows:
            print("   %-42s %.3e %s" % r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag_%d.json" % size), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
