"""Generate tests/golden/tracking_video.npz by running the REFERENCE's own FaceAna (Skps/core/api/facer.py and
Skps/core/smoother/lk.py executed from source; the two network sessions are oracle callables fed planted detector rows)
over the 5-frame synthetic video of tests/tracking_video.py.  Build container only: needs /root/reference.
Usage:  python tests/golden/make_tracking_golden.py

The vector holds, per frame, the number of results and each result's float64 'box' (4), 'kps' (98, 2) and 'scores' (98)
as facer.py:88-118 returns them -- what the diff gate, judge_boxs, sort_and_filter, the One-Euro GroupTrack and the hull
boxes produce between the networks.
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_import as ri  # noqa: E402
from oracle import synth_weights as sw  # noqa: E402
from tests.test_tracking_parity import reference_run  # noqa: E402
from tests.tracking_video import GOLDEN  # noqa: E402


def main():
    assert ri.available(), "needs /root/reference"
    runs, track_dtype = reference_run(sw.student_weights())
    n, top = len(runs), max(len(r) for r in runs)
    box = np.zeros((n, top, 4), np.float64)
    kps = np.zeros((n, top, 98, 2), np.float64)
    scores = np.zeros((n, top, 98), np.float64)
    counts = np.zeros(n, np.int64)
    for i, res in enumerate(runs):
        counts[i] = len(res)
        for j, r in enumerate(res):
            box[i, j], kps[i, j], scores[i, j] = r["box"], r["kps"], r["scores"]
    np.savez_compressed(GOLDEN, counts=counts, box=box, kps=kps, scores=scores, track_box_dtype=str(track_dtype),
                        numpy_version=np.__version__)
    print("tracking golden written: counts", counts.tolist(), "track_box dtype", track_dtype)


if __name__ == "__main__":
    main()
