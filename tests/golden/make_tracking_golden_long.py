"""Generate tests/golden/tracking_video_long.npz: the REFERENCE's own FaceAna (facer.py + lk.py executed from source; the
detector session returns planted rows, the landmark session is the engine's f32 Student on the SIMT emulator -- see
tests/test_tracking_parity.py::reference_run_long for why not the torch oracle) over the 13-frame / two-segment video of
tests/tracking_video.py::video_long, with reset() between the segments.  Build container only: needs /root/reference.
Usage:  python tests/golden/make_tracking_golden_long.py"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_import as ri  # noqa: E402
from oracle import synth_weights as sw  # noqa: E402
from tests.test_tracking_parity import reference_run_long  # noqa: E402
from tests.tracking_video import GOLDEN_LONG  # noqa: E402


def main():
    assert ri.available(), "needs /root/reference"
    from tests.simt_emu import build_emu
    runs, detector_ran = reference_run_long(sw.student_weights(), build_emu.build_emu())
    n, top = len(runs), max(len(r) for r in runs)
    box = np.zeros((n, top, 4), np.float64)
    kps = np.zeros((n, top, 98, 2), np.float64)
    scores = np.zeros((n, top, 98), np.float64)
    counts = np.zeros(n, np.int64)
    for i, res in enumerate(runs):
        counts[i] = len(res)
        for j, r in enumerate(res):
            box[i, j], kps[i, j], scores[i, j] = r["box"], r["kps"], r["scores"]
    np.savez_compressed(GOLDEN_LONG, counts=counts, box=box, kps=kps, scores=scores, detector_ran=np.asarray(detector_ran, np.int64),
                        numpy_version=np.__version__)
    print("long tracking golden written: counts", counts.tolist(), "detector ran", detector_ran)


if __name__ == "__main__":
    main()
