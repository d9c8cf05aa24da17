"""The short synthetic video of the frame-to-frame tests (tests/test_tracking_parity.py) and of the golden vector the
reference's own facer.py / lk.py produced for it (tests/golden/make_tracking_golden.py)."""
import os

import numpy as np

from peppa_pig_face_landmark_amd.synth import make_frame, plant_rows

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracking_video.npz")
S = 64          # landmark crop size of these tests


def video():
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
