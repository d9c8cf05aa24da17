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


# ---- the long video (round 3) -------------------------------------------------------------------------------------------------
GOLDEN_LONG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracking_video_long.npz")
S_LONG = 128    # landmark crop size of the long video


def scene(h, w, faces, seed):
    """BGR uint8 frame with one ellipse-face per (cx, cy, width, height) entry; returns (frame, boxes_xyxy float32)."""
    rng = np.random.default_rng(seed)
    frame = np.clip(np.rint(114 + rng.normal(0, 6, (h, w, 3))), 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    boxes = []
    for cx, cy, fw, fh in faces:
        frame[((xx - cx) / (fw / 2)) ** 2 + ((yy - cy) / (fh / 2)) ** 2 <= 1.0] = (140, 170, 210)
        for dx, dy, r in ((-0.2, -0.15, 0.09), (0.2, -0.15, 0.09), (0.0, 0.25, 0.14)):
            frame[(xx - (cx + dx * fw)) ** 2 + (yy - (cy + dy * fh)) ** 2 <= (r * fw) ** 2] = (40, 40, 60)
        boxes.append([cx - fw / 2, cy - fh / 2, cx + fw / 2, cy + fh / 2])
    return frame, np.asarray(boxes, np.float32)


def long_video_weights(student_weights):
    """The BN-calibrated synthetic Student with its 2 x 98 offset channels damped: random-init offsets are tens of heat-map
    cells, which throws landmarks (and the hull boxes the tracker makes from them) thousands of pixels out of the crop; at
    1/50 they stay within a cell, landmarks stay inside the crop, and parity can be judged in absolute pixels."""
    w = dict(student_weights)
    hw, hb = np.array(w["hm.weight"], np.float32), np.array(w["hm.bias"], np.float32)
    hw[98:] *= np.float32(0.02)
    hb[98:] *= np.float32(0.02)
    w["hm.weight"], w["hm.bias"] = hw, hb
    return w


def video_long():
    """13 frames in two segments (reset() between them -- the frame size changes).  What the frames exercise, in order:
    three faces; two repeats of the same frame (the difference gate closes: detector skipped, float64 track boxes feed the
    landmark stage, One-Euro history grows); a 5-pixel shift (gate opens, IoU-matched boxes are EMA-smoothed); a fourth face
    enters; seven faces of distinct sizes (more than top_k = 5: the five largest survive); the same frame again; three of
    them leave; down to two faces; the two shift; then 360 x 640 frames after reset(): three faces, a repeat, a shift.
    Returns [(frames, planted rows, (h, w))] per segment.

    Face widths avoid multiples of 5: FaceLandmark.preprocess computes ``face_width = 1.4 * w`` (face_landmark.py:83), which
    for a float32 ``w`` is a float64 product under the reference's pinned numpy 1.23 (1.4 * 90 = 125.99999999999999, // 2 =
    62) and a float32 one under numpy >= 2 (126.0, // 2 = 63): the engine follows the pinned version, the reference executed
    in this container follows numpy 2, and the two differ exactly where 1.4 w is an integer."""
    h, w = 270, 480
    A, B, C = (70, 80, 64, 84), (200, 180, 81, 104), (380, 90, 72, 94)
    D = (330, 200, 58, 76)
    seven = [(60, 70, 51, 66), (180, 70, 58, 76), (300, 70, 66, 86), (420, 70, 74, 96),
             (60, 200, 82, 106), (190, 200, 62, 80), (320, 200, 91, 116)]
    sh = lambda fs, dx: [(cx + dx, cy, fw, fh) for cx, cy, fw, fh in fs]           # noqa: E731
    layouts = [[A, B, C], [A, B, C], [A, B, C], sh([A, B, C], 5), sh([A, B, C], 5) + [D], seven, seven,
               [seven[3], seven[4], seven[6], seven[2]], [seven[4], seven[6]], sh([seven[4], seven[6]], 7)]
    seeds = [21, 21, 21, 22, 23, 24, 24, 25, 26, 27]
    seg1 = [scene(h, w, lay, seed) for lay, seed in zip(layouts, seeds)]
    h2, w2 = 360, 640
    E, F, G = (120, 120, 96, 124), (330, 220, 111, 142), (520, 130, 88, 114)
    lay2 = [[E, F, G], [E, F, G], sh([E, F, G], 6)]
    seg2 = [scene(h2, w2, lay, seed) for lay, seed in zip(lay2, [31, 31, 32])]
    out = []
    for seg, hw in ((seg1, (h, w)), (seg2, (h2, w2))):
        frames = [f for f, _ in seg]
        rows = [plant_rows(b, hw, 15120, (384, 640), 6, seed=40 + i) for i, (_, b) in enumerate(seg)]
        out.append((frames, rows, hw))
    return out
