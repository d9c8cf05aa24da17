"""Synthetic 1080p-style frames + planted detector rows for tests and bench."""
import numpy as np


def make_frame(h=1080, w=1920, n_faces=8, seed=7, face_w=200, face_h=260):
    """BGR uint8 frame: noisy grey background + n skin-coloured ellipses with dark blobs; returns
    (frame, boxes_xyxy float32 [n,4]) with boxes laid out on a 4-column grid like the survey's C3."""
    rng = np.random.default_rng(seed)
    frame = np.clip(np.rint(114 + rng.normal(0, 6, (h, w, 3))), 0, 255).astype(np.uint8)
    cols = 4 if n_faces > 1 else 1
    rows = (n_faces + cols - 1) // cols
    boxes = []
    yy, xx = np.mgrid[0:h, 0:w]
    for k in range(n_faces):
        i, j = k % cols, k // cols
        cx = (i + 0.5) * w / cols
        cy = (j + 0.5) * h / rows
        fw, fh = face_w * h / 1080.0, face_h * h / 1080.0
        m = ((xx - cx) / (fw / 2)) ** 2 + ((yy - cy) / (fh / 2)) ** 2 <= 1.0
        frame[m] = (140, 170, 210)
        for dx, dy, r in ((-0.2, -0.15, 0.09), (0.2, -0.15, 0.09), (0.0, 0.25, 0.14)):
            mm = (xx - (cx + dx * fw)) ** 2 + (yy - (cy + dy * fh)) ** 2 <= (r * fw) ** 2
            frame[mm] = (40, 40, 60)
        boxes.append([cx - fw / 2, cy - fh / 2, cx + fw / 2, cy + fh / 2])
    return frame, np.asarray(boxes, np.float32)


def make_frame_grid(h, w, cols, rows, face_w=200.0, face_h=260.0, seed=5):
    """BGR uint8 frame with cols x rows faces of fixed size (2160x3840, 8 x 4 grid, 200 x 260); the k-th
    face is 3k pixels wider so that top-k-by-area is unambiguous.  Returns (frame, boxes_xyxy float32 [cols*rows, 4])."""
    rng = np.random.default_rng(seed)
    frame = np.clip(np.rint(114 + rng.normal(0, 6, (h, w, 3))), 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    boxes = []
    for k in range(cols * rows):
        i, j = k % cols, k // cols
        cx, cy = (i + 0.5) * w / cols, (j + 0.5) * h / rows
        fw, fh = face_w + 3 * k, face_h
        frame[((xx - cx) / (fw / 2)) ** 2 + ((yy - cy) / (fh / 2)) ** 2 <= 1.0] = (140, 170, 210)
        for dx, dy, r in ((-0.2, -0.15, 0.09), (0.2, -0.15, 0.09), (0.0, 0.25, 0.14)):
            frame[(xx - (cx + dx * fw)) ** 2 + (yy - (cy + dy * fh)) ** 2 <= (r * fw) ** 2] = (40, 40, 60)
        boxes.append([cx - fw / 2, cy - fh / 2, cx + fw / 2, cy + fh / 2])
    return frame, np.asarray(boxes, np.float32)


def plant_rows(boxes_xyxy, frame_hw, n_rows=15120, input_hw=(384, 640), per_box=24, seed=7):
    """Decoded-detector-output rows [n_rows,16] in letterboxed coordinates: for every box 24 jittered
    candidates with distinct scores in (0.55, 0.99) -- the best one is the un-jittered box -- and
    every other row scored below 0.3, so NMS must return exactly one row per box."""
    rng = np.random.default_rng(seed)
    h, w = frame_hw
    scale = min(input_hw[0] / h, input_hw[1] / w)
    rw, rh = int(w * scale), int(h * scale)
    top = int(round((input_hw[0] - rh) / 2 - 0.1))
    left = int(round((input_hw[1] - rw) / 2 - 0.1))
    rows = np.zeros((n_rows, 16), np.float32)
    rows[:, 0] = rng.uniform(0, input_hw[1], n_rows)
    rows[:, 1] = rng.uniform(0, input_hw[0], n_rows)
    rows[:, 2:4] = rng.uniform(4, 60, (n_rows, 2))
    rows[:, 4] = rng.uniform(0.0, 0.3, n_rows)
    rows[:, 5:] = rng.uniform(0, 1, (n_rows, 11))
    slots = rng.choice(n_rows, size=len(boxes_xyxy) * per_box, replace=False)
    scores = rng.permutation(np.linspace(0.55, 0.985, len(slots))).astype(np.float32)
    for k, b in enumerate(boxes_xyxy):
        cx, cy = (b[0] + b[2]) / 2 * scale + left, (b[1] + b[3]) / 2 * scale + top
        bw, bh = (b[2] - b[0]) * scale, (b[3] - b[1]) * scale
        for t in range(per_box):
            r = slots[k * per_box + t]
            if t == 0:
                rows[r, :4] = (cx, cy, bw, bh)
                rows[r, 4] = 0.99 + 0.0005 * k
            else:
                rows[r, :4] = (cx + rng.uniform(-2, 2), cy + rng.uniform(-2, 2),
                               bw * rng.uniform(0.95, 1.05), bh * rng.uniform(0.95, 1.05))
                rows[r, 4] = scores[k * per_box + t]
    return rows
