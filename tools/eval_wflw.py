#!/usr/bin/env python
"""WFLW NME evaluation on the MI355X engine -- the counterpart of the reference's
TRAIN/face_landmark/tools/eval_WFLW.py:96-142 (the only accuracy figure the reference publishes, README.md:34-37).

    python tools/eval_wflw.py --data_dir WFLW --weight kps_student.onnx|cotrain.pth|student.npz --model student --img_size 256

Same protocol as the reference, line for line: for every annotation of `WFLW_annotations/list_98pt_test/*.txt` the face
box is the hull of the 98 ground-truth points, the crop extends it by base_extend_range = [0.2, 0.3] (eval_WFLW.py:54-55,
train_config DATA.base_extend_range) on a zero-padded frame, the crop is resized to img_size x img_size (aspect NOT
preserved), the network predicts crop-normalised landmarks and NME = mean point error / inter-ocular distance (points 60
and 72) per class.  What differs is the execution: crops are resized by the engine (pf_resize, OpenCV's arithmetic) and go
through the regressor in batches (pf_landmark_forward) instead of one torch call per image.
Images are read with PIL (cv2 is not a dependency); the dataset itself is not redistributable and must be supplied.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

EXTEND = (0.2, 0.3)      # TRAIN/face_landmark/train_config.py DATA.base_extend_range


def load_test_lists(data_dir: str):
    """eval_WFLW.py:19-35: {class: [annotation lines]} keyed by the last '_' token of every list file."""
    d = os.path.join(data_dir, "WFLW_annotations", "list_98pt_test")
    out = {}
    for txt in sorted(os.listdir(d)):
        if "txt" not in txt:
            continue
        cls = txt.rsplit(".")[0].rsplit("_")[-1]
        with open(os.path.join(d, txt)) as f:
            out[cls] = [ln for ln in f.readlines() if ln.strip()]
    return out


def crop_for_eval(img: np.ndarray, kps: np.ndarray):
    """augmentationCropImage(img, bbox, joints, is_training=False) (eval_WFLW.py:38-82) with bbox = hull of the points
    (:109-112).  Returns (crop uint8 HxWx3, joints in crop pixels)."""
    bbox = np.array([float(np.min(kps[:, 0])), float(np.min(kps[:, 1])), float(np.max(kps[:, 0])), float(np.max(kps[:, 1]))]).astype(np.float32)
    add = int(max(bbox[2] - bbox[0], bbox[3] - bbox[1]))
    h, w = img.shape[:2]
    bimg = np.zeros((h + 2 * add, w + 2 * add, 3), np.uint8)
    bimg[add:add + h, add:add + w] = img
    center = np.array([(bbox[0] + bbox[2]) / 2.0, (bbox[1] + bbox[3]) / 2.0])
    bbox = bbox + np.float32(add)
    center = center + add
    joints = kps.astype(np.float32).copy()
    joints[:, :2] += add
    gt_w, gt_h = bbox[2] - bbox[0], bbox[3] - bbox[1]
    half_w = gt_w * (1 + EXTEND[0] * 2) // 2
    half_h = gt_h * (1 + EXTEND[1] * 2) // 2
    min_x, max_x = int(center[0] - half_w), int(center[0] + half_w)
    min_y, max_y = int(center[1] - half_h), int(center[1] + half_h)
    joints[:, 0] -= min_x
    joints[:, 1] -= min_y
    return bimg[max(min_y, 0):max_y, max(min_x, 0):max_x, :], joints


def nme(target: np.ndarray, preds: np.ndarray) -> float:
    """eval_WFLW.py:84-94 (inter-ocular normalisation: points 60 and 72)."""
    target = np.reshape(target, [-1, 98, 2])
    preds = np.reshape(preds, [-1, 98, 2])
    norm = np.linalg.norm(target[:, 60, :] - target[:, 72, :], axis=-1)
    return float(np.mean(np.mean(np.linalg.norm(preds - target, axis=-1), axis=-1) / norm))


def read_bgr(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])      # cv2.imread order


def evaluate(data_dir: str, weights, model: str = "student", img_size: int = 256, batch: int = 256, dtype: str = "f32s",
             library=None, device: int = 0, limit: int = 0, log=print):
    """Returns {class: mean NME}.  `weights`: a path (.onnx/.npz/.pth) or a {name: ndarray} dictionary."""
    from peppa_pig_face_landmark_amd._native import Engine
    from peppa_pig_face_landmark_amd.weights import load_weights
    if isinstance(weights, str):
        weights = load_weights(weights, "teacher" if model == "teacher" else "keypoints")
    if model == "teacher":
        from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program as build
    else:
        from peppa_pig_face_landmark_amd.graph.student import build_student_program as build
    eng = Engine(device, library)
    try:
        blob, _ = build(weights, img_size, dtype)
        eng.load_program(0, blob, batch)
        result = {}
        for cls, lines in load_test_lists(data_dir).items():
            if limit:
                lines = lines[:limit]
            scores = []
            crops, labels = [], []

            def flush():
                if not crops:
                    return
                loc, _ = eng.landmark_forward(np.stack(crops))
                for p, t in zip(loc, labels):
                    scores.append(nme(t, p[:196]))
                crops.clear()
                labels.clear()

            for ln in lines:
                dp = ln.split()
                kps = np.array(dp[:196], dtype=np.float32).reshape(-1, 2)
                image = read_bgr(os.path.join(data_dir, "WFLW_images", dp[-1]))
                crop, label = crop_for_eval(image, kps)
                h, w = crop.shape[:2]
                label[:, 0] /= w
                label[:, 1] /= h
                crops.append(eng.resize(crop, (img_size, img_size)))
                labels.append(label)
                if len(crops) == batch:
                    flush()
            flush()
            result[cls] = float(np.mean(scores)) if scores else float("nan")
            log("for cls: %s  nme: %s  (%d faces)" % (cls, result[cls], len(scores)))
        return result
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--weight", required=True, help=".onnx (kps_student.onnx), .pth COTRAIN checkpoint or .npz")
    ap.add_argument("--data_dir", required=True, help="directory holding WFLW_images/ and WFLW_annotations/")
    ap.add_argument("--img_size", type=int, default=256)
    ap.add_argument("--model", default="student", choices=["student", "teacher"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f32s", choices=["f32", "f32s", "f16"])
    ap.add_argument("--limit", type=int, default=0, help="evaluate only the first N faces of every class")
    a = ap.parse_args()
    evaluate(a.data_dir, a.weight, a.model, a.img_size, a.batch, a.dtype, limit=a.limit)


if __name__ == "__main__":
    main()
