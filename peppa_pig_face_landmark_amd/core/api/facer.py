"""FaceAna: the public API of the reference (Skps/core/api/facer.py:25-208) on the MI355X engine.

``FaceAna().run(image_bgr) -> [{'box': (4,), 'kps': (98,2), 'scores': (98,)}, ...]`` and
``reset()`` keep the reference's semantics, including the frame-difference gate that skips the
detector on static video (facer.py:98-118), IoU matching + EMA of boxes against the previous frame
(:144-189), top-k by area (:120-142) and One-Euro landmark smoothing.  The two network stages run
on the GPU through ``pf_detect`` / ``pf_landmarks``; the per-frame bookkeeping stays host-side
Python exactly where the reference has it."""
from __future__ import annotations

import logging
import os
import pathlib
from typing import Dict, Optional

import numpy as np
import yaml

from ... import _native
from ...logger.logger import logger
from ..smoother.lk import EmaFilter, GroupTrack
from .face_detector import FaceDetector
from .face_landmark import FaceLandmark
from .hip_model_base import run_guarded


def get_cfg(path: Optional[str] = None):
    root = pathlib.Path(__file__).resolve().parents[2]
    path = path or os.path.join(root, "config", "Skps.yml")
    with open(path, encoding="UTF-8") as f:
        return yaml.safe_load(f)


def _load_weights(root, rel_path: str, what: str) -> Dict[str, np.ndarray]:
    """``model_path`` of Skps.yml -> weight dictionary.  Accepts the reference's own files --
    ``pretrained/yolov5n-0.5.onnx`` / ``pretrained/kps_student.onnx`` (Skps/config/Skps.yml:4,12; read without
    onnx / onnxruntime, weights.weights_from_onnx) -- as well as ``.npz`` archives and torch checkpoints."""
    from ...weights import load_weights
    path = rel_path if os.path.isabs(rel_path) else os.path.join(root, rel_path)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{what} weights not found at {path}.  Put the reference's model file there (the .onnx the reference "
            "loads at onnx_model_base.py:14 works as is; so do an .npz of state_dict arrays or a .pth checkpoint) or "
            "pass FaceAna(weights={'detector': {...}, 'keypoints': {...}}).")
    return load_weights(path, what)


def _host_imread(data: bytes):
    """Host decode of an image file the device decoder refuses: BGR uint8 [H,W,3] like cv2.imread(path) (IMREAD_COLOR: alpha
    dropped, greyscale / palette expanded), or None when no host decoder can read it (cv2.imread's answer too).  Uses Pillow
    when it is installed; nothing else in the package depends on it."""
    try:
        import io
        from PIL import Image
    except ImportError:
        return None
    try:
        from PIL import ImageOps
        with Image.open(io.BytesIO(data)) as im:
            im = ImageOps.exif_transpose(im)         # cv2.imread applies the EXIF orientation (IMREAD_COLOR without IGNORE_ORIENTATION)
            if im.mode in ("I;16", "I;16B", "I;16L", "I"):      # 16-bit greyscale: cv2.imread(IMREAD_COLOR) keeps the HIGH byte
                im = im.point(lambda v: v / 256.0).convert("L")
            rgb = np.asarray(im.convert("RGB"))
    except Exception:  # noqa: BLE001  (truncated / unknown format)
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _box_iou(a, b) -> float:
    total = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1])
    iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / (total - inter)


class FaceAna:
    def __init__(self, verbose: bool = False, cfg: Optional[dict] = None, weights: Optional[dict] = None,
                 device: Optional[int] = None, library: Optional[str] = None):
        if verbose:
            logger.setLevel(logging.DEBUG)
        cfg = cfg or get_cfg()
        sk = cfg["Skps"]
        eng_cfg = sk.get("Engine", {})
        dev = int(eng_cfg.get("device", 0)) if device is None else int(device)
        dtype = eng_cfg.get("dtype", "f32")
        root = pathlib.Path(__file__).resolve().parents[2]
        weights = weights or {}
        det_w = weights.get("detector") or _load_weights(root, sk["Detect"]["model_path"], "detector")
        kps_arch = str(sk["Keypoints"].get("model", "student"))
        kps_w = weights.get("keypoints") or _load_weights(root, sk["Keypoints"]["model_path"], "teacher" if kps_arch == "teacher" else "keypoints")

        # Engine.device_tracking: keep track_box / previous landmarks / One-Euro state on the GPU (pf_track_frame) instead of
        # walking the boxes through numpy between the two networks on every frame
        self.device_tracking = bool(eng_cfg.get("device_tracking", False))
        self._planted_rows = None    # test instrument: callable returning decoded detector rows that replace the detector's own
        self._det_cfg = sk["Detect"]
        self.top_k = sk["Detect"]["topk"]
        self.engine = _native.Engine(dev, library)      # one GPU, one stream, shared by both stages
        max_faces = max(int(eng_cfg.get("max_faces", 8)), int(self.top_k))
        self.face_detector = FaceDetector(sk["Detect"], det_w, engine=self.engine, dtype=dtype)
        self.face_landmark = FaceLandmark(sk["Keypoints"], kps_w, engine=self.engine, dtype=dtype, max_batch=max_faces)
        self.trace = GroupTrack(sk["Trace"])
        logger.info("model init done!")

        self.track_box = None
        self.previous_image = None
        self.previous_box = None
        self.diff_thres = 5
        self.min_face = sk["Detect"]["min_face"]
        self.iou_thres = sk["Trace"]["iou_thres"]
        self.alpha = sk["Trace"]["smooth_box"]
        self.filter = EmaFilter(self.alpha)

    # ---- per-frame entry ---------------------------------------------------------------------------
    def run(self, image: np.ndarray):
        # one host->device copy per frame: the frame stays resident for the gate, the detector and the
        # landmark stage; the frame-difference gate (facer.py:98-118) is evaluated on the GPU against the
        # previous resident frame (exact integer sum, same decision as the numpy code in diff_frames()).
        if self.device_tracking:
            boxes, kps, scores, _ = run_guarded(
                [self.face_detector.model, self.face_landmark.model], self.engine.track_frame, image, float(self._det_cfg["score_thrs"]), float(self._det_cfg["iou_thrs"]),
                float(self.min_face), int(self.top_k), float(self.iou_thres), float(self.alpha), float(self.diff_thres),
                self._planted_rows() if self._planted_rows is not None else None)
            self.previous_image = image
            self.track_box = boxes
            return self.to_dict(boxes, kps, scores)
        diff = self.engine.set_frame(image)
        self.previous_image = image
        if diff is None or self.track_box is None or diff > self.diff_thres:
            boxes = self.face_detector(None)
            boxes = self.judge_boxs(self.track_box, boxes)
            self.trace.previous_landmarks_set = None     # detector ran: smoothing history is dropped
        else:
            boxes = self.track_box
        boxes = self.sort_and_filter(boxes)
        boxes_return = np.array(boxes)
        landmarks, states = self.face_landmark(None, boxes)
        landmarks = self.trace.calculate(image, landmarks)
        hulls = [[np.min(l[:, 0]), np.min(l[:, 1]), np.max(l[:, 0]), np.max(l[:, 1])] for l in landmarks]
        self.track_box = self.judge_boxs(boxes_return, np.array(hulls))
        return self.to_dict(self.track_box, landmarks, states)

    def imread(self, path_or_bytes, want_host: bool = True):
        """``cv2.imread(path)`` of the reference's demo (demo.py:76) for baseline JPEG files: the Huffman stream is decoded on
        the host, dequantisation / inverse DCT / chroma upsampling / colour conversion run on the GPU (bit-identical with
        cv2.imread's libjpeg) and the frame never exists in host memory unless asked for.  Returns a ``DeviceFrame`` that
        ``run()`` accepts like an array; ``frame.numpy()`` is the BGR array for drawing."""
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else None
        if data is None:
            try:
                with open(path_or_bytes, "rb") as f:
                    data = f.read()
            except OSError:
                return None                      # cv2.imread returns None for a path it cannot read
        try:
            return self.engine.imread(bytes(data), want_host)
        except _native.PeppaHipError as e:
            # Everything else cv2.imread opens (demo.py:76) -- progressive / arithmetic-coded / CMYK JPEG, PNG, BMP, ... -- is
            # decoded on the HOST, as the reference itself does for every file, and handed to run() as the numpy array
            # cv2.imread would have returned; only baseline JPEG has a device decoder (csrc/jpeg.inl).
            # NOTE the type: a host-decoded file comes back as the ndarray cv2.imread returns, not as a DeviceFrame (run() takes both)
            frame = _host_imread(bytes(data))
            if frame is None:
                logger.warning("imread: %s", e)
            return frame

    def to_dict(self, bboxes, kps, states):
        return [{"box": bboxes[i], "kps": kps[i], "scores": states[i]} for i in range(len(bboxes))]

    def diff_frames(self, previous_frame, image) -> bool:
        """True -> run the detector (facer.py:98-118): mean absolute difference of the frames > 5."""
        if previous_frame is None:
            return True
        if previous_frame.shape != image.shape:
            return True
        diff = np.abs(previous_frame.astype(np.int16) - image.astype(np.int16)).sum(dtype=np.int64)
        return diff / previous_frame.shape[0] / previous_frame.shape[1] / 3.0 > self.diff_thres

    def sort_and_filter(self, bboxes):
        if len(bboxes) < 1:
            return []
        area = (bboxes[:, 2] - bboxes[:, 0]) * (bboxes[:, 3] - bboxes[:, 1])
        keep = area > self.min_face
        area, bboxes = area[keep], bboxes[keep, :]
        if bboxes.shape[0] > self.top_k:
            bboxes = bboxes[area.argsort()[-self.top_k:][::-1]]
        return np.array(bboxes)

    def judge_boxs(self, previuous_bboxs, now_bboxs):
        """Match each current box to the first previous box with IoU > thres and EMA-smooth it."""
        if previuous_bboxs is None:
            return now_bboxs
        out = []
        for i in range(now_bboxs.shape[0]):
            for j in range(previuous_bboxs.shape[0]):
                if _box_iou(now_bboxs[i], previuous_bboxs[j]) > self.iou_thres:
                    out.append(self.smooth(now_bboxs[i], previuous_bboxs[j]))
                    break
            else:
                out.append(now_bboxs[i][0:4])
        return np.array(out)

    def smooth(self, now_box, previous_box):
        return self.filter(now_box[:4], previous_box[:4])

    def reset(self):
        self.track_box = None
        self.previous_image = None
        self.previous_box = None
        if self.device_tracking:
            self.engine.track_reset()
        self.engine.forget_frames()
