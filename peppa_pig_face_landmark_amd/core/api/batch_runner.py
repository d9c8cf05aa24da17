"""FrameBatchRunner: ``FaceAna.run(image)`` + ``reset()`` (Skps/core/api/facer.py:52-85, the demo's still-image loop
demo.py:76-86) over a BATCH of frames in one call.

The reference processes one frame per call and one face per session run (face_landmark.py:40-48; "batched" is a TODO at
:119).  Here the frames of a call are split over ``lanes`` engines on one GPU (``pf_batch_*`` of the C ABI: one HIP stream
and one activation arena per lane), each lane running letterbox -> detector -> NMS -> area filter / top-k -> crop ->
landmark regressor -> back-projection for its slice; the lanes' kernels overlap on the device.  Same configuration keys
as ``FaceAna`` (``Skps.yml``), same result format per frame.  No tracking state: every frame is detected afresh, which is
what ``run()`` followed by ``reset()`` computes (tracking over a video stream is ``FaceAna`` with
``Engine.device_tracking``, one stream per instance)."""
from __future__ import annotations

import pathlib
from typing import Dict, List, Optional

import numpy as np

from ... import _native
from ...logger.logger import logger
from ...graph.detector import build_detector_program
from ...graph.student import build_student_program
from ..smoother.lk import EmaFilter
from .facer import _box_iou, _load_weights, get_cfg
from .hip_model_base import _RANGE_ERR


class FrameBatchRunner:
    def __init__(self, cfg: Optional[dict] = None, weights: Optional[dict] = None, lanes: int = 3, frames_per_lane: int = 32,
                 device: Optional[int] = None, library: Optional[str] = None, graph: bool = True, top_k: Optional[int] = None):
        cfg = cfg or get_cfg()
        sk = cfg["Skps"]
        eng_cfg = sk.get("Engine", {})
        self.device = int(eng_cfg.get("device", 0)) if device is None else int(device)
        self.dtype = eng_cfg.get("dtype", "f32s")
        root = pathlib.Path(__file__).resolve().parents[2]
        weights = weights or {}
        self._det_w = weights.get("detector") or _load_weights(root, sk["Detect"]["model_path"], "detector")
        self._arch = str(sk["Keypoints"].get("model", "student"))
        self._kps_w = weights.get("keypoints") or _load_weights(root, sk["Keypoints"]["model_path"],
                                                                "teacher" if self._arch == "teacher" else "keypoints")
        self._det_shape = tuple(int(v) for v in sk["Detect"]["input_shape"][:2])
        self._kps_size = int(sk["Keypoints"]["input_shape"][0])
        self.score_thrs = float(sk["Detect"]["score_thrs"])
        self.iou_thrs = float(sk["Detect"]["iou_thrs"])
        self.min_face = float(sk["Detect"]["min_face"])
        self.top_k = int(top_k if top_k is not None else sk["Detect"]["topk"])
        self.track_iou_thres = float(sk["Trace"]["iou_thres"])
        self._box_filter = EmaFilter(float(sk["Trace"]["smooth_box"]))
        self.lanes, self.frames_per_lane = int(lanes), int(frames_per_lane)
        self.engine = _native.BatchEngine(self.device, self.lanes, library)
        self.engine.set_option(_native.PF_OPT_HIP_GRAPH, 1 if graph else 0)
        self._load(_native.PF_NET_DETECTOR, self.dtype)
        self._load(_native.PF_NET_LANDMARK, self.dtype)

    @property
    def max_frames(self) -> int:
        return self.lanes * self.frames_per_lane

    def _load(self, slot: int, dtype: str):
        if slot == _native.PF_NET_DETECTOR:
            blob, _ = build_detector_program(self._det_w, self._det_shape, dtype)
            self.engine.load_program(slot, blob, self.frames_per_lane)
        else:
            if self._arch == "teacher":
                from ...graph.teacher import build_teacher_program
                blob, _ = build_teacher_program(self._kps_w, self._kps_size, dtype)
            else:
                blob, _ = build_student_program(self._kps_w, self._kps_size, dtype)
            self.engine.load_program(slot, blob, self.frames_per_lane * self.top_k)

    def _guarded(self, fn, *args, **kw):
        """The f32s range guard (PF_OPT_RANGE_CHECK) names the program whose activations left the representable range: that
        network is reloaded on every lane with exact-f32 convolutions and the call repeated (as FaceAna does)."""
        for _ in range(3):
            try:
                return fn(*args, **kw)
            except _native.PeppaHipError as e:
                m = _RANGE_ERR.search(str(e))
                if not m:
                    raise
                logger.warning("range guard: %s -- reloading network %s as exact f32 on all %d lanes", e, m.group(1), self.lanes)
                self._load(int(m.group(1)), "f32")
        raise _native.PeppaHipError("range guard fallback did not converge")

    def run_arrays(self, frames: np.ndarray, planted_rows: Optional[np.ndarray] = None):
        """frames ``[F,H,W,3]`` BGR uint8 (F <= lanes * frames_per_lane) -> (counts [F], boxes [F,top_k,4], landmarks
        [F,top_k,98,2], scores [F,top_k,98]); rows of a frame beyond its count are undefined."""
        frames = np.asarray(frames)
        if frames.ndim != 4 or frames.shape[-1] != 3 or frames.dtype != np.uint8:
            raise ValueError("frames must be uint8 [F,H,W,3]")
        if frames.shape[0] > self.max_frames:
            raise ValueError("%d frames exceed lanes * frames_per_lane = %d" % (frames.shape[0], self.max_frames))
        return self._guarded(self.engine.run_frames, frames, self.score_thrs, self.iou_thrs, self.min_face, self.top_k, planted_rows)

    def _returned_boxes(self, det_boxes: np.ndarray, kps: np.ndarray) -> np.ndarray:
        """The 'box' a fresh ``FaceAna.run`` hands back (facer.py:81-84): not the detector's box but the hull of the face's
        landmarks, EMA-smoothed against the first detector box it overlaps (``judge_boxs(boxes_return, hulls)``)."""
        hulls = np.array([[np.min(l[:, 0]), np.min(l[:, 1]), np.max(l[:, 0]), np.max(l[:, 1])] for l in kps])
        out = []
        for i in range(hulls.shape[0]):
            for j in range(det_boxes.shape[0]):
                if _box_iou(hulls[i], det_boxes[j]) > self.track_iou_thres:
                    out.append(self._box_filter(hulls[i][:4], det_boxes[j][:4]))
                    break
            else:
                out.append(hulls[i][0:4])
        return np.array(out)

    def run(self, frames) -> List[List[Dict[str, np.ndarray]]]:
        """Per frame what ``FaceAna.run(frame)`` returns for a fresh instance: ``[{'box', 'kps', 'scores'}, ...]`` -- 'box' is
        the smoothed landmark hull like the reference's, the detector's own box rides along as 'det_box'."""
        frames = np.stack(frames) if isinstance(frames, (list, tuple)) else np.asarray(frames)
        out: List[List[Dict[str, np.ndarray]]] = []
        for s in range(0, frames.shape[0], self.max_frames):
            counts, boxes, kps, scores = self.run_arrays(frames[s:s + self.max_frames])
            for f in range(counts.shape[0]):
                n = int(counts[f])
                ret = self._returned_boxes(boxes[f, :n], kps[f, :n]) if n else np.zeros((0, 4), np.float32)
                out.append([{"box": ret[i], "kps": kps[f, i], "scores": scores[f, i], "det_box": boxes[f, i]} for i in range(n)])
        return out

    def close(self):
        self.engine.close()
