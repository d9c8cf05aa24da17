"""FaceDetector: same interface as the reference stage (Skps/core/api/face_detector.py:11-42) --
``FaceDetector(cfg)(image_bgr) -> ndarray (n,16)`` -- with preprocess (letterbox), the yolov5-face
network, head decode, xywh->xyxy, score filter, greedy NMS and scale_coords all executed on the GPU
by ``pf_detect`` (one call, one small device->host copy of the kept rows)."""
from __future__ import annotations

import time

import numpy as np

from ...logger.logger import logger
from .hip_model_base import HIPEngine


class FaceDetector:
    def __init__(self, cfg, weights, engine=None, device: int = 0, dtype: str = "f32", max_batch: int = 1,
                 library=None):
        self.input_size = cfg["input_shape"]
        self.score_thrs = cfg["score_thrs"]
        self.iou_thrs = cfg["iou_thrs"]
        self.model = HIPEngine(weights, "detector", self.input_size, device=device, dtype=dtype,
                               max_batch=max_batch, engine=engine, library=library)
        self.engine = self.model.engine

    def __call__(self, image) -> np.ndarray:
        """image: BGR uint8 frame, or None for the frame made resident by Engine.set_frame()."""
        t0 = time.time()
        boxes = self.model.guarded(self.engine.detect, image, float(self.score_thrs), float(self.iou_thrs))
        logger.info("detect done, time consume: %.5f", time.time() - t0)
        return boxes
