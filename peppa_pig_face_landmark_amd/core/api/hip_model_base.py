"""HIPEngine: the MI355X replacement of the reference's ``ONNXEngine``
(Skps/core/api/onnx_model_base.py:6-27): construct from a model, call with one float32 NCHW (or
uint8 NHWC) array, get the list of output arrays back.  Execution happens in libpeppa_hip.so."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from ... import _native
from ...graph.detector import build_detector_program
from ...graph.student import build_student_program


class HIPEngine:
    """``HIPEngine(weights, kind)(data) -> [outputs...]`` -- same call shape as ``ONNXEngine``.

    kind='keypoints': data [B,3,S,S] float32 (/255) or [B,S,S,3] uint8 -> [landmark [B,196], score [B,98]]
                      (the two outputs of kps_student.onnx, face_landmark.py:48-51)
    kind='detector' : data [B,3,H,W] float32 RGB/255 or [B,H,W,3] uint8 -> [rows [B,R,16]]
                      (yolov5n-0.5.onnx, face_detector.py:29-31)
    """

    def __init__(self, weights: Dict[str, np.ndarray], kind: str, input_shape, device: int = 0, dtype: str = "f32",
                 max_batch: int = 8, engine: Optional[_native.Engine] = None, library: Optional[str] = None):
        self.kind = kind
        self.engine = engine if engine is not None else _native.Engine(device, library)
        if kind == "keypoints":
            blob, self.info = build_student_program(weights, int(input_shape[0]), dtype)
            self.slot = _native.PF_NET_LANDMARK
        elif kind == "detector":
            blob, self.info = build_detector_program(weights, (int(input_shape[0]), int(input_shape[1])), dtype)
            self.slot = _native.PF_NET_DETECTOR
        else:
            raise ValueError(kind)
        self.max_batch = max_batch
        self.engine.load_program(self.slot, blob, max_batch)

    def __call__(self, data: np.ndarray) -> List[np.ndarray]:
        if data.shape[0] > self.max_batch:
            raise ValueError(f"batch {data.shape[0]} exceeds max_batch {self.max_batch}")
        if self.kind == "keypoints":
            loc, score = self.engine.landmark_forward(data)
            return [loc, score]
        return [self.engine.detector_forward(data, self.info["rows"])]
