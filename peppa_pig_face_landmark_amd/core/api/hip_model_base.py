"""HIPEngine: the MI355X replacement of the reference's ``ONNXEngine``
(Skps/core/api/onnx_model_base.py:6-27): construct from a model, call with one float32 NCHW (or
uint8 NHWC) array, get the list of output arrays back.  Execution happens in libpeppa_hip.so."""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence

import numpy as np

from ... import _native
from ...graph.detector import build_detector_program
from ...graph.student import build_student_program


_RANGE_ERR = re.compile(r"activation range check failed: input of op \d+ of program (\d+)")


def run_guarded(models: Sequence["HIPEngine"], fn, *args):
    """``fn(*args)`` with the f32s range-guard fallback for a call that spans several networks (``pf_track_frame`` runs the
    detector AND the landmark regressor): the error names the program slot whose activations left the representable range,
    and THAT network is rebuilt with exact-f32 convolutions before the call is repeated -- reloading the wrong one would fail
    again on every later frame.  One retry per network at most."""
    for _ in range(len(models) + 1):
        try:
            return fn(*args)
        except _native.PeppaHipError as e:
            m = _RANGE_ERR.search(str(e))
            if not m:
                raise
            failed = [x for x in models if x.slot == int(m.group(1)) and x.dtype != "f32"]
            if not failed:
                raise
            from ...logger.logger import logger
            logger.warning("%s network: %s -- falling back to dtype f32", failed[0].kind, e)
            failed[0]._load("f32")
    raise _native.PeppaHipError("range guard fallback did not converge")


class HIPEngine:
    """``HIPEngine(weights, kind)(data) -> [outputs...]`` -- same call shape as ``ONNXEngine``.

    kind='keypoints': data [B,3,S,S] float32 (/255) or [B,S,S,3] uint8 -> [landmark [B,196], score [B,98]]
                      (the two outputs of kps_student.onnx, face_landmark.py:48-51)
    kind='detector' : data [B,3,H,W] float32 RGB/255 or [B,H,W,3] uint8 -> [rows [B,R,16]]
                      (yolov5n-0.5.onnx, face_detector.py:29-31)
    """

    def __init__(self, weights: Dict[str, np.ndarray], kind: str, input_shape, device: int = 0, dtype: str = "f32",
                 max_batch: int = 8, engine: Optional[_native.Engine] = None, library: Optional[str] = None,
                 arch: str = "student"):
        if kind not in ("keypoints", "detector"):
            raise ValueError(kind)
        self.kind = kind
        self.engine = engine if engine is not None else _native.Engine(device, library)
        self.slot = _native.PF_NET_LANDMARK if kind == "keypoints" else _native.PF_NET_DETECTOR
        self.max_batch = max_batch
        if arch not in ("student", "teacher"):
            raise ValueError("keypoint architecture must be 'student' or 'teacher'")
        self.arch = arch           # TeacherNet (HRNet-W18 encoder, model.py:302-345) exports to the same two-output graph
        self._weights, self._input_shape = weights, input_shape
        self._load(dtype)

    def _load(self, dtype: str):
        if self.kind == "keypoints" and self.arch == "teacher":
            from ...graph.teacher import build_teacher_program
            blob, self.info = build_teacher_program(self._weights, int(self._input_shape[0]), dtype)
        elif self.kind == "keypoints":
            blob, self.info = build_student_program(self._weights, int(self._input_shape[0]), dtype)
        else:
            blob, self.info = build_detector_program(self._weights, (int(self._input_shape[0]), int(self._input_shape[1])), dtype)
        self.dtype = dtype
        self.engine.load_program(self.slot, blob, self.max_batch)

    def guarded(self, fn, *args):
        """Run ``fn(*args)``; if the engine's range guard reports activations the split-precision (f32s) convolutions
        cannot represent (PF_OPT_RANGE_CHECK: outputs NaN + error), reload this network with exact-f32 MFMA convolutions
        and run again -- slower, never wrong."""
        return run_guarded([self], fn, *args)

    def __call__(self, data: np.ndarray) -> List[np.ndarray]:
        if data.shape[0] > self.max_batch:
            raise ValueError(f"batch {data.shape[0]} exceeds max_batch {self.max_batch}")
        if self.kind == "keypoints":
            loc, score = self.guarded(self.engine.landmark_forward, data)
            return [loc, score]
        return [self.guarded(self.engine.detector_forward, data, self.info["rows"])]
