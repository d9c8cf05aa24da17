"""FaceLandmark: same interface as the reference stage (Skps/core/api/face_landmark.py:14-64) --
``FaceLandmark(cfg)(image, bboxes) -> (landmarks (n,98,2), states (n,98))`` -- but all faces of a
frame go through ONE batched GPU call (``pf_landmarks``): crop-box arithmetic, zero-pad crop +
cv2-style resize, Student regressor, heat-map decode and back-projection to frame coordinates.

Documented deviation: boxes with w <= 20 or h <= 20 px make the reference crash
(``preprocess`` returns ``(None, None)`` at :76-77 and ``.transpose`` is called on None at :44);
here they are skipped (no row is returned for them), as SURVEY.md section 8(b) specifies."""
from __future__ import annotations

import time

import numpy as np

from ...logger.logger import logger
from .hip_model_base import HIPEngine


class FaceLandmark:
    def __init__(self, cfg, weights, engine=None, device: int = 0, dtype: str = "f32", max_batch: int = 8,
                 library=None):
        self.min_face = 20
        self.keypoints_num = cfg["num_points"]
        self.input_size = cfg["input_shape"]
        self.extend = cfg["base_extend_range"]
        if abs(float(self.extend[0]) - 0.2) > 1e-12:
            raise ValueError("the engine implements base_extend_range[0] == 0.2 (Skps.yml:14)")
        # Keypoints.model: "student" (kps_student.onnx, the file the reference ships) or "teacher" (COTRAIN's TeacherNet,
        # what convert_to_onnx.py --model teacher exports)
        self.model = HIPEngine(weights, "keypoints", self.input_size, device=device, dtype=dtype,
                               max_batch=max_batch, engine=engine, library=library, arch=str(cfg.get("model", "student")))
        self.engine = self.model.engine

    def __call__(self, img, bboxes):
        """img: BGR uint8 frame, or None for the frame made resident by Engine.set_frame()."""
        bboxes = np.asarray(bboxes)
        if bboxes.dtype != np.float64:       # float64 rows (tracked frames) are passed through as they are
            bboxes = bboxes.astype(np.float32)
        bboxes = bboxes.reshape(-1, bboxes.shape[-1] if len(bboxes) else 4)
        if bboxes.shape[0] == 0:
            return np.array([]), np.array([])
        t0 = time.time()
        kps, scores, valid = self.model.guarded(self.engine.landmarks, img, bboxes[:, :4])
        dt = time.time() - t0
        logger.info("keypoints done, time consume: %.5f and %.5f per face", dt, dt / len(bboxes))
        return kps[valid], scores[valid]
