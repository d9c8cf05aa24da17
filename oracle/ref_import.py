"""Oracle helper: run the REFERENCE's own ``model.py``.  TEST INFRASTRUCTURE ONLY.

Only usable where /root/reference exists (the build container); the GPU box never has it, so
nothing under ``-m gpu`` tests / ``smoke()`` / ``bench.py`` may call into this module.  Its one
job is to pin ``oracle/landmark_net.py`` (decoder, heads, ``postp``) against the reference's
executable source and to produce the committed golden vectors
(``tests/golden/make_golden.py``).

``TRAIN/face_landmark/lib/core/base_trainer/model.py`` imports ``timm`` (:9) and
``torchvision.models.mobilenetv3`` (:11), neither of which is installed; both are stubbed in
``sys.modules``.  ``timm.create_model`` hands back the oracle's restated MobileNetV3 feature
extractor (student, model.py:252-258) or a tiny stand-in with the right output shapes
(teacher, model.py:306-311; the teacher's output is discarded for ``inference='student'``).
No reference source is copied: the module is imported from where it lies, read-only, with
``sys.dont_write_bytecode`` set so the read-only tree is never written to.
"""
from __future__ import annotations

import os
import sys
import types
from typing import Dict

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"
_TRAIN_DIR = os.path.join(REFERENCE_ROOT, "TRAIN", "face_landmark")


def available() -> bool:
    return os.path.isfile(os.path.join(_TRAIN_DIR, "lib", "core", "base_trainer", "model.py"))


class _StudentEncoderStub(nn.Module):
    """Holds the oracle's restated encoder behind the attribute surface model.py touches
    (``.blocks[6]`` is overwritten with Identity at model.py:262)."""

    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([nn.Identity() for _ in range(7)])
        self.weights: Dict[str, torch.Tensor] = {}

    def forward(self, x):
        from . import landmark_net as ln

        return ln.encoder_forward(self.weights, x)


class _TeacherEncoderStub(nn.Module):
    """Stands in for timm hrnet_w18 features (model.py:306-311).  With ``weights`` set it runs the
    oracle's restated HRNet-W18; otherwise a cheap shape-compatible dummy (the teacher's output is
    discarded when ``inference='student'``)."""

    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(3, c, 1) for c in (64, 128, 256, 512)])
        self.weights: Dict[str, torch.Tensor] = {}

    def forward(self, x):
        if self.weights:
            from . import teacher_net as tn

            return tn.encoder_forward(self.weights, x)
        outs = []
        for i, c in enumerate(self.convs):
            s = 2 << i
            outs.append(c(torch.nn.functional.avg_pool2d(x, s, s)))
        return outs


def _install_stubs():
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")

        def create_model(model_name=None, **kw):
            if "mobilenetv3" in model_name:
                return _StudentEncoderStub()
            return _TeacherEncoderStub()

        timm.create_model = create_model
        sys.modules["timm"] = timm
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvmm = types.ModuleType("torchvision.models.mobilenetv3")
        tvmm.InvertedResidual = type("InvertedResidual", (nn.Module,), {})
        tvmm.InvertedResidualConfig = type("InvertedResidualConfig", (), {})
        tv.models = tvm
        tvm.mobilenetv3 = tvmm
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
        sys.modules["torchvision.models.mobilenetv3"] = tvmm


def load_reference_cotrain(weights_np: Dict[str, np.ndarray], teacher_np: Dict[str, np.ndarray] = None,
                           inference: str = "student"):
    """Instantiate the reference's ``COTRAIN(inference=...)`` and load ``weights_np`` (keys relative to
    ``student.``) and optionally ``teacher_np`` (keys relative to ``teacher.``) into it."""
    assert available(), "reference checkout not present"
    sys.dont_write_bytecode = True
    _install_stubs()
    if _TRAIN_DIR not in sys.path:
        sys.path.insert(0, _TRAIN_DIR)
    from lib.core.base_trainer.model import COTRAIN  # the reference's own class

    torch.manual_seed(0)
    model = COTRAIN(inference=inference, inp_size=(256, 256))
    model.eval()
    sd = model.state_dict()
    own = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    loaded = 0
    for k in list(sd.keys()):
        if not k.startswith("student."):
            continue
        rel = k[len("student."):]
        if rel.startswith("encoder."):
            continue  # stub encoder has no parameters of its own
        if rel.endswith("num_batches_tracked"):
            continue
        assert rel in own, f"oracle inventory is missing reference tensor {rel}"
        assert tuple(sd[k].shape) == tuple(own[rel].shape), (rel, sd[k].shape, own[rel].shape)
        sd[k] = own[rel].clone()
        loaded += 1
    model.load_state_dict(sd)
    model.student.encoder.weights = {k: v for k, v in own.items() if k.startswith("encoder.")}
    ref_names = {k[len("student."):] for k in sd if k.startswith("student.")
                 and not k.endswith("num_batches_tracked") and not k.startswith("student.encoder.")}
    own_names = {k for k in own if not k.startswith("encoder.")}
    assert ref_names == own_names, (sorted(ref_names ^ own_names))
    if teacher_np is not None:
        town = {k: torch.from_numpy(v) for k, v in teacher_np.items()}
        sd = model.state_dict()
        for k in list(sd.keys()):
            if not k.startswith("teacher.") or k.startswith("teacher.encoder.") or k.endswith("num_batches_tracked"):
                continue
            rel = k[len("teacher."):]
            assert rel in town and tuple(sd[k].shape) == tuple(town[rel].shape), rel
            sd[k] = town[rel].clone()
        model.load_state_dict(sd)
        model.teacher.encoder.weights = {k: v for k, v in town.items() if k.startswith("encoder.")}
    return model


# --------------------------------------------------------------------------------------
# the reference's own numpy pre/post-processing (face_detector.py / face_landmark.py)
# --------------------------------------------------------------------------------------
def _install_cv2_stub():
    """``cv2`` and ``onnxruntime`` are not installed; the reference modules import both at module
    scope.  cv2 is stubbed with the oracle's OpenCV restatement (so what gets pinned is the
    reference's box arithmetic / slicing / NMS / call structure, not OpenCV itself)."""
    from . import prepost as pp

    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.COLOR_BGR2RGB = 4
        cv2.BORDER_CONSTANT = 0
        cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
        cv2.resize = lambda img, dsize: pp.resize_linear_u8(np.ascontiguousarray(img), int(dsize[0]), int(dsize[1]))

        def copyMakeBorder(img, top, bottom, left, right, borderType=0, value=0):
            v = value if np.isscalar(value) else np.asarray(value)[:img.shape[2]]
            return pp.pad_constant(img, top, bottom, left, right, v)

        cv2.copyMakeBorder = copyMakeBorder
        cv2.absdiff = lambda a, b: np.abs(a.astype(np.int16) - b.astype(np.int16)).astype(np.uint8)
        sys.modules["cv2"] = cv2
    if "onnxruntime" not in sys.modules:
        sys.modules["onnxruntime"] = types.ModuleType("onnxruntime")


def load_reference_stages():
    """Returns (FaceDetector, FaceLandmark) -- the reference's classes, importable because cv2 /
    onnxruntime are stubbed.  Instances must be made with ``__new__`` (the constructors open .onnx files)."""
    assert available()
    sys.dont_write_bytecode = True
    _install_cv2_stub()
    skps = os.path.join(REFERENCE_ROOT, "Skps")
    if skps not in sys.path:
        sys.path.append(skps)  # the reference does the same at Skps/__init__.py:3-5
    import logging
    level = logging.getLogger().level
    from core.api.face_detector import FaceDetector
    from core.api.face_landmark import FaceLandmark
    logging.getLogger().setLevel(level)  # logger/logger.py:25 sets the root logger to DEBUG
    return FaceDetector, FaceLandmark


def reference_detector_stage(input_shape=(384, 640, 3), score_thrs=0.5, iou_thrs=0.3):
    FaceDetector, _ = load_reference_stages()
    d = FaceDetector.__new__(FaceDetector)
    d.input_size, d.score_thrs, d.iou_thrs = list(input_shape), score_thrs, iou_thrs
    return d


def reference_landmark_stage(input_shape=(256, 256, 3)):
    _, FaceLandmark = load_reference_stages()
    l = FaceLandmark.__new__(FaceLandmark)
    l.min_face, l.keypoints_num, l.input_size, l.extend = 20, 98, list(input_shape), [0.2, 0.3]
    return l


def reference_faceana(detector_model, landmark_model, top_k=5, min_face=1600, det_input=(384, 640, 3), kps_input=(256, 256, 3)):
    """The reference's own ``FaceAna`` (Skps/core/api/facer.py:25-208) with its own FaceDetector / FaceLandmark /
    GroupTrack / EmaFilter classes, executed from source; only the two ``ONNXEngine`` sessions are replaced by the given
    callables (``detector_model(x[1,3,H,W]) -> [rows]``, ``landmark_model(x[1,3,S,S]) -> (landmark[1,196], score[1,98])``)
    and cv2 by the oracle's restatement.  This is how the frame-to-frame logic -- diff gate, judge_boxs, sort_and_filter,
    One-Euro smoothing, float64 track boxes -- is pinned."""
    FaceDetector, FaceLandmark = load_reference_stages()
    import logging
    level = logging.getLogger().level
    from core.api.facer import FaceAna
    from core.smoother.lk import EmaFilter, GroupTrack
    logging.getLogger().setLevel(level)
    fa = FaceAna.__new__(FaceAna)
    fa.face_detector = reference_detector_stage(det_input)
    fa.face_detector.model = detector_model
    fa.face_landmark = reference_landmark_stage(kps_input)
    fa.face_landmark.model = landmark_model
    fa.trace = GroupTrack({"pixel_thres": 3, "smooth_box": 0.3, "iou_thres": 0.5})
    fa.track_box = fa.previous_image = fa.previous_box = None
    fa.diff_thres, fa.top_k, fa.min_face, fa.iou_thres, fa.alpha = 5, top_k, min_face, 0.5, 0.3
    fa.filter = EmaFilter(fa.alpha)
    return fa
