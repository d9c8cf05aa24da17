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
    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(3, c, 1) for c in (64, 128, 256, 512)])

    def forward(self, x):
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


def load_reference_cotrain(weights_np: Dict[str, np.ndarray]):
    """Instantiate the reference's ``COTRAIN(inference='student')`` and load ``weights_np``
    (keys relative to ``student.``) into it.  Returns the nn.Module in eval mode."""
    assert available(), "reference checkout not present"
    sys.dont_write_bytecode = True
    _install_stubs()
    if _TRAIN_DIR not in sys.path:
        sys.path.insert(0, _TRAIN_DIR)
    from lib.core.base_trainer.model import COTRAIN  # the reference's own class

    torch.manual_seed(0)
    model = COTRAIN(inference="student", inp_size=(256, 256))
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
    return model
