"""Checkpoint importer (importer half): a COTRAIN-style state_dict round-trips into the engine's weight
dictionaries, unused / bookkeeping tensors are dropped, and a wrong architecture is rejected."""
import numpy as np
import pytest
import torch

from peppa_pig_face_landmark_amd import weights as wimp
from peppa_pig_face_landmark_amd.graph.random_init import random_student_weights
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from peppa_pig_face_landmark_amd.graph.teacher import random_teacher_weights


def _checkpoint(student, teacher=None, prefix=""):
    sd = {}
    for k, v in student.items():
        sd[f"{prefix}student.{k}"] = torch.from_numpy(v)
    sd[f"{prefix}student.encoder.bn1.num_batches_tracked"] = torch.tensor(7)
    sd[f"{prefix}student.fc.weight"] = torch.zeros(3, 5)          # dead head of the reference (model.py), not on the path
    if teacher is not None:
        for k, v in teacher.items():
            sd[f"{prefix}teacher.{k}"] = torch.from_numpy(v)
    return sd


def test_roundtrip_student_and_teacher(tmp_path):
    sw_, tw_ = random_student_weights(3), random_teacher_weights(4)
    path = tmp_path / "cotrain.pth"
    torch.save(_checkpoint(sw_, tw_, prefix="module."), path)
    written = wimp.import_checkpoint(str(path), str(tmp_path / "out"))
    s = dict(np.load(written["student"]))
    t = dict(np.load(written["teacher"]))
    assert set(s) == set(sw_) and all(np.array_equal(s[k], sw_[k]) for k in sw_)
    assert set(t) == set(tw_) and all(np.array_equal(t[k], tw_[k]) for k in tw_)
    blob_a, _ = build_student_program(s, 128, "f32s")
    blob_b, _ = build_student_program(sw_, 128, "f32s")
    assert blob_a == blob_b                      # the imported weights pack to the identical program


def test_student_only_and_rejections():
    sw_ = random_student_weights(5)
    s, t = wimp.split_cotrain_state_dict(_checkpoint(sw_))
    assert t is None and set(s) == set(sw_)
    bad = _checkpoint(sw_)
    del bad["student.hm.bias"]
    with pytest.raises(ValueError, match="lacks"):
        wimp.split_cotrain_state_dict(bad)
    bad = _checkpoint(sw_)
    bad["student.hm.weight"] = torch.zeros(10, 128, 1, 1)
    with pytest.raises(ValueError, match="shape"):
        wimp.split_cotrain_state_dict(bad)
    with pytest.raises(ValueError, match="no 'student"):
        wimp.split_cotrain_state_dict({"something.else": torch.zeros(1)})
