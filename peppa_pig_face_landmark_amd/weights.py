"""Checkpoint importer (importer half): a COTRAIN ``state_dict`` as the reference's trainer saves
it (``TRAIN/face_landmark/lib/core/base_trainer/net_work.py`` -> ``torch.save(model.state_dict(), ...)``, keys
``student.*`` / ``teacher.*``, optionally behind DataParallel's ``module.``) becomes the flat ``{name: ndarray}``
dictionaries / ``.npz`` files that ``graph/student.py::build_student_program`` and ``graph/teacher.py::
build_teacher_program`` consume.  Every tensor the inference graph needs is checked for presence and shape against the
architecture inventory (``graph/random_init.py::student_param_shapes``, ``graph/teacher.py::teacher_param_shapes``), so
a checkpoint of a different architecture fails here, loudly, not as garbage landmarks later.

    python -m peppa_pig_face_landmark_amd.weights cotrain.pth --out-dir weights/      # writes kps_student.npz [, kps_teacher.npz]
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Iterable, Mapping, Optional, Tuple

import numpy as np

_BN_FIELDS = ("weight", "bias", "running_mean", "running_var")


def _expected(shapes: Iterable[Tuple[str, Tuple[int, ...], str]]) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}
    for name, shape, kind in shapes:
        if kind == "bn":
            for f in _BN_FIELDS:
                out[f"{name}.{f}"] = tuple(shape)
        else:
            out[name] = tuple(shape)
    return out


def _to_numpy(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def split_cotrain_state_dict(sd: Mapping[str, object]) -> Tuple[Dict[str, np.ndarray], Optional[Dict[str, np.ndarray]]]:
    """(student, teacher-or-None) from a COTRAIN state_dict; names relative to ``student.`` / ``teacher.``."""
    from .graph.random_init import student_param_shapes
    from .graph.teacher import teacher_param_shapes
    flat = {}
    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        if k.endswith("num_batches_tracked"):
            continue
        flat[k] = v

    def take(prefix: str, shapes) -> Optional[Dict[str, np.ndarray]]:
        exp = _expected(shapes)
        have = {k[len(prefix):]: v for k, v in flat.items() if k.startswith(prefix)}
        if not have:
            return None
        missing = sorted(set(exp) - set(have))
        if missing:
            raise ValueError("checkpoint lacks %d tensors of %s (first: %s)" % (len(missing), prefix.rstrip("."), missing[:3]))
        out: Dict[str, np.ndarray] = {}
        for name, shape in exp.items():
            arr = _to_numpy(have[name])
            if tuple(arr.shape) != shape:
                raise ValueError("%s%s has shape %s, the architecture needs %s" % (prefix, name, tuple(arr.shape), shape))
            out[name] = arr
        return out            # tensors the inference graph does not use (e.g. the dead `fc` head) are dropped

    student = take("student.", student_param_shapes())
    if student is None:
        raise ValueError("no 'student.*' tensors in the checkpoint")
    return student, take("teacher.", teacher_param_shapes())


def import_checkpoint(path: str, out_dir: str) -> Dict[str, str]:
    """``torch.load`` the checkpoint at ``path`` and write ``kps_student.npz`` (and ``kps_teacher.npz``) to out_dir."""
    import torch
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd and not any(str(k).startswith(("student.", "module.")) for k in sd):
        sd = sd["state_dict"]
    student, teacher = split_cotrain_state_dict(sd)
    os.makedirs(out_dir, exist_ok=True)
    written = {"student": os.path.join(out_dir, "kps_student.npz")}
    np.savez(written["student"], **student)
    if teacher is not None:
        written["teacher"] = os.path.join(out_dir, "kps_teacher.npz")
        np.savez(written["teacher"], **teacher)
    return written


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    out = sys.argv[sys.argv.index("--out-dir") + 1] if "--out-dir" in sys.argv else "."
    for model, p in import_checkpoint(sys.argv[1], out).items():
        print("%s -> %s" % (model, p))
