#!/usr/bin/env python
"""ONNX files written by PyTorch's OWN exporter (test / fixture tooling; needs torch, never imported by the product).

The reference ships its networks only as ONNX exports: ``pretrained/kps_student.onnx`` written by
TRAIN/face_landmark/tools/convert_to_onnx.py:54-61 (``torch.onnx.export(model, dummy, path, opset_version=12)`` on
``COTRAIN(inference='student' | 'teacher')``) and ``pretrained/yolov5n-0.5.onnx`` by yolov5-face's export.py.  Both blobs
are absent from the checkout.  This module reproduces the export with the exporter that would have written them:

  * ``export_cotrain(path, inference, weights...)``: the REFERENCE's ``COTRAIN`` class (decoder, heads, ``postp`` from its
    own source; the timm encoder stands behind the oracle's restatement, as everywhere in oracle/ref_import.py) through
    ``torch.onnx.export(..., opset_version=12, dynamo=False)`` exactly as convert_to_onnx.py calls it;
  * ``export_detector(path, weights)``: oracle/detector_net.py wrapped in an ``nn.Module``.

torch's TorchScript exporter imports the ``onnx`` package only to attach onnxscript functions to the finished proto
(``onnx_proto_utils._add_onnxscript_fn``); there are none here, and ``onnx`` is not installed, so that hook is replaced by
the identity.

    python tools/export_onnx_genuine.py --write-topology     # regenerates peppa_pig_face_landmark_amd/graph/onnx_topology.json
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _export(model, x, path, output_names):
    import torch
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model, x, path, opset_version=12, dynamo=False, input_names=["input"], output_names=output_names)


def export_cotrain(path: str, inference: str, student_np, teacher_np=None, size: int = 256):
    import torch
    from oracle import ref_import as ri
    model = ri.load_reference_cotrain(student_np, teacher_np, inference=inference)
    _export(model, torch.zeros(1, 3, size, size), path, ["landmark", "score"])


def export_oracle_landmark(path: str, weights_np, arch: str = "student", size: int = 256):
    """The same export where the reference checkout is absent: the oracle's functional restatement of the whole network
    (oracle/landmark_net.py / teacher_net.py) behind an ``nn.Module`` -- same architecture, same exporter."""
    import torch
    from oracle import landmark_net as ln
    from oracle import teacher_net as tn

    class Landmark(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in weights_np.items()}

        def forward(self, x):
            out = ln.student_forward(self.w, x) if arch == "student" else tn.teacher_forward(self.w, x)
            return out[0], out[1]

    _export(Landmark().eval(), torch.zeros(1, 3, size, size), path, ["landmark", "score"])


def export_detector(path: str, weights_np, hw=(384, 640)):
    import torch
    from oracle import detector_net as dn

    class Detector(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in weights_np.items()}

        def forward(self, x):
            return dn.detector_forward(self.w, x)

    _export(Detector().eval(), torch.zeros(1, 3, hw[0], hw[1]), path, ["rows"])


def write_topology():
    """Conv-to-Conv adjacency of the three architectures as the genuine exporter lays them out, committed next to the
    graph builders: weights.weights_from_onnx checks every file against it (node ORDER alone cannot tell two neighbouring
    convolutions of identical shape apart)."""
    import tempfile
    from oracle import synth_weights as sw
    from peppa_pig_face_landmark_amd import onnx_lite, weights as W
    out = {}
    with tempfile.TemporaryDirectory() as d:
        sw_student, sw_teacher = sw.student_weights(), sw.teacher_weights()
        jobs = (("student", lambda p: export_cotrain(p, "student", sw_student, size=128)),
                ("teacher", lambda p: export_cotrain(p, "teacher", sw_student, sw_teacher, size=128)),
                ("detector", lambda p: export_detector(p, sw.detector_weights(), (128, 160))))
        for arch, job in jobs:
            p = os.path.join(d, arch + ".onnx")
            job(p)
            out[arch] = W.conv_topology(onnx_lite.read_model(p))
    dst = os.path.join(ROOT, "peppa_pig_face_landmark_amd", "graph", "onnx_topology.json")
    with open(dst, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print(dst, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    if "--write-topology" in sys.argv:
        write_topology()
    else:
        sys.exit(__doc__)
