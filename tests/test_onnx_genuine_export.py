"""The weight importer on files PYTORCH wrote (round-2 verdict: it had only ever read files written by its own writer).

tools/export_onnx_genuine.py reproduces the reference's export -- TRAIN/face_landmark/tools/convert_to_onnx.py:54-61,
``torch.onnx.export(COTRAIN(inference=...), dummy, path, opset_version=12)`` -- with torch's TorchScript exporter, on the
reference's own ``COTRAIN`` class where the checkout is present; ``weights.weights_from_onnx`` must recover weights that drive
the oracle to the outputs of the original weights, for the Student, the Teacher and the detector.  The wiring check
(``weights.conv_topology``) is pinned here too: a file whose convolutions are ordered differently is refused, not mis-paired."""
import numpy as np
import pytest
import torch

from oracle import detector_net as dn
from oracle import landmark_net as ln
from oracle import ref_import as ri
from oracle import synth_weights as sw
from oracle import teacher_net as tn
from peppa_pig_face_landmark_amd import onnx_lite as ol
from peppa_pig_face_landmark_amd import weights as W
from tools import export_onnx_genuine as ex


def _x(n, size, seed):
    return torch.from_numpy(sw.smooth_blob_images(n, size, seed=seed).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()


def _assert_same_outputs(fwd, w_orig, w_got, x, tol_loc=1e-5, tol_score=2e-4):
    with torch.no_grad():
        a = fwd(ln.to_torch(w_orig), x)
        b = fwd(ln.to_torch(w_got), x)
    d_loc = float((a[0] - b[0]).abs().max())
    d_score = float((a[1] - b[1]).abs().max() / max(1.0, float(a[1].abs().max())))
    assert d_loc < tol_loc and d_score < tol_score, (d_loc, d_score)


@pytest.mark.skipif(not ri.available(), reason="reference checkout not present")
def test_student_export_of_the_reference_cotrain(tmp_path, student_weights):
    p = str(tmp_path / "kps_student.onnx")
    ex.export_cotrain(p, "student", student_weights)
    m = ol.read_model(p)
    convs = [n for n in m.nodes if n.op_type == "Conv"]
    assert len(convs) == 75 and any(n.inputs[1].startswith("onnx::Conv_") for n in convs)     # the exporter's anonymous fused tensors
    got = W.weights_from_onnx(p, "student")
    _assert_same_outputs(ln.student_forward, student_weights, got, _x(2, 256, 31))


@pytest.mark.skipif(not ri.available(), reason="reference checkout not present")
def test_teacher_export_of_the_reference_cotrain(tmp_path, student_weights):
    """convert_to_onnx.py:26-28 exports either model; the Teacher (HRNet-W18 encoder, 325 convolutions) imports too."""
    tw = sw.teacher_weights()
    p = str(tmp_path / "kps_teacher.onnx")
    ex.export_cotrain(p, "teacher", student_weights, tw)
    got = W.weights_from_onnx(p, "teacher")
    _assert_same_outputs(tn.teacher_forward, tw, got, _x(1, 256, 32))
    assert set(W.load_weights(p, "teacher")) == set(got)


def test_detector_export(tmp_path, detector_weights):
    p = str(tmp_path / "yolov5n-0.5.onnx")
    ex.export_detector(p, detector_weights)
    got = W.weights_from_onnx(p, "detector")
    x = _x(1, 640, 33)[:, :, :384]
    with torch.no_grad():
        a = dn.detector_forward(ln.to_torch(detector_weights), x)
        b = dn.detector_forward(ln.to_torch(got), x)
    assert a.shape == (1, 15120, 16)
    # the exporter folds BatchNorm into the convolutions in float32; through 30 layers that is ~1e-4 on the sigmoid scores
    assert float((a[..., 4] - b[..., 4]).abs().max()) < 1e-3                                   # objectness, in [0, 1]
    assert float((a - b).abs().max()) < 2e-4 * float(a.abs().max())                           # boxes / landmarks (pixels; random weights reach 1e4)


def test_oracle_only_export_has_the_same_wiring(tmp_path, student_weights):
    """Where the reference is absent the functional oracle goes through the same exporter: same Conv-to-Conv adjacency."""
    p = str(tmp_path / "kps_student_oracle.onnx")
    ex.export_oracle_landmark(p, student_weights, "student", size=128)
    got = W.weights_from_onnx(p, "student")
    assert np.array_equal(got["hm.weight"], student_weights["hm.weight"])


def test_reordered_convolutions_are_refused(tmp_path, detector_weights):
    """C3's cv1 and cv2 have identical shapes and the same input; only the wiring tells them apart (round-2 advisor finding).
    A file that lists them in the other order must fail the import instead of swapping their weights silently."""
    p = str(tmp_path / "det.onnx")
    ex.export_detector(p, detector_weights, (128, 160))
    m = ol.read_model(p)
    conv_idx = [i for i, n in enumerate(m.nodes) if n.op_type == "Conv"]
    from peppa_pig_face_landmark_amd.graph.detector import detector_param_shapes
    units, _ = W._conv_units(detector_param_shapes())
    names = [u[0] for u in units]
    i1, i2 = names.index("model.10.cv1.conv.weight"), names.index("model.10.cv2.conv.weight")
    assert units[i1][3] == units[i2][3]
    nodes = list(m.nodes)
    nodes[conv_idx[i1]], nodes[conv_idx[i2]] = nodes[conv_idx[i2]], nodes[conv_idx[i1]]
    p2 = str(tmp_path / "det_swapped.onnx")
    ol.write_model(p2, nodes, m.initializers, m.inputs, m.outputs)
    with pytest.raises(ValueError, match="orders .* its convolutions differently"):
        W.weights_from_onnx(p2, "detector")
    W.weights_from_onnx(p2, "detector", check_topology=False)      # ... which position-only pairing would have accepted
