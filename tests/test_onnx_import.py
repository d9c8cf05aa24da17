"""The reference ships its networks only as ONNX files (Skps/config/Skps.yml:4,12).  The engine lifts the weights out of
such a file without onnx / onnxruntime / torch (peppa_pig_face_landmark_amd/onnx_lite.py + weights.weights_from_onnx).
The real blobs are absent from the checkout, so the tests write synthetic exports of the oracle's weights in both flavours
torch.onnx.export produces -- BatchNorm folded into the preceding Conv with the tensor names lost (the default), and
BatchNormalization nodes kept -- and check that the imported weights drive the ORACLE to the same outputs."""
import numpy as np
import pytest
import torch

from oracle import detector_net as dn
from oracle import landmark_net as ln
from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd import onnx_lite as ol
from peppa_pig_face_landmark_amd import weights as W
from peppa_pig_face_landmark_amd.graph.detector import detector_param_shapes
from peppa_pig_face_landmark_amd.graph.random_init import student_param_shapes


def _bn_of(w, p):
    return [np.asarray(w[f"{p}.{k}"], np.float64) for k in ("weight", "bias", "running_mean", "running_var")]


def write_synthetic_export(path, weights, shapes, eps, fold: bool, anonymous: bool = True):
    """Conv / BatchNormalization nodes in execution order, as torch.onnx.export(model.eval()) emits them."""
    units, lone = W._conv_units(shapes)
    nodes, inits = [], {}
    prev = "input"
    lone_iter = iter(lone)
    for i, (wname, bname, bnp, shape) in enumerate(units):
        wt = np.asarray(weights[wname], np.float64)
        cb = np.asarray(weights[bname], np.float64) if bname else None
        if bnp is not None and fold:
            g, beta, mean, var = _bn_of(weights, bnp)
            s = g / np.sqrt(var + eps)
            wt = wt * s.reshape(-1, 1, 1, 1)
            cb = ((cb if cb is not None else 0.0) - mean) * s + beta
        wn = f"onnx::Conv_{1000 + 2 * i}" if (anonymous and bnp is not None and fold) else "student." + wname
        ins = [prev, wn]
        inits[wn] = wt.astype(np.float32)
        if cb is not None:
            bn_name = f"onnx::Conv_{1001 + 2 * i}" if (anonymous and bnp is not None and fold) else "student." + wname[:-6] + "bias"
            inits[bn_name] = cb.astype(np.float32)
            ins.append(bn_name)
        out = f"conv_out_{i}"
        nodes.append(ol.Node("Conv", f"Conv_{i}", ins, [out], {"kernel_shape": [shape[2], shape[3]], "group": 1, "strides": [1, 1]}))
        prev = out
        if bnp is not None and not fold:
            names = [f"student.{bnp}.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
            for nme, v in zip(names, _bn_of(weights, bnp)):
                inits[nme] = v.astype(np.float32)
            nodes.append(ol.Node("BatchNormalization", f"BN_{i}", [prev] + names, [f"bn_out_{i}"], {"epsilon": float(eps), "momentum": 0.9}))
            prev = f"bn_out_{i}"
        nodes.append(ol.Node("Relu", f"Relu_{i}", [prev], [f"act_{i}"], {}))
        prev = f"act_{i}"
        if wname.endswith("aspp.fm_pool.pool.1.weight"):      # the BatchNorm over the ASPP concat: its input is NOT a conv
            bnp2, ch = next(lone_iter)
            nodes.append(ol.Node("Concat", "Concat_aspp", [prev, prev], ["cat_aspp"], {"axis": 1}))
            names = [f"student.{bnp2}.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
            for nme, v in zip(names, _bn_of(weights, bnp2)):
                inits[nme] = v.astype(np.float32)
            nodes.append(ol.Node("BatchNormalization", "BN_aspp", ["cat_aspp"] + names, ["bn_aspp"], {"epsilon": float(eps)}))
            prev = "bn_aspp"
    ol.write_model(path, nodes, inits, ["input"], [prev])


def test_wire_format_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    inits = {"a": rng.standard_normal((3, 4, 1, 1)).astype(np.float32), "idx": np.arange(5, dtype=np.int64),
             "d": rng.standard_normal(7).astype(np.float64)}
    nodes = [ol.Node("Conv", "c0", ["x", "a"], ["y"], {"kernel_shape": [1, 1], "group": 1, "auto_pad": "NOTSET", "alpha": 0.5}),
             ol.Node("Constant", "k", [], ["kout"], {"value": np.float32([1.5, -2.0])})]
    p = str(tmp_path / "m.onnx")
    ol.write_model(p, nodes, inits, ["x"], ["y"])
    m = ol.read_model(p)
    assert [n.op_type for n in m.nodes] == ["Conv", "Constant"] and m.inputs == ["x"] and m.outputs == ["y"]
    assert m.nodes[0].attrs["kernel_shape"] == [1, 1] and m.nodes[0].attrs["auto_pad"] == b"NOTSET" and m.nodes[0].attrs["alpha"] == 0.5
    for k, v in inits.items():
        assert m.initializers[k].dtype == v.dtype and np.array_equal(m.initializers[k], v)
    assert np.array_equal(m.initializers["kout"], np.float32([1.5, -2.0]))
    with pytest.raises(ValueError):
        ol.parse_model(b"\x0a\x03abc")          # a ModelProto without a graph


@pytest.mark.parametrize("fold", [True, False])
def test_student_weights_from_onnx(tmp_path, student_weights, fold):
    p = str(tmp_path / "kps_student.onnx")
    write_synthetic_export(p, student_weights, student_param_shapes(), 1e-5, fold)
    got = W.weights_from_onnx(p, "student", check_topology=False)
    assert set(got) == set(W._expected(student_param_shapes()))
    crops = sw.smooth_blob_images(1, 128, seed=77)
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        taps = {}
        ref = ln.student_forward(ln.to_torch(student_weights), x, taps)
        out = ln.student_forward(ln.to_torch(got), x)
    hm = taps["hm"].numpy()[:, :98].reshape(1, 98, -1)
    part = np.partition(hm, -2, axis=2)
    safe = (part[:, :, -1] - part[:, :, -2]) > 1e-3
    d = np.abs(out[0].numpy() - ref[0].numpy()).reshape(1, 98, 2).max(2)
    assert d[safe].max() < 1e-5 and np.abs(out[1].numpy() - ref[1].numpy())[safe].max() < 1e-3
    if not fold:   # un-folded export: every tensor comes back bit-identical
        for k in got:
            assert np.allclose(got[k], student_weights[k], rtol=0, atol=1e-7), k


def test_detector_weights_from_onnx(tmp_path, detector_weights):
    p = str(tmp_path / "yolov5n-0.5.onnx")
    write_synthetic_export(p, detector_weights, detector_param_shapes(), 1e-3, True)
    got = W.weights_from_onnx(p, "detector", check_topology=False)
    x = torch.from_numpy(np.random.default_rng(3).uniform(0, 1, (1, 3, 96, 160)).astype(np.float32))
    with torch.no_grad():
        ref = dn.detector_forward(ln.to_torch(detector_weights), x)
        out = dn.detector_forward(ln.to_torch(got), x)
    assert float((out - ref).abs().max()) < 2e-3 * float(ref.abs().max())


def test_wrong_architecture_is_rejected(tmp_path, detector_weights, student_weights):
    p = str(tmp_path / "det.onnx")
    write_synthetic_export(p, detector_weights, detector_param_shapes(), 1e-3, True)
    with pytest.raises(ValueError, match="Conv nodes"):
        W.weights_from_onnx(p, "student", check_topology=False)
    shapes = student_param_shapes()
    w2 = dict(student_weights)
    w2["encoder.blocks.2.0.conv_dw.weight"] = np.zeros((72, 1, 3, 3), np.float32)      # 5x5 in the real architecture
    shapes = [(n, ((72, 1, 3, 3) if n == "encoder.blocks.2.0.conv_dw.weight" else s), k) for n, s, k in shapes]
    p2 = str(tmp_path / "bad.onnx")
    write_synthetic_export(p2, w2, shapes, 1e-5, True)
    with pytest.raises(ValueError, match="encoder.blocks.2.0.conv_dw.weight"):
        W.weights_from_onnx(p2, "student", check_topology=False)


def test_load_weights_dispatch(tmp_path, student_weights):
    p = str(tmp_path / "kps.npz")
    np.savez(p, **student_weights)
    got = W.load_weights(p, "keypoints")
    assert set(got) == set(student_weights)
    sd = {"student." + k: torch.from_numpy(np.asarray(v)) for k, v in student_weights.items()}
    pt = str(tmp_path / "cotrain.pth")
    torch.save(sd, pt)
    got = W.load_weights(pt, "keypoints")           # torch.load(..., weights_only=True)
    assert np.array_equal(got["hm.weight"], student_weights["hm.weight"])
    with pytest.raises(ValueError):
        W.load_weights(str(tmp_path / "x.bin"), "detector")


def _write_both(tmp_path, student_weights, detector_weights):
    """The two files of Skps.yml as PyTorch's own exporter writes them (tools/export_onnx_genuine.py): the importer's
    wiring check (weights.conv_topology) is on for everything a FaceAna user loads."""
    from oracle import ref_import as ri
    from tools import export_onnx_genuine as ex
    ps, pd = str(tmp_path / "kps_student.onnx"), str(tmp_path / "yolov5n-0.5.onnx")
    if ri.available():
        ex.export_cotrain(ps, "student", student_weights)
    else:
        ex.export_oracle_landmark(ps, student_weights)
    ex.export_detector(pd, detector_weights)
    return ps, pd


def test_programs_build_from_onnx_without_torch(tmp_path, student_weights, detector_weights):
    """The drop-in path of a reference user: .onnx files -> packed HIP programs with torch / onnx / onnxruntime unimportable."""
    import subprocess
    import sys
    ps, pd = _write_both(tmp_path, student_weights, detector_weights)
    code = (
        "import sys\n"
        "for m in ('torch', 'onnx', 'onnxruntime', 'cv2'): sys.modules[m] = None\n"
        "from peppa_pig_face_landmark_amd.core.api.facer import _load_weights\n"
        "from peppa_pig_face_landmark_amd.graph.student import build_student_program\n"
        "from peppa_pig_face_landmark_amd.graph.detector import build_detector_program\n"
        f"k = _load_weights('/', {ps!r}, 'keypoints'); d = _load_weights('/', {pd!r}, 'detector')\n"
        "b1, _ = build_student_program(k, 256, 'f32s'); b2, _ = build_detector_program(d, (384, 640), 'f32s')\n"
        "print(len(b1), len(b2))\n")
    import os
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    n1, n2 = (int(v) for v in r.stdout.split())
    assert n1 > 5_000_000 and n2 > 500_000


def test_faceana_constructs_and_runs_from_onnx_paths(tmp_path, emu_library, student_weights, detector_weights):
    """Skps.yml with the reference's own model_path values (.onnx): FaceAna() loads them and run() works (CPU tier: the
    SIMT-emulator build of the engine, small network input sizes to keep it quick)."""
    from Skps import FaceAna
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    from peppa_pig_face_landmark_amd.synth import make_frame
    ps, pd = _write_both(tmp_path, student_weights, detector_weights)
    cfg = get_cfg()
    assert cfg["Skps"]["Detect"]["model_path"].endswith("yolov5n-0.5.onnx") and cfg["Skps"]["Keypoints"]["model_path"].endswith("kps_student.onnx")
    cfg["Skps"]["Detect"]["model_path"], cfg["Skps"]["Keypoints"]["model_path"] = pd, ps
    cfg["Skps"]["Detect"]["input_shape"] = [96, 160, 3]
    cfg["Skps"]["Keypoints"]["input_shape"] = [64, 64, 3]
    cfg["Skps"]["Engine"]["dtype"] = "f32"
    facer = FaceAna(cfg=cfg, library=emu_library)
    frame, boxes = make_frame(270, 480, 1, seed=2)
    lm, states = facer.face_landmark(frame, boxes)
    assert lm.shape == (1, 98, 2) and states.shape == (1, 98) and np.isfinite(lm).all()
    res = facer.run(frame)
    assert isinstance(res, list)
    facer.engine.close()
