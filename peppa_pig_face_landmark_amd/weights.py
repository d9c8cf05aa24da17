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


# ---------------------------------------------------------------------------------------------------------------------
# ONNX files -- the only weight format the reference ships (Skps/config/Skps.yml:4,12: pretrained/yolov5n-0.5.onnx,
# pretrained/kps_student.onnx; exported by TRAIN/face_landmark/tools/convert_to_onnx.py:54-61 resp. yolov5-face's
# export.py).  torch.onnx.export in eval mode folds every BatchNorm that directly follows a Conv into that Conv and
# renames the fused tensors ("onnx::Conv_1234"), so names cannot be trusted; what survives is the ORDER: Conv nodes appear
# in the order the convolutions execute, which is the order of the architecture inventories
# (graph/random_init.py::student_param_shapes, graph/detector.py::detector_param_shapes).  Every node is checked against
# the inventory's shape before it is accepted, so a different architecture fails here with the offending layer named.
_BN_EPS = {"student": 1e-5, "teacher": 1e-5, "detector": 1e-3}


def _conv_units(shapes):
    """[(conv weight name, conv bias name or None, fused BN prefix or None, weight shape)] and the stand-alone BNs
    [(prefix, channels)], both in execution order."""
    units, lone_bn = [], []
    for name, shape, kind in shapes:
        if kind == "conv":
            units.append([name, None, None, tuple(shape)])
        elif kind == "bias":
            assert units and name == units[-1][0][:-len("weight")] + "bias", name
            units[-1][1] = name
        else:
            if units and units[-1][2] is None and shape[0] == units[-1][3][0]:
                units[-1][2] = name
            else:
                lone_bn.append((name, shape[0]))
    return units, lone_bn


def conv_topology(model) -> list:
    """For every Conv node, in file order: the sorted indices of the Conv nodes whose outputs reach its DATA input through
    non-Conv nodes only (-1 = the graph input).  This Conv-to-Conv adjacency is a property of the architecture -- BatchNorm
    folded or kept, Clip/Mul or HardSwish, Resize or Upsample make no difference -- so it pins WHICH convolution a node
    is, where the node order alone cannot: C3's cv1 / cv2 or a ShuffleNetV2 unit's two 1x1s have identical shapes and
    differ only in what feeds them and what they feed."""
    producer = {o: n for n in model.nodes for o in n.outputs}
    index = {id(n): i for i, n in enumerate(x for x in model.nodes if x.op_type == "Conv")}
    memo: Dict[str, frozenset] = {}

    def sources(name: str) -> frozenset:
        if name in memo:
            return memo[name]
        memo[name] = frozenset()                 # cycle guard (ONNX graphs are acyclic; a malformed file must not hang us)
        n = producer.get(name)
        if n is None:
            r = frozenset([-1]) if (name in model.inputs and name not in model.initializers) else frozenset()
        elif n.op_type == "Conv":
            r = frozenset([index[id(n)]])
        else:
            r = frozenset().union(*[sources(i) for i in n.inputs if i]) if n.inputs else frozenset()
        memo[name] = r
        return r

    import sys
    limit = sys.getrecursionlimit()
    sys.setrecursionlimit(max(limit, 20000))
    try:
        return [sorted(sources(n.inputs[0])) for n in model.nodes if n.op_type == "Conv"]
    finally:
        sys.setrecursionlimit(limit)


def _check_topology(path: str, arch: str, model) -> None:
    import json
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph", "onnx_topology.json")
    with open(ref) as f:
        want = json.load(f)[arch]
    got = conv_topology(model)
    for i, (a, b) in enumerate(zip(got, want)):
        if list(a) != list(b):
            raise ValueError("%s: Conv node #%d is fed by convolutions %s where the %s architecture has %s there -- the file "
                             "orders (or wires) its convolutions differently from the exporter the importer was written "
                             "against; refusing to pair weights by position" % (path, i, a, arch, b))


def weights_from_onnx(path: str, arch: str, check_topology: bool = True) -> Dict[str, np.ndarray]:
    """Weights of ``arch`` ('student' | 'detector') lifted out of an ONNX export, keyed by the reference's state_dict
    names -- i.e. exactly what ``build_student_program`` / ``build_detector_program`` consume.  Handles both export
    flavours: BatchNorm folded into the convolutions (the default; the fused bias is carried by an identity BatchNorm in
    the result) and BatchNormalization nodes kept.  Nodes are paired with the architecture by execution order, shape-checked
    one by one, and the file's Conv-to-Conv wiring must equal the architecture's (``conv_topology``)."""
    from . import onnx_lite
    if arch == "student":
        from .graph.random_init import student_param_shapes as shapes_fn
    elif arch == "teacher":          # convert_to_onnx.py:26-28 exports either model of COTRAIN
        from .graph.teacher import teacher_param_shapes as shapes_fn
    elif arch == "detector":
        from .graph.detector import detector_param_shapes as shapes_fn
    else:
        raise ValueError("arch must be 'student', 'teacher' or 'detector'")
    eps = _BN_EPS[arch]
    shapes = list(shapes_fn())
    dead = []
    if arch == "teacher":      # convolutions no output depends on are absent from an export (graph/teacher.py DEAD_FUSE_PREFIX)
        from .graph.teacher import DEAD_FUSE_PREFIX
        dead = [e for e in shapes if e[0].startswith(DEAD_FUSE_PREFIX)]
        shapes = [e for e in shapes if not e[0].startswith(DEAD_FUSE_PREFIX)]
    units, lone_bn = _conv_units(shapes)
    model = onnx_lite.read_model(path)
    convs = [n for n in model.nodes if n.op_type == "Conv"]
    bns = [n for n in model.nodes if n.op_type == "BatchNormalization"]
    if len(convs) != len(units):
        raise ValueError("%s: %d Conv nodes, the %s architecture has %d convolutions" % (path, len(convs), arch, len(units)))
    if check_topology:       # off only for the hand-written single-chain files of tests/test_onnx_import.py
        _check_topology(path, arch, model)
    producer = {o: n for n in model.nodes for o in n.outputs}
    conv_unit = {id(n): u for n, u in zip(convs, units)}

    def init(name: str, what: str) -> np.ndarray:
        if name not in model.initializers:
            raise ValueError("%s: %s %r is not a constant of the graph" % (path, what, name))
        return np.ascontiguousarray(np.asarray(model.initializers[name], np.float32))

    out: Dict[str, np.ndarray] = {}
    fused_bias: Dict[str, np.ndarray] = {}
    for n, (wname, bname, bnp, shape) in zip(convs, units):
        w = init(n.inputs[1], "weight of Conv node %s," % (n.name or n.outputs[0]))
        if tuple(w.shape) != shape:
            raise ValueError("%s: Conv node %s has weight %s where %s needs %s" % (path, n.name or n.outputs[0], tuple(w.shape), wname, shape))
        out[wname] = w
        b = init(n.inputs[2], "bias") if len(n.inputs) > 2 and n.inputs[2] else None
        if b is not None and b.shape != (shape[0],):
            raise ValueError("%s: bias of %s has shape %s" % (path, wname, b.shape))
        if bnp is not None:
            fused_bias[bnp] = b if b is not None else np.zeros(shape[0], np.float32)
            if bname is not None:
                out[bname] = np.zeros(shape[0], np.float32)       # conv bias + BN both folded into the ONNX bias
        elif bname is not None:
            if b is None:
                raise ValueError("%s: %s has no bias in the ONNX graph" % (path, wname))
            out[bname] = b
        elif b is not None and np.any(b != 0):
            raise ValueError("%s: %s carries a bias the architecture has no place for" % (path, wname))
    lone_nodes = []
    for n in bns:
        src = producer.get(n.inputs[0])
        u = conv_unit.get(id(src)) if src is not None else None
        vals = [init(n.inputs[k], "BatchNormalization input") for k in range(1, 5)]
        node_eps = float(n.attrs.get("epsilon", 1e-5))
        if u is not None and u[2] is not None and u[2] in fused_bias:
            # un-folded export: the conv kept its own bias, this node is the BatchNorm behind it
            g, beta, mean, var = vals
            bnp = u[2]
            if u[1] is not None:
                out[u[1]] = fused_bias[bnp]
            elif np.any(fused_bias[bnp] != 0):
                raise ValueError("%s: %s has a conv bias AND a BatchNorm but no bias slot" % (path, u[0]))
            del fused_bias[bnp]
            out[bnp + ".weight"], out[bnp + ".bias"], out[bnp + ".running_mean"] = g, beta, mean
            out[bnp + ".running_var"] = (var.astype(np.float64) + node_eps - eps).astype(np.float32)   # re-expressed for the graph builder's eps
        else:
            lone_nodes.append((n, vals, node_eps))
    if len(lone_nodes) != len(lone_bn):
        raise ValueError("%s: %d stand-alone BatchNormalization nodes, the %s architecture has %d" % (path, len(lone_nodes), arch, len(lone_bn)))
    for (n, (g, beta, mean, var), node_eps), (bnp, ch) in zip(lone_nodes, lone_bn):
        if g.shape != (ch,):
            raise ValueError("%s: BatchNormalization %s has %d channels, %s needs %d" % (path, n.name, g.shape[0], bnp, ch))
        out[bnp + ".weight"], out[bnp + ".bias"], out[bnp + ".running_mean"] = g, beta, mean
        out[bnp + ".running_var"] = (var.astype(np.float64) + node_eps - eps).astype(np.float32)
    for bnp, b in fused_bias.items():       # folded export: identity BatchNorm that carries the fused bias
        ch = b.shape[0]
        out[bnp + ".weight"] = np.ones(ch, np.float32)
        out[bnp + ".bias"] = b
        out[bnp + ".running_mean"] = np.zeros(ch, np.float32)
        out[bnp + ".running_var"] = np.full(ch, 1.0 - eps, np.float32)
    for name, shape, kind in dead:                # never executed: any finite value will do, the inventory stays complete
        if kind == "bn":
            out[name + ".weight"], out[name + ".bias"] = np.ones(shape, np.float32), np.zeros(shape, np.float32)
            out[name + ".running_mean"], out[name + ".running_var"] = np.zeros(shape, np.float32), np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    exp = _expected(shapes_fn())
    missing = sorted(set(exp) - set(out))
    if missing:
        raise ValueError("%s: could not recover %d tensors (first: %s)" % (path, len(missing), missing[:3]))
    return out


def load_weights(path: str, arch: str) -> Dict[str, np.ndarray]:
    """One entry point for every weight file ``FaceAna`` accepts as ``model_path``: ``.onnx`` (the reference's own
    files), ``.npz`` (arrays keyed by state_dict names) or a torch checkpoint ``.pth/.pt`` (needs torch; loaded with
    ``weights_only=True``)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".onnx":
        return weights_from_onnx(path, "student" if arch == "keypoints" else arch)     # 'teacher' and 'detector' pass through
    if ext == ".npz":
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if ext in (".pth", ".pt", ".ckpt"):
        import torch
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and not any(str(k).startswith(("student.", "module.", "model.")) for k in sd):
            sd = sd["state_dict"]
        if arch in ("student", "keypoints", "teacher"):
            student, teacher = split_cotrain_state_dict(sd)
            return teacher if arch == "teacher" else student
        return detector_state_dict(sd)
    raise ValueError("unsupported weight file %r (expected .onnx, .npz, .pth or .pt)" % path)


def detector_state_dict(sd: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """yolov5-face ``model.state_dict()`` (keys ``model.N....``, optionally behind ``module.``) -> the detector's
    weight dictionary, presence- and shape-checked against ``graph/detector.py::detector_param_shapes``."""
    from .graph.detector import detector_param_shapes
    exp = _expected(detector_param_shapes())
    flat = {}
    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        flat[k] = v
    missing = sorted(set(exp) - set(flat))
    if missing:
        raise ValueError("detector checkpoint lacks %d tensors (first: %s)" % (len(missing), missing[:3]))
    out = {}
    for name, shape in exp.items():
        arr = _to_numpy(flat[name])
        if tuple(arr.shape) != shape:
            raise ValueError("%s has shape %s, the architecture needs %s" % (name, tuple(arr.shape), shape))
        out[name] = arr
    return out


def import_checkpoint(path: str, out_dir: str) -> Dict[str, str]:
    """``torch.load`` the checkpoint at ``path`` and write ``kps_student.npz`` (and ``kps_teacher.npz``) to out_dir."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
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
