"""tools/eval_wflw.py (the engine-side twin of the reference's eval_WFLW.py:96-142) on a synthetic miniature WFLW tree:
the per-class NME must equal what the reference's protocol gives with the oracle as the network."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import landmark_net as ln
from oracle import prepost as pp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
spec = importlib.util.spec_from_file_location("eval_wflw", os.path.join(ROOT, "tools", "eval_wflw.py"))
ev = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ev)


def _make_dataset(root, n_per_class=3):
    from PIL import Image
    rng = np.random.default_rng(12)
    os.makedirs(os.path.join(root, "WFLW_images", "0--Parade"), exist_ok=True)
    os.makedirs(os.path.join(root, "WFLW_annotations", "list_98pt_test"), exist_ok=True)
    for cls in ("test", "largepose"):
        lines = []
        for i in range(n_per_class):
            h, w = int(rng.integers(200, 320)), int(rng.integers(260, 400))
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            name = "0--Parade/%s_%d.png" % (cls, i)                     # PNG: lossless, so every reader sees the same pixels
            Image.fromarray(img[:, :, ::-1]).save(os.path.join(root, "WFLW_images", name))
            cx, cy, r = rng.uniform(0.4, 0.6) * w, rng.uniform(0.4, 0.6) * h, rng.uniform(40, 70)
            pts = np.stack([cx + r * rng.uniform(-1, 1, 98), cy + r * rng.uniform(-1, 1, 98)], 1).astype(np.float32)
            lines.append(" ".join("%.4f" % v for v in pts.reshape(-1)) + " 0 0 0 0 0 0 0 0 0 0 " + name + "\n")
        with open(os.path.join(root, "WFLW_annotations", "list_98pt_test", "list_98pt_%s.txt" % cls), "w") as f:
            f.writelines(lines)


def _reference_protocol(root, weights, size):
    """eval_WFLW.py:96-142 restated with the oracle network and the oracle's cv2 stand-ins (checker)."""
    from PIL import Image
    W = ln.to_torch(weights)
    out = {}
    for cls, lines in ev.load_test_lists(root).items():
        scores = []
        for ln_ in lines:
            dp = ln_.split()
            kps = np.array(dp[:196], np.float32).reshape(-1, 2)
            img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(root, "WFLW_images", dp[-1])).convert("RGB"))[:, :, ::-1])
            crop, label = ev.crop_for_eval(img, kps)
            h, w = crop.shape[:2]
            label[:, 0] /= w
            label[:, 1] /= h
            x = pp.landmark_input(pp.resize_linear_u8(np.ascontiguousarray(crop), size, size))
            with torch.no_grad():
                loc = ln.student_forward(W, torch.from_numpy(x))[0].numpy()[0]
            scores.append(ev.nme(label, loc[:196]))
        out[cls] = float(np.mean(scores))
    return out


def _check(library, weights, tmp_path, size):
    root = str(tmp_path / "WFLW")
    _make_dataset(root)
    got = ev.evaluate(root, weights, "student", size, batch=4, dtype="f32", library=library, log=lambda *_: None)
    ref = _reference_protocol(root, weights, size)
    assert set(got) == {"test", "largepose"} == set(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) < 2e-3 * max(1.0, abs(ref[k])), (k, got[k], ref[k])


def test_eval_wflw_emulator(emu_library, student_weights, tmp_path):
    _check(emu_library, student_weights, tmp_path, 64)


@pytest.mark.gpu
def test_eval_wflw_gpu(hip_library, student_weights, tmp_path):
    _check(hip_library, student_weights, tmp_path, 256)


def test_resize_matches_oracle_bit_exact(emu_engine):
    rng = np.random.default_rng(3)
    for (h, w, oh, ow) in ((60, 80, 64, 64), (128, 96, 64, 48), (50, 31, 64, 64), (71, 133, 32, 40)):
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(emu_engine.resize(src, (oh, ow)), pp.resize_linear_u8(src, ow, oh)), (h, w, oh, ow)
