#!/usr/bin/env python
"""Pin the THIRD-PARTY halves of the oracle against the real libraries, wherever they can be imported.

The reference delegates arithmetic to packages that are neither vendored in it nor installed in the build container:
OpenCV (cv2.resize / copyMakeBorder: Skps/core/api/face_detector.py:53,63, face_landmark.py:79,97), timm 0.6.11
(mobilenetv3_large_100 / hrnet_w18 encoders: TRAIN/face_landmark/lib/core/base_trainer/model.py:252-258,306-311) and
onnxruntime on the two shipped blobs (Skps/core/api/onnx_model_base.py:14,23-24).  The oracle RESTATES them
(oracle/prepost.py, oracle/landmark_net.py, oracle/teacher_net.py, oracle/detector_net.py) and says "parity unpinned".
Run this script on any machine that has one of those packages:

    python tools/pin_third_party.py [--student-onnx kps_student.onnx] [--detector-onnx yolov5n-0.5.onnx]

For every package found it compares the restatement with the real thing on seeded inputs, prints the worst difference, and
writes a small golden file under tests/golden/ (third_party_cv2.npz, third_party_timm_mobilenetv3.npz,
third_party_timm_hrnet.npz, third_party_ort_student.npz, third_party_ort_detector.npz).  tests/test_third_party_pins.py
picks up whichever goldens exist, so committing them flips that segment from "unpinned" to pinned on every box.
Exit status: 0 if everything that could be checked agreed, 1 otherwise, 2 if nothing could be checked.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

CV2_CASES = [  # (src_h, src_w, dst_w, dst_h): shrink, enlarge, exact 2x (box-average branch), letterbox-like, crop-like
    (60, 80, 37, 23), (60, 80, 160, 120), (64, 96, 48, 32), (273, 410, 576, 384), (108, 192, 64, 36),
    (70, 70, 64, 64), (128, 128, 64, 64), (50, 31, 64, 64),
]


def make_cv2_inputs():
    rng = np.random.default_rng(20260925)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w, _, _ in CV2_CASES]


def pin_cv2():
    import cv2
    from oracle import prepost as pp
    srcs, outs, worst, bad = make_cv2_inputs(), {}, 0, 0
    for i, (src, (h, w, dw, dh)) in enumerate(zip(srcs, CV2_CASES)):
        real = cv2.resize(src, (dw, dh))                      # default interpolation = INTER_LINEAR, as the reference calls it
        mine = pp.resize_linear_u8(src, dw, dh)
        d = int(np.abs(real.astype(np.int32) - mine.astype(np.int32)).max())
        worst, bad = max(worst, d), bad + int(d != 0)
        outs[f"resize_{i}"] = real
        print("cv2.resize %dx%d -> %dx%d : max |cv2 - oracle| = %d LSB" % (w, h, dw, dh, d))
    b = cv2.copyMakeBorder(srcs[0], 3, 5, 7, 2, cv2.BORDER_CONSTANT, value=114)
    m = pp.pad_constant(srcs[0], 3, 5, 7, 2, 114)
    bad += int(not np.array_equal(b, m))
    outs["border_0"] = b
    np.savez_compressed(os.path.join(GOLD, "third_party_cv2.npz"), version=np.bytes_(cv2.__version__.encode()), **outs)
    print("cv2 %s: %s (golden written)" % (cv2.__version__, "bit-exact" if bad == 0 else "%d cases differ, worst %d LSB" % (bad, worst)))
    return bad == 0


def _load_into_timm(model, weights, prefix="encoder."):
    import torch
    sd = model.state_dict()
    missing = []
    for k in sd:
        if k.endswith("num_batches_tracked"):
            continue
        src = prefix + k
        if src not in weights:
            missing.append(k)
            continue
        assert tuple(sd[k].shape) == tuple(weights[src].shape), (k, tuple(sd[k].shape), weights[src].shape)
        sd[k] = torch.from_numpy(np.asarray(weights[src]))
    model.load_state_dict(sd)
    return missing


def pin_timm_mobilenetv3():
    import timm
    import torch
    from oracle import landmark_net as ln
    from oracle import synth_weights as sw
    w = sw.student_weights()
    model = timm.create_model("mobilenetv3_large_100", pretrained=False, features_only=True, out_indices=[0, 1, 2, 4],
                              in_chans=3, output_stride=16)          # model.py:252-258
    model.blocks[6] = torch.nn.Identity()                            # model.py:262
    missing = _load_into_timm(model, w)
    model.eval()
    x = torch.from_numpy(sw.smooth_blob_images(2, 64, seed=4242).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        real = model(x)
        mine = ln.encoder_forward(ln.to_torch(w), x)
    worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(real, mine))
    np.savez_compressed(os.path.join(GOLD, "third_party_timm_mobilenetv3.npz"), version=np.bytes_(timm.__version__.encode()),
                        seed=4242, size=64, **{f"feat{i}": f.numpy() for i, f in enumerate(real)})
    print("timm %s mobilenetv3_large_100: worst relative feature difference %.2e (%d timm tensors not in the oracle inventory: %s)"
          % (timm.__version__, worst, len(missing), missing[:3]))
    return worst < 1e-5 and not missing


def pin_timm_hrnet():
    import timm
    import torch
    from oracle import landmark_net as ln
    from oracle import synth_weights as sw
    from oracle import teacher_net as tn
    w = sw.teacher_weights()
    model = timm.create_model("hrnet_w18", pretrained=False, features_only=True, out_indices=[0, 1, 2, 3], in_chans=3)   # model.py:306-311
    missing = _load_into_timm(model, w)
    model.eval()
    x = torch.from_numpy(sw.smooth_blob_images(1, 64, seed=4343).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        real = model(x)
        mine = tn.encoder_forward(ln.to_torch(w), x)
    worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(real, mine))
    np.savez_compressed(os.path.join(GOLD, "third_party_timm_hrnet.npz"), version=np.bytes_(timm.__version__.encode()),
                        seed=4343, size=64, **{f"feat{i}": f.numpy() for i, f in enumerate(real)})
    print("timm %s hrnet_w18: worst relative feature difference %.2e (%d timm tensors not in the oracle inventory)"
          % (timm.__version__, worst, len(missing)))
    return worst < 1e-5


def pin_ort(path, arch):
    """The real blob: onnxruntime's outputs against the oracle driven by the weights OUR importer lifts out of the file."""
    import onnxruntime as rt
    import torch
    from oracle import detector_net as dn
    from oracle import landmark_net as ln
    from oracle import synth_weights as sw
    from peppa_pig_face_landmark_amd.weights import weights_from_onnx
    w = ln.to_torch(weights_from_onnx(path, arch))
    sess = rt.InferenceSession(path, providers=["CPUExecutionProvider"])
    name = sess.get_inputs()[0].name
    if arch == "student":
        size = int(sess.get_inputs()[0].shape[2])
        x = (sw.smooth_blob_images(2, size, seed=4444).astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)
        real = [sess.run([], {name: x[i:i + 1]}) for i in range(2)]             # the reference feeds batch 1 (face_landmark.py:48)
        with torch.no_grad():
            loc, score = ln.student_forward(w, torch.from_numpy(np.ascontiguousarray(x)))[:2]
        rl = np.concatenate([r[0].reshape(1, -1)[:, :196] for r in real])
        worst = float(np.abs(rl - loc.numpy()).max())
        np.savez_compressed(os.path.join(GOLD, "third_party_ort_student.npz"), seed=4444, size=size, landmark=rl,
                            score=np.concatenate([r[1].reshape(1, -1) for r in real]))
        print("onnxruntime %s on %s: max |landmark - oracle| = %.2e (north-star tolerance 1e-3)" % (rt.__version__, os.path.basename(path), worst))
        return worst < 1e-3
    x = np.random.default_rng(4545).uniform(0, 1, (1, 3, 384, 640)).astype(np.float32)
    real = sess.run([], {name: x})[0].reshape(-1, 16)
    with torch.no_grad():
        mine = dn.detector_forward(w, torch.from_numpy(x))[0].numpy()
    worst = float(np.abs(real - mine).max() / (np.abs(real).max() + 1e-12))
    np.savez_compressed(os.path.join(GOLD, "third_party_ort_detector.npz"), seed=4545, rows=real[::16].copy())
    print("onnxruntime %s on %s: worst relative row difference %.2e" % (rt.__version__, os.path.basename(path), worst))
    return worst < 1e-4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--student-onnx", default=os.environ.get("PEPPA_REAL_STUDENT_ONNX", ""))
    ap.add_argument("--detector-onnx", default=os.environ.get("PEPPA_REAL_DETECTOR_ONNX", ""))
    args = ap.parse_args()
    checks = [("cv2", pin_cv2), ("timm mobilenetv3", pin_timm_mobilenetv3), ("timm hrnet", pin_timm_hrnet)]
    if args.student_onnx:
        checks.append(("onnxruntime student", lambda: pin_ort(args.student_onnx, "student")))
    if args.detector_onnx:
        checks.append(("onnxruntime detector", lambda: pin_ort(args.detector_onnx, "detector")))
    ran, ok = 0, True
    for what, fn in checks:
        try:
            good = fn()
        except ImportError as e:
            print("skip %-22s (%s)" % (what, e))
            continue
        ran += 1
        ok = ok and good
    if ran == 0:
        print("nothing to pin on this machine: cv2, timm and onnxruntime are all missing")
        return 2
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
