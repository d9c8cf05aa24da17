"""Detector parity per tapped tensor against the oracle for one or more builds of the library (run on the GPU box; test
infrastructure: uses oracle/).  usage: python tools/det_parity.py <lib.so> [<lib.so> ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import detector_net as dn                                    # noqa: E402
from oracle import synth_weights as sw                                   # noqa: E402
from peppa_pig_face_landmark_amd import _native                           # noqa: E402
from peppa_pig_face_landmark_amd.graph.detector import build_detector_program   # noqa: E402
w = sw.detector_weights()
img = sw.smooth_blob_images(2, 640, seed=9)[:, :384]
W = {k: torch.from_numpy(v) for k, v in w.items()}
x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
taps = {}
with torch.no_grad():
    ref = dn.detector_forward(W, x, taps).numpy()
for dtype in ("f32s",):
    blob, info = build_detector_program(w, (384, 640), dtype, keep_all=True)
    for lib in sys.argv[1:]:
        eng = _native.Engine(0, os.path.abspath(lib))
        eng.load_program(1, blob, 2)
        rows = eng.detector_forward(img, 15120)
        worst = []
        for name, tid in info["tensors"].items():
            if name in taps:
                r = taps[name].permute(0, 2, 3, 1).numpy()
                g = eng.read_tensor(1, tid, 2, r.shape[1:])
                worst.append((float(np.abs(g - r).max() / (np.abs(r).max() + 1e-9)), name))
        eng.close()
        print(os.path.basename(lib), " ".join("%s=%.1e" % (n, e) for e, n in worst))
        worst.sort(reverse=True)
        print("%-50s rows %.3e | %s" % (os.path.basename(lib), float(np.abs(rows - ref).max() / np.abs(ref).max()),
                                        "  ".join("%s %.2e" % (n, e) for e, n in worst[:5])))
