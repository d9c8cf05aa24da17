"""yolov5n-0.5 detector program on the CPU SIMT emulator vs the oracle restatement (small input)."""
import numpy as np
import pytest
import torch

from oracle import detector_net as dn
from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd.graph.detector import build_detector_program, random_detector_weights


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_detector_f32_layers_and_rows(emu_engine, detector_weights, dtype):
    """f32s additionally runs the stride-1 ShuffleNetV2 units as ONE fused launch each (csrc/k_mbconv.h)."""
    H, W = 128, 160
    blob, info = build_detector_program(detector_weights, (H, W), dtype, keep_all=True)
    emu_engine.load_program(1, blob, 1)
    img = sw.smooth_blob_images(1, 160, seed=5)[:, :H]
    rows = emu_engine.detector_forward(img, info["rows"])
    Wt = {k: torch.from_numpy(v) for k, v in detector_weights.items()}
    x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    with torch.no_grad():
        ref = dn.detector_forward(Wt, x, taps).numpy()
    assert rows.shape == ref.shape == (1, 3 * (16 * 20 + 8 * 10 + 4 * 5), 16)
    for name, tid in info["tensors"].items():
        if name in taps:
            r = taps[name].permute(0, 2, 3, 1).numpy()
            g = emu_engine.read_tensor(1, tid, 1, r.shape[1:])
            assert np.abs(g - r).max() / (np.abs(r).max() + 1e-9) < 1e-4, name
    assert np.abs(rows - ref).max() / np.abs(ref).max() < 1e-4
    # float32 NCHW input seam (face_detector.py:65-69) gives the same rows
    rows_f = emu_engine.detector_forward(np.ascontiguousarray(x.numpy()), info["rows"])
    assert np.abs(rows_f - rows).max() / np.abs(ref).max() < 1e-4


def test_random_detector_weights_cover_the_inventory(detector_weights):
    rw = random_detector_weights()
    assert set(rw) == set(detector_weights)
    assert all(rw[k].shape == detector_weights[k].shape for k in rw)
