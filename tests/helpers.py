"""Shared helpers for the parity tests (oracle side)."""
import numpy as np
import torch

from oracle import landmark_net as ln
from oracle import synth_weights as sw


def oracle_student(weights, crops_u8, dtype=torch.float32, want_taps=True):
    """Run the oracle on uint8 NHWC crops exactly like face_landmark.py:44-48 feeds the ONNX."""
    W = ln.to_torch(weights, dtype)
    x = torch.from_numpy(crops_u8.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous().to(dtype)
    taps = {} if want_taps else None
    with torch.no_grad():
        loc, score = ln.student_forward(W, x, taps)
    return loc.float().numpy(), score.float().numpy(), taps


def heat_margins(taps):
    """top1 - top2 of every score heat-map [B,98] (how close the arg-max is to flipping)."""
    hm = taps["hm"].float().numpy()
    b = hm.shape[0]
    flat = hm[:, :98].reshape(b, 98, -1)
    part = np.partition(flat, -2, axis=2)
    return part[:, :, -1] - part[:, :, -2]


def tap_nhwc(taps, name):
    return taps[name].float().permute(0, 2, 3, 1).numpy()


def read_engine_tensor(eng, slot, info, name, batch, ref_shape_hwc, ve):
    h, w, c = ref_shape_hwc
    cpad = (c + ve - 1) // ve * ve
    got = eng.read_tensor(slot, info["tensors"][name], batch, (h, w, cpad))
    return got[..., :c]
