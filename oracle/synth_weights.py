"""Oracle: deterministic synthetic weights.  TEST INFRASTRUCTURE ONLY.

The reference's trained blobs (``Skps/pretrained/kps_student.onnx``, ``yolov5n-0.5.onnx``) are
absent from the checkout (``.MISSING_LARGE_BLOBS:1-2``) and there is no network, so the oracle
and the HIP engine are compared on *shared synthetic weights*:

* conv / linear tensors follow the reference's own ``weight_init`` rule
  (``TRAIN/face_landmark/lib/core/base_trainer/model.py:199-209``): kaiming-normal(fan_out, relu)
  for convs, xavier-normal for linears;
* conv biases where the reference has them (model.py:23-25,122,124,127,166-170,271) ~ N(0, 0.05);
* every BatchNorm gets gamma ~ U(0.5, 1.5), beta ~ N(0, 0.3) (so BN folding is really exercised),
  and -- like a trained network -- running_mean / running_var equal to the statistics of its
  own input on a calibration batch (float64 forward, rounded through float32, so every machine
  derives the same numbers).  Without this step 60 un-normalised layers drift far outside any
  realistic activation range and a reduced-precision comparison would mean nothing.

All randomness comes from ``numpy.random.Generator(PCG64(seed))`` drawn in inventory order,
which is stable across numpy versions and platforms.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch

SEED = 20260925


def smooth_blob_images(n: int, size: int, seed: int = 1235, noise: float = 6.0) -> np.ndarray:
    """``n`` uint8 HWC images made of a few Gaussian blobs per channel + mild noise
    (set B).  Deterministic in (n, size, seed)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    out = np.empty((n, size, size, 3), np.uint8)
    for i in range(n):
        img = np.full((size, size, 3), 40.0)
        for c in range(3):
            for _ in range(6):
                cx, cy = rng.uniform(0, size, 2)
                sig = rng.uniform(8.0, 40.0) * size / 256.0
                amp = rng.uniform(40.0, 200.0)
                img[:, :, c] += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig * sig))
        img += rng.normal(0.0, noise, img.shape)
        out[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def _draw(rng: np.random.Generator, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    if kind == "conv":
        cout, _, kh, kw = shape
        std = np.sqrt(2.0 / (cout * kh * kw))  # kaiming_normal_(mode='fan_out', relu)
        return (rng.standard_normal(shape) * std).astype(np.float32)
    if kind == "linear":
        fan_out, fan_in = shape
        std = np.sqrt(2.0 / (fan_in + fan_out))  # xavier_normal_
        return (rng.standard_normal(shape) * std).astype(np.float32)
    if kind == "bias":
        return (rng.standard_normal(shape) * 0.05).astype(np.float32)
    if kind == "bn_gamma":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if kind == "bn_beta":
        return (rng.standard_normal(shape) * 0.3).astype(np.float32)
    if kind == "bn_mean":
        return np.zeros(shape, np.float32)
    if kind == "bn_var":
        return np.ones(shape, np.float32)
    raise ValueError(kind)


def draw_inventory(inventory: Sequence[Tuple[str, Tuple[int, ...], str]], seed: int) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    return {name: _draw(rng, tuple(shape), kind) for name, shape, kind in inventory}


def calibrate_bn(weights: Dict[str, np.ndarray], forward: Callable, calib_input: torch.Tensor,
                 module) -> Dict[str, np.ndarray]:
    """Run ``forward(W, calib_input)`` once in float64 with ``module._CALIBRATING`` switched on;
    every BN records its input statistics as running_mean / running_var."""
    W = {k: torch.from_numpy(v).double() for k, v in weights.items()}
    module._CALIBRATING = []
    try:
        with torch.no_grad():
            forward(W, calib_input.double())
    finally:
        module._CALIBRATING = None
    return {k: v.float().numpy().copy() for k, v in W.items()}


def _cache_dir() -> str:
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache")
    os.makedirs(d, exist_ok=True)
    return d


def student_weights(seed: int = SEED, cache: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic weights for ``COTRAIN.student`` keyed by reference state_dict names."""
    from . import landmark_net as ln

    path = os.path.join(_cache_dir(), f"student_{seed}.npz")
    if cache and os.path.exists(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    w = draw_inventory(ln.param_inventory(), seed)
    calib = smooth_blob_images(8, 128, seed=seed + 1).astype(np.float32) / 255.0
    calib_t = torch.from_numpy(calib).permute(0, 3, 1, 2).contiguous()
    w = calibrate_bn(w, lambda W, x: ln.student_forward(W, x), calib_t, ln)
    if cache:
        np.savez(path, **w)
    return w


def detector_weights(seed: int = SEED + 7, cache: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic weights for the restated yolov5n-0.5 detector."""
    from . import detector_net as dn

    path = os.path.join(_cache_dir(), f"detector_{seed}.npz")
    if cache and os.path.exists(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    w = draw_inventory(dn.param_inventory(), seed)
    calib = smooth_blob_images(2, 640, seed=seed + 1)[:, :384].astype(np.float32) / 255.0
    calib_t = torch.from_numpy(calib).permute(0, 3, 1, 2).contiguous()
    w = calibrate_bn(w, lambda W, x: dn.detector_forward(W, x), calib_t, dn)
    if cache:
        np.savez(path, **w)
    return w


def teacher_weights(seed: int = SEED + 13, cache: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic weights for ``COTRAIN.teacher`` (HRNet-W18 encoder + decoder + hm head)."""
    from . import landmark_net as ln
    from . import teacher_net as tn

    path = os.path.join(_cache_dir(), f"teacher_{seed}.npz")
    if cache and os.path.exists(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    w = draw_inventory(tn.param_inventory(), seed)
    calib = smooth_blob_images(4, 128, seed=seed + 1).astype(np.float32) / 255.0
    calib_t = torch.from_numpy(calib).permute(0, 3, 1, 2).contiguous()
    W = {k: torch.from_numpy(v).double() for k, v in w.items()}
    ln._CALIBRATING = []
    tn._CALIBRATING = []
    try:
        with torch.no_grad():
            tn.teacher_forward(W, calib_t.double())
    finally:
        ln._CALIBRATING = None
        tn._CALIBRATING = None
    w = {k: v.float().numpy().copy() for k, v in W.items()}
    if cache:
        np.savez(path, **w)
    return w
