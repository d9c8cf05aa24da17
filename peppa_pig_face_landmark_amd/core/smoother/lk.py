"""Temporal smoothing of landmarks / boxes -- host-side, 98x2 floats per face.
Behaviour follows the reference's Skps/core/smoother/lk.py: GroupTrack.calculate :19-56 (match
faces between frames by the IoU of their landmark hulls, then a One-Euro style filter in
image-normalised coordinates), OneEuroFilter.__call__ :117-149, EmaFilter :155-162."""
from __future__ import annotations

import math

import numpy as np


def _alpha(cutoff, t_e: float = 1.0):
    r = 2.0 * math.pi * cutoff * t_e
    return r / (r + 1.0)


def _lerp(a, new, old):
    return a * new + (1.0 - a) * old


class OneEuroFilter:
    """x_hat = a*x + (1-a)*x_prev with a driven by the (smoothed) per-point speed."""

    def __init__(self, dx0: float = 0.0, min_cutoff: float = 0.15, beta: float = 0.8, d_cutoff: float = 1.0):
        self.min_cutoff, self.beta, self.d_cutoff = min_cutoff, beta, d_cutoff
        self.dx_prev = dx0

    def __call__(self, x, x_prev, dx_prev):
        speed = np.sqrt(np.sum((x - x_prev) ** 2, axis=1))
        speed_prev = np.sqrt(np.sum(np.asarray(dx_prev) ** 2, axis=1))
        speed_hat = _lerp(_alpha(self.d_cutoff), speed, speed_prev)
        a = _alpha(self.min_cutoff + self.beta * np.abs(speed_hat))[:, None]
        a[speed < 0.002] = 0.01          # nearly static points are frozen (lk.py:140-141)
        self.dx_prev = speed_hat
        return _lerp(a, x, x_prev)


class EmaFilter:
    def __init__(self, alpha: float):
        self.alpha = alpha

    def __call__(self, p_now, p_previous):
        return _lerp(self.alpha, p_now, p_previous)


def _hull(points):
    return [np.min(points[:, 0]), np.min(points[:, 1]), np.max(points[:, 0]), np.max(points[:, 1])]


def _iou(r1, r2) -> float:
    total = (r1[2] - r1[0]) * (r1[3] - r1[1]) + (r2[2] - r2[0]) * (r2[3] - r2[1])
    iw = max(0, min(r1[2], r2[2]) - max(r1[0], r2[0]))
    ih = max(0, min(r1[3], r2[3]) - max(r1[1], r2[1]))
    inter = iw * ih
    return inter / (total - inter)


class GroupTrack:
    def __init__(self, cfg):
        self.previous_landmarks_set = None
        self.previous_dx = None
        self.thres = cfg["pixel_thres"]     # read but unused, as in the reference (lk.py:12)
        self.iou_thres = cfg["iou_thres"]
        self.filter = OneEuroFilter()

    def iou(self, p_set0, p_set1) -> float:
        return _iou(_hull(p_set0), _hull(p_set1))

    def smooth(self, now_landmarks, previous_landmarks, previous_df):
        return self.filter(now_landmarks, previous_landmarks, previous_df)

    def calculate(self, img, now_landmarks_set):
        h, w = img.shape[0], img.shape[1]
        scale = [w, h]
        prev = self.previous_landmarks_set
        if prev is None or prev.shape[0] == 0:
            result = now_landmarks_set
            dxs = np.zeros_like(now_landmarks_set)
        else:
            result, dxs = [], []
            for cur in now_landmarks_set:
                for j in range(prev.shape[0]):
                    if self.iou(cur, prev[j]) > self.iou_thres:
                        filt = self.smooth(cur / scale, prev[j] / scale, self.previous_dx[j] / scale) * scale
                        result.append(filt)
                        dxs.append(prev[j] - filt)
                        break
                else:
                    result.append(cur)
                    dxs.append(np.zeros_like(cur))
        result = np.array(result)
        self.previous_landmarks_set = result
        self.previous_dx = np.array(dxs)
        return result
