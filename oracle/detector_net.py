"""Oracle: torch-CPU restatement of the yolov5n-0.5 face detector.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference ships no source for this network: it loads a pre-exported
``pretrained/yolov5n-0.5.onnx`` (Skps/config/Skps.yml:4, absent -- ``.MISSING_LARGE_BLOBS:2``) from
deepcam-cn/yolov5-face (README.md:24-26, no commit pinned).  What the reference does pin is the
I/O contract (Skps/core/api/face_detector.py:29-37): float32 ``[1,3,384,640]`` RGB/255 in,
``(15120,16)`` rows out with cols 0:4 = cx,cy,w,h (letterboxed pixels) and col 4 = objectness.

The architecture below restates the published ``models/yolov5n-0.5.yaml`` / ``models/common.py``
(StemBlock, ShuffleV2Block, Conv = Conv2d+BN(eps 1e-3)+SiLU, C3, nearest Upsample, Concat) and the
export-time Detect decode of ``models/yolo.py`` (sigmoid on cols 0:5 and 15, xy = (2s-0.5+grid)*stride,
wh = (2s)^2*anchor, landmarks = raw*anchor + grid*stride, rows ordered level, anchor, y, x).
State-dict names follow upstream (``model.<i>....``) so a real checkpoint would drop in.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # yolov5 initialize_weights() sets BatchNorm2d.eps = 1e-3
ANCHORS = [[4, 5, 8, 10, 13, 16], [23, 29, 43, 55, 73, 105], [146, 217, 231, 300, 335, 433]]
STRIDES = [8, 16, 32]
NO = 16  # 4 box + 1 obj + 10 landmark + 1 class

_CALIBRATING = None  # see oracle.landmark_net._CALIBRATING


def _bn_entries(prefix: str, ch: int):
    return [(f"{prefix}.weight", (ch,), "bn_gamma"), (f"{prefix}.bias", (ch,), "bn_beta"),
            (f"{prefix}.running_mean", (ch,), "bn_mean"), (f"{prefix}.running_var", (ch,), "bn_var")]


def _conv_entries(prefix: str, cin: int, cout: int, k: int):
    return [(f"{prefix}.conv.weight", (cout, cin, k, k), "conv")] + _bn_entries(f"{prefix}.bn", cout)


def _shuffle_entries(prefix: str, inp: int, oup: int, stride: int):
    bf = oup // 2
    out = []
    if stride > 1:
        out += [(f"{prefix}.branch1.0.weight", (inp, 1, 3, 3), "conv")] + _bn_entries(f"{prefix}.branch1.1", inp)
        out += [(f"{prefix}.branch1.2.weight", (bf, inp, 1, 1), "conv")] + _bn_entries(f"{prefix}.branch1.3", bf)
    cin2 = inp if stride > 1 else bf
    out += [(f"{prefix}.branch2.0.weight", (bf, cin2, 1, 1), "conv")] + _bn_entries(f"{prefix}.branch2.1", bf)
    out += [(f"{prefix}.branch2.3.weight", (bf, 1, 3, 3), "conv")] + _bn_entries(f"{prefix}.branch2.4", bf)
    out += [(f"{prefix}.branch2.5.weight", (bf, bf, 1, 1), "conv")] + _bn_entries(f"{prefix}.branch2.6", bf)
    return out


def _c3_entries(prefix: str, c1: int, c2: int):
    c_ = c2 // 2
    return (_conv_entries(f"{prefix}.cv1", c1, c_, 1) + _conv_entries(f"{prefix}.cv2", c1, c_, 1) +
            _conv_entries(f"{prefix}.cv3", 2 * c_, c2, 1) + _conv_entries(f"{prefix}.m.0.cv1", c_, c_, 1) +
            _conv_entries(f"{prefix}.m.0.cv2", c_, c_, 3))


# backbone stages after width_multiple 0.5: (layer index, in, out, repeats of stride-1 blocks)
_BACKBONE = [(1, 16, 64, 3), (3, 64, 128, 7), (5, 128, 256, 3)]


def param_inventory() -> List[Tuple[str, Tuple[int, ...], str]]:
    inv = []
    inv += _conv_entries("model.0.stem_1", 3, 16, 3)
    inv += _conv_entries("model.0.stem_2a", 16, 8, 1)
    inv += _conv_entries("model.0.stem_2b", 8, 16, 3)
    inv += _conv_entries("model.0.stem_3", 32, 16, 1)
    for li, cin, cout, reps in _BACKBONE:
        inv += _shuffle_entries(f"model.{li}", cin, cout, 2)
        for r in range(reps):
            inv += _shuffle_entries(f"model.{li + 1}.{r}", cout, cout, 1)
    inv += _conv_entries("model.7", 256, 64, 1)
    inv += _c3_entries("model.10", 192, 64)
    inv += _conv_entries("model.11", 64, 64, 1)
    inv += _c3_entries("model.14", 128, 64)
    inv += _conv_entries("model.15", 64, 64, 3)
    inv += _c3_entries("model.17", 128, 64)
    inv += _conv_entries("model.18", 64, 64, 3)
    inv += _c3_entries("model.20", 128, 64)
    for i in range(3):
        inv += [(f"model.21.m.{i}.weight", (3 * NO, 64, 1, 1), "conv"), (f"model.21.m.{i}.bias", (3 * NO,), "bias")]
    return inv


def _bn(W, prefix, x):
    if _CALIBRATING is not None:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        W[f"{prefix}.running_mean"] = mean.float().to(x.dtype)
        W[f"{prefix}.running_var"] = var.float().clamp_min(1e-3).to(x.dtype)
        _CALIBRATING.append(prefix)
    return F.batch_norm(x, W[f"{prefix}.running_mean"], W[f"{prefix}.running_var"],
                        W[f"{prefix}.weight"], W[f"{prefix}.bias"], False, 0.0, BN_EPS)


def _conv(W, p, x, k=1, s=1):
    return F.silu(_bn(W, f"{p}.bn", F.conv2d(x, W[f"{p}.conv.weight"], None, s, k // 2)))


def _shuffle(x):
    b, c, h, w = x.shape
    return x.view(b, 2, c // 2, h, w).transpose(1, 2).reshape(b, c, h, w)


def _shuffle_block(W, p, x, stride):
    if stride == 1:
        x1, x2 = x.chunk(2, dim=1)
    else:
        x1 = F.conv2d(x, W[f"{p}.branch1.0.weight"], None, stride, 1, groups=x.shape[1])
        x1 = _bn(W, f"{p}.branch1.1", x1)
        x1 = F.silu(_bn(W, f"{p}.branch1.3", F.conv2d(x1, W[f"{p}.branch1.2.weight"])))
        x2 = x
    y = F.silu(_bn(W, f"{p}.branch2.1", F.conv2d(x2, W[f"{p}.branch2.0.weight"])))
    y = _bn(W, f"{p}.branch2.4", F.conv2d(y, W[f"{p}.branch2.3.weight"], None, stride, 1, groups=y.shape[1]))
    y = F.silu(_bn(W, f"{p}.branch2.6", F.conv2d(y, W[f"{p}.branch2.5.weight"])))
    return _shuffle(torch.cat([x1, y], 1))


def _c3(W, p, x):
    y1 = _conv(W, f"{p}.cv1", x)
    y1 = _conv(W, f"{p}.m.0.cv2", _conv(W, f"{p}.m.0.cv1", y1), 3)  # Bottleneck, shortcut=False
    y2 = _conv(W, f"{p}.cv2", x)
    return _conv(W, f"{p}.cv3", torch.cat([y1, y2], 1))


def detector_features(W: Dict[str, torch.Tensor], x: torch.Tensor, taps=None):
    """Raw Detect conv outputs [P3, P4, P5], each [B, 48, ny, nx]."""
    def tap(n, t):
        if taps is not None:
            taps[n] = t
    s1 = _conv(W, "model.0.stem_1", x, 3, 2)
    tap("model.0.stem_1", s1)
    s2 = _conv(W, "model.0.stem_2b", _conv(W, "model.0.stem_2a", s1), 3, 2)
    sp = F.max_pool2d(s1, 2, 2, ceil_mode=True)
    x = _conv(W, "model.0.stem_3", torch.cat([s2, sp], 1))
    tap("model.0", x)
    feats = {}
    for li, cin, cout, reps in _BACKBONE:
        x = _shuffle_block(W, f"model.{li}", x, 2)
        tap(f"model.{li}", x)
        for r in range(reps):
            x = _shuffle_block(W, f"model.{li + 1}.{r}", x, 1)
        tap(f"model.{li + 1}", x)
        feats[li + 1] = x
    l7 = _conv(W, "model.7", feats[6])
    l10 = _c3(W, "model.10", torch.cat([F.interpolate(l7, scale_factor=2, mode="nearest"), feats[4]], 1))
    l11 = _conv(W, "model.11", l10)
    l14 = _c3(W, "model.14", torch.cat([F.interpolate(l11, scale_factor=2, mode="nearest"), feats[2]], 1))
    l17 = _c3(W, "model.17", torch.cat([_conv(W, "model.15", l14, 3, 2), l11], 1))
    l20 = _c3(W, "model.20", torch.cat([_conv(W, "model.18", l17, 3, 2), l7], 1))
    for n, t in (("model.14", l14), ("model.17", l17), ("model.20", l20)):
        tap(n, t)
    outs = []
    for i, f in enumerate((l14, l17, l20)):
        o = F.conv2d(f, W[f"model.21.m.{i}.weight"], W[f"model.21.m.{i}.bias"])
        tap(f"model.21.m.{i}", o)
        outs.append(o)
    return outs


def detect_decode(raw: List[torch.Tensor]) -> torch.Tensor:
    """Detect.forward in export/concat mode -> [B, rows, 16]."""
    zs = []
    for i, x in enumerate(raw):
        b, _, ny, nx = x.shape
        x = x.view(b, 3, NO, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).to(x.dtype)
        anchor = torch.tensor(ANCHORS[i], dtype=x.dtype).view(1, 3, 1, 1, 2)
        stride = float(STRIDES[i])
        y = torch.zeros_like(x)
        cr = [0, 1, 2, 3, 4, 15]
        y[..., cr] = x[..., cr].sigmoid()
        y[..., 5:15] = x[..., 5:15]
        xy = (y[..., 0:2] * 2.0 - 0.5 + grid) * stride
        wh = (y[..., 2:4] * 2) ** 2 * anchor
        parts = [xy, wh, y[..., 4:5]]
        for k in range(5):
            parts.append(y[..., 5 + 2 * k:7 + 2 * k] * anchor + grid * stride)
        parts.append(y[..., 15:16])
        zs.append(torch.cat(parts, -1).view(b, -1, NO))
    return torch.cat(zs, 1)


def detector_forward(W: Dict[str, torch.Tensor], x: torch.Tensor, taps=None) -> torch.Tensor:
    """float [B,3,384,640] RGB/255 -> [B,15120,16] (the ONNX output consumed at face_detector.py:29-31)."""
    return detect_decode(detector_features(W, x, taps))
