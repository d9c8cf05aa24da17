"""Oracle: numpy restatement of the reference's pre/post-processing.  TEST INFRASTRUCTURE ONLY.

Restates (paths relative to /root/reference):
  * ``Skps/core/api/face_detector.py``  preprocess :45-71, xywh2xyxy :73-80,
    scale_coords :82-93, py_nms :95-136, __call__ :23-42
  * ``Skps/core/api/face_landmark.py``  preprocess :66-104, postprocess :106-115,
    the per-face normalisation :44-47
  * ``Skps/core/api/facer.py``          sort_and_filter :120-142 (top-k by area)

and the two OpenCV primitives those call (``opencv_python==4.6.0.66``, requirements.txt:5 --
third-party, absent here => PARITY UNPINNED, restated from the published algorithm of
``modules/imgproc/src/resize.cpp``):
  * ``cv2.resize(..., INTER_LINEAR)`` on uint8: 11-bit fixed-point coefficients
    (``INTER_RESIZE_COEF_BITS``), horizontal pass in int32, vertical pass
    ``(((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2``; exact-2x down-scale is routed to the
    2x2 box average like OpenCV does.
  * ``cv2.copyMakeBorder(..., BORDER_CONSTANT)``.

numpy pin: the reference targets numpy==1.23.4 (requirements.txt:4) whose scalar promotion
differs from the numpy 2.x installed here (NEP 50); the box arithmetic below is therefore written
with explicit float32 / float64 steps that reproduce the 1.23 result on any numpy.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


# --------------------------------------------------------------------------------------
# OpenCV-style resize
# --------------------------------------------------------------------------------------
def _round_half_even_to_short(v: np.ndarray) -> np.ndarray:
    # saturate_cast<short>(float) == cvRound == round-half-to-even
    return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int32)


def linear_coeffs(src_len: int, dst_len: int):
    """Per-destination-index (offset, a0, a1) of OpenCV's fixed-point bilinear.  Returns
    ``ofs`` (int32 source index of tap 0), ``a0``/``a1`` (int32, 11-bit weights) and ``nmax``
    = first destination index whose second tap falls outside the source (for the horizontal
    pass those indices use tap 0 with weight ONE)."""
    inv_scale = float(dst_len) / float(src_len)
    scale = 1.0 / inv_scale
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low = s < 0
    f[low] = 0.0
    s[low] = 0
    over = (s + 1) >= src_len
    nmax = int(np.argmax(over)) if over.any() else dst_len
    hi = s >= src_len - 1
    f[hi] = 0.0
    s[hi] = src_len - 1
    a0 = _round_half_even_to_short((np.float32(1.0) - f) * np.float32(COEF_ONE))
    a1 = _round_half_even_to_short(f * np.float32(COEF_ONE))
    return s, a0, a1, nmax


def linear_coeffs_vertical(src_len: int, dst_len: int):
    """Vertical flavour: weights are never reset, the two rows are clamped instead."""
    inv_scale = float(dst_len) / float(src_len)
    scale = 1.0 / inv_scale
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    b0 = _round_half_even_to_short((np.float32(1.0) - f) * np.float32(COEF_ONE))
    b1 = _round_half_even_to_short(f * np.float32(COEF_ONE))
    r0 = np.clip(s, 0, src_len - 1)
    r1 = np.clip(s + 1, 0, src_len - 1)
    return r0, r1, b0, b1


def resize_linear_u8(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(src, (dst_w, dst_h)) for uint8 HxWxC, default INTER_LINEAR."""
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw, _ = src.shape
    if sw == 2 * dst_w and sh == 2 * dst_h:
        # OpenCV: "INTER_AREA (fast) also is equal to INTER_LINEAR" for an exact 2x shrink
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xo, a0, a1, xmax = linear_coeffs(sw, dst_w)
    r0, r1, b0, b1 = linear_coeffs_vertical(sh, dst_h)
    s = src.astype(np.int32)
    x1 = np.minimum(xo + 1, sw - 1)
    a0e = a0.copy()
    a1e = a1.copy()
    a0e[xmax:] = COEF_ONE
    a1e[xmax:] = 0
    # horizontal pass for every source row that is needed
    hrow = s[:, xo, :] * a0e[None, :, None] + s[:, x1, :] * a1e[None, :, None]
    top = hrow[r0]
    bot = hrow[r1]
    out = (((b0[:, None, None] * (top >> 4)) >> 16) + ((b1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
    return (out & 0xFF).astype(np.uint8)


def pad_constant(img: np.ndarray, top: int, bottom: int, left: int, right: int, value=0) -> np.ndarray:
    h, w, c = img.shape
    out = np.empty((h + top + bottom, w + left + right, c), img.dtype)
    out[...] = np.asarray(value, img.dtype)
    out[top:top + h, left:left + w] = img
    return out


# --------------------------------------------------------------------------------------
# detector side (face_detector.py)
# --------------------------------------------------------------------------------------
def letterbox_geometry(h: int, w: int, input_hw=(384, 640)):
    """scale / resized size / paddings of face_detector.py:51-61."""
    scale = min(input_hw[0] / h, input_hw[1] / w)
    rw, rh = int(w * scale), int(h * scale)
    dh = (input_hw[0] - rh) / 2
    dw = (input_hw[1] - rw) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return scale, rw, rh, top, bottom, left, right


def detector_preprocess_u8(image_bgr: np.ndarray, input_hw=(384, 640), pad_value=114) -> Tuple[np.ndarray, list]:
    """face_detector.py:45-63 up to (not including) the float conversion: RGB uint8 letterboxed
    image HxWx3 plus ``[scale, left, top]``."""
    rgb = image_bgr[:, :, ::-1]
    h, w, _ = rgb.shape
    scale, rw, rh, top, bottom, left, right = letterbox_geometry(h, w, input_hw)
    img = resize_linear_u8(np.ascontiguousarray(rgb), rw, rh)
    img = pad_constant(img, top, bottom, left, right, pad_value)
    return img, [scale, left, top]


def detector_preprocess(image_bgr: np.ndarray, input_hw=(384, 640)) -> Tuple[np.ndarray, list]:
    """Full face_detector.py:45-71: float32 [1,3,H,W] in [0,1] + recover info."""
    img, info = detector_preprocess_u8(image_bgr, input_hw)
    x = img.transpose(2, 0, 1).astype(np.float32)
    x /= np.float32(255.0)
    return x[None], info


def xywh_to_xyxy(x: np.ndarray) -> np.ndarray:
    """face_detector.py:73-80 (float32 arithmetic)."""
    y = x.copy()
    half_w = x[:, 2] / np.float32(2)
    half_h = x[:, 3] / np.float32(2)
    y[:, 0] = x[:, 0] - half_w
    y[:, 1] = x[:, 1] - half_h
    y[:, 2] = x[:, 0] + half_w
    y[:, 3] = x[:, 1] + half_h
    return y


def greedy_nms(rows: np.ndarray, iou_thres: float, score_thres: float) -> np.ndarray:
    """face_detector.py:95-136.  ``rows``: (n,16) float32 with xyxy in cols 0:4 and the score
    in col 4.  strict ``>`` on the score, descending score order, strict ``<`` on IoU,
    IoU = inter / (a + b - inter) with no +1 and no epsilon (0/0 -> NaN -> suppressed).
    Returns the kept rows (all 16 columns) in keep order; also see ``greedy_nms_indices``."""
    idx = greedy_nms_indices(rows, iou_thres, score_thres)
    return rows[idx]


def greedy_nms_indices(rows: np.ndarray, iou_thres: float, score_thres: float, ties_by_row: bool = False) -> np.ndarray:
    """``ties_by_row``: equal scores are visited in ascending row order (the engine's rule).  The reference's
    ``np.argsort(...)[::-1]`` (face_detector.py:106) leaves the order of equal scores to the sort implementation, so with
    ties present only this variant is a well-defined checker."""
    cand = np.nonzero(rows[:, 4] > score_thres)[0]
    sub = rows[cand]
    order = np.argsort(-sub[:, 4], kind="stable") if ties_by_row else np.argsort(sub[:, 4])[::-1]
    x1, y1, x2, y2 = sub[:, 0], sub[:, 1], sub[:, 2], sub[:, 3]
    kept: List[int] = []
    with np.errstate(invalid="ignore", divide="ignore"):
        while order.size:
            c = order[0]
            kept.append(int(cand[c]))
            rest = order[1:]
            area_c = (x2[c] - x1[c]) * (y2[c] - y1[c])
            iw = np.maximum(np.float32(0), np.minimum(x2[c], x2[rest]) - np.maximum(x1[c], x1[rest]))
            ih = np.maximum(np.float32(0), np.minimum(y2[c], y2[rest]) - np.maximum(y1[c], y1[rest]))
            inter = ih * iw
            iou = inter / (area_c + (y2[rest] - y1[rest]) * (x2[rest] - x1[rest]) - inter)
            order = rest[iou < iou_thres]
    return np.asarray(kept, np.int64)


def unletterbox(boxes_xyxy: np.ndarray, info) -> np.ndarray:
    """face_detector.py:82-93: subtract the padding, divide by the scale (float32 array op
    with python scalars => float32 arithmetic)."""
    scale, dx, dy = info
    b = boxes_xyxy.astype(np.float32).copy()
    b[:, 0] -= np.float32(dx)
    b[:, 1] -= np.float32(dy)
    b[:, 2] -= np.float32(dx)
    b[:, 3] -= np.float32(dy)
    b /= np.float32(scale)
    return b


def detector_postprocess(raw: np.ndarray, info, iou_thres=0.3, score_thres=0.5, ties_by_row: bool = False) -> np.ndarray:
    """face_detector.py:31-37 on the raw network output (15120,16)."""
    out = np.array(raw, np.float32).reshape(-1, 16).copy()
    out[:, :4] = xywh_to_xyxy(out[:, :4])
    kept = out[greedy_nms_indices(out, iou_thres, score_thres, ties_by_row)]
    kept[:, :4] = unletterbox(kept[:, :4], info)
    return kept


def sort_and_filter(boxes: np.ndarray, min_face: float, top_k: int) -> np.ndarray:
    """facer.py:120-142: drop boxes with area <= min_face, keep the top_k largest."""
    if len(boxes) < 1:
        return np.zeros((0, 4), np.float32)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sel = area > min_face
    area, boxes = area[sel], boxes[sel]
    if boxes.shape[0] > top_k:
        pick = area.argsort()[-top_k:][::-1]
        boxes = boxes[pick]
    return np.array(boxes)


# --------------------------------------------------------------------------------------
# landmark side (face_landmark.py)
# --------------------------------------------------------------------------------------
class CropInfo:
    """Integer crop description of one face: the crop window in *padded* frame coordinates
    (x0..x1, y0..y1 as sliced by face_landmark.py:94), the padding ``add`` and the resulting
    crop size (after numpy's slice clamping)."""
    __slots__ = ("valid", "add", "x0", "y0", "x1", "y1", "w_crop", "h_crop")


def landmark_crop_box(bbox_xyxy: Sequence[float], frame_h: int, frame_w: int,
                      min_face: float = 20.0, extend: float = 0.2, numpy1_promotion: bool = True) -> CropInfo:
    """Box arithmetic of face_landmark.py:74-94 under numpy-1.23 promotion rules:
    ``bbox`` is a float32 row; width/height are float32; ``(1+2*extend)*width`` and every
    ``// 2`` promote to float64; results are stored back into the float32 row and truncated
    by ``astype(int32)``."""
    if np.asarray(bbox_xyxy).dtype == np.float64:
        return _landmark_crop_box_f64(np.asarray(bbox_xyxy, np.float64)[:4].copy(), frame_h, frame_w, min_face, extend)
    b = np.asarray(bbox_xyxy, np.float32)[:4].copy()
    ci = CropInfo()
    w = np.float32(b[2] - b[0])
    h = np.float32(b[3] - b[1])
    ci.valid = not (w <= min_face or h <= min_face)
    ci.add = ci.x0 = ci.y0 = ci.x1 = ci.y1 = ci.w_crop = ci.h_crop = 0
    if not ci.valid:
        return ci
    add = int(max(w, h))
    b = (b + np.float32(add)).astype(np.float32)
    if numpy1_promotion:
        face_width = (1 + 2 * extend) * float(w)               # float64
        cx = float(np.float32(b[0] + b[2])) // 2                # float64 floor-div
        cy = float(np.float32(b[1] + b[3])) // 2
        half = face_width // 2
    else:
        # numpy >= 2 (NEP 50): python scalars are weak, everything stays float32.  Only used to pin
        # this restatement against the reference source executed under the numpy installed here.
        face_width = np.float32(1 + 2 * extend) * w
        cx = float(np.float32(b[0] + b[2]) // np.float32(2))
        cy = float(np.float32(b[1] + b[3]) // np.float32(2))
        half = float(face_width // np.float32(2))
    box = np.array([cx - half, cy - half, cx + half, cy + half], np.float64).astype(np.float32)
    x0, y0, x1, y1 = (int(v) for v in box.astype(np.int32))
    ph, pw = frame_h + 2 * add, frame_w + 2 * add
    # numpy slice semantics of bimg[y0:y1, x0:x1]; negative starts (face far outside the frame)
    # wrap around in the reference (SURVEY App. D "quirks") -- here they clamp to 0 instead.
    xs, xe = min(max(x0, 0), pw), min(max(x1, 0), pw)
    ys, ye = min(max(y0, 0), ph), min(max(y1, 0), ph)
    ci.add, ci.x0, ci.y0, ci.x1, ci.y1 = add, x0, y0, x1, y1
    ci.w_crop, ci.h_crop = max(xe - xs, 0), max(ye - ys, 0)
    if ci.w_crop == 0 or ci.h_crop == 0:
        ci.valid = False
    return ci


def _landmark_crop_box_f64(b: np.ndarray, frame_h: int, frame_w: int, min_face: float, extend: float) -> CropInfo:
    """face_landmark.py:74-93 for a float64 row (tracked frames: FaceAna.track_box is float64): every operation is
    float64 under numpy 1.23 and numpy 2 alike -- no float32 store-back between the steps."""
    ci = CropInfo()
    w, h = b[2] - b[0], b[3] - b[1]
    ci.valid = not (w <= min_face or h <= min_face)
    ci.add = ci.x0 = ci.y0 = ci.x1 = ci.y1 = ci.w_crop = ci.h_crop = 0
    if not ci.valid:
        return ci
    add = int(max(w, h))
    b = b + add
    face_width = (1 + 2 * extend) * w
    cx, cy = (b[0] + b[2]) // 2, (b[1] + b[3]) // 2
    box = np.array([cx - face_width // 2, cy - face_width // 2, cx + face_width // 2, cy + face_width // 2], np.float64)
    x0, y0, x1, y1 = (int(v) for v in box.astype(np.int32))
    ph, pw = frame_h + 2 * add, frame_w + 2 * add
    xs, xe = min(max(x0, 0), pw), min(max(x1, 0), pw)
    ys, ye = min(max(y0, 0), ph), min(max(y1, 0), ph)
    ci.add, ci.x0, ci.y0, ci.x1, ci.y1 = add, x0, y0, x1, y1
    ci.w_crop, ci.h_crop = max(xe - xs, 0), max(ye - ys, 0)
    if ci.w_crop == 0 or ci.h_crop == 0:
        ci.valid = False
    return ci


def landmark_crop(image: np.ndarray, ci: CropInfo, out_hw=(256, 256)) -> np.ndarray:
    """face_landmark.py:79-98: zero-pad by ``add``, slice, cv2.resize to the network size."""
    bimg = pad_constant(image, ci.add, ci.add, ci.add, ci.add, 0)
    ph, pw, _ = bimg.shape
    xs, ys = min(max(ci.x0, 0), pw), min(max(ci.y0, 0), ph)
    crop = bimg[ys:ys + ci.h_crop, xs:xs + ci.w_crop]
    return resize_linear_u8(np.ascontiguousarray(crop), out_hw[1], out_hw[0])


def landmark_input(crop_u8: np.ndarray) -> np.ndarray:
    """face_landmark.py:44-47: HWC uint8 -> [1,3,H,W] float32 / 255 (channel order untouched)."""
    x = crop_u8.transpose(2, 0, 1).astype(np.float32)
    x = x / np.float32(255.0)
    return x[None]


def landmark_backproject(loc_fix: np.ndarray, ci: CropInfo) -> np.ndarray:
    """face_landmark.py:50-54,112-113: [196] normalised -> (98,2) frame pixels, float32 ops in
    the reference's order ((v * size) + origin) - add."""
    lm = np.asarray(loc_fix, np.float32).reshape(-1, 2).copy()
    lm[:, 0] = (lm[:, 0] * np.float32(ci.w_crop) + np.float32(ci.x0)) - np.float32(ci.add)
    lm[:, 1] = (lm[:, 1] * np.float32(ci.h_crop) + np.float32(ci.y0)) - np.float32(ci.add)
    return lm
