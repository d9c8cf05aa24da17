"""ctypes binding of ``libpeppa_hip.so`` (C ABI: ``include/peppa_hip.h``).

No torch, no onnxruntime: numpy arrays in, numpy arrays out.  There is deliberately no CPU
fallback -- if the HIP library is missing or no GPU is visible, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

PF_MEM_HOST, PF_MEM_DEVICE, PF_MEM_RESIDENT, PF_MEM_HOST_PINNED = 0, 1, 2, 3
PF_MEM_ROWS_DEVICE = 0x100
PF_NET_LANDMARK, PF_NET_DETECTOR = 0, 1
PF_INPUT_U8_NHWC, PF_INPUT_F32_NCHW = 0, 1
PF_OPT_HIP_GRAPH = 1
PF_OPT_RANGE_CHECK = 2
PF_OPT_JPEG_ENTROPY, PF_OPT_JPEG_SYNC_ROUNDS = 3, 4       # entropy decoding: 0 automatic / 1 host / 2 device; sync rounds 1..10 (0 = all)
PF_OPT_BATCH_FRONT = 5      # BatchEngine.set_option only: 1 (default) detector + NMS once per call on the front engine, 0 per lane
PF_COMM_ID_BYTES = 128

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "libpeppa_hip.so")

_lib_cache = {}


class PeppaHipError(RuntimeError):
    pass


def _declare(lib):
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.pf_version.restype = C.c_char_p
    lib.pf_version.argtypes = []
    lib.pf_create.argtypes = [i, C.POINTER(vp)]
    lib.pf_destroy.argtypes = [vp]
    lib.pf_destroy.restype = None
    lib.pf_last_error.argtypes = [vp]
    lib.pf_last_error.restype = C.c_char_p
    lib.pf_sync.argtypes = [vp]
    lib.pf_load_program.argtypes = [vp, i, vp, sz, i]
    lib.pf_landmark_forward.argtypes = [vp, vp, i, i, i, vp, vp, i]
    lib.pf_detector_forward.argtypes = [vp, vp, i, i, i, vp, i]
    lib.pf_read_tensor.argtypes = [vp, i, i, i, fp, sz]
    lib.pf_detect.argtypes = [vp, vp, i, i, i, i, f, f, fp, i, ip]
    lib.pf_landmarks.argtypes = [vp, vp, i, i, i, i, fp, i, fp, fp, ip]
    lib.pf_landmarks_f64.argtypes = [vp, vp, i, i, i, i, C.POINTER(C.c_double), i, fp, fp, ip]
    lib.pf_landmarks_f64.restype = i
    lib.pf_crop_faces_f64.argtypes = [vp, vp, i, i, i, i, C.POINTER(C.c_double), i, i, vp, ip]
    lib.pf_crop_faces_f64.restype = i
    dp = C.POINTER(C.c_double)
    lib.pf_track_frame.argtypes = [vp, vp, i, i, i, i, f, f, f, i, f, f, f, i, ip, dp, dp, fp, ip]
    lib.pf_track_frame.restype = i
    lib.pf_track_frame_planted.argtypes = [vp, vp, i, i, i, i, fp, i, f, f, f, i, f, f, f, ip, dp, dp, fp, ip]
    lib.pf_track_frame_planted.restype = i
    lib.pf_track_reset.argtypes = [vp]
    lib.pf_track_reset.restype = i
    lib.pf_run_frames.argtypes = [vp, vp, i, i, i, i, f, f, f, i, vp, vp, vp, vp, i]
    lib.pf_run_frames_planted.argtypes = [vp, vp, i, i, i, i, vp, i, f, f, f, i, vp, vp, vp, vp, i]
    lib.pf_letterbox.argtypes = [vp, vp, i, i, i, i, i, i, vp, fp]
    lib.pf_nms_rows.argtypes = [vp, fp, i, f, f, f, f, f, fp, i, ip]
    lib.pf_resize.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    lib.pf_resize.restype = i
    lib.pf_crop_faces.argtypes = [vp, vp, i, i, i, i, fp, i, i, vp, ip]
    lib.pf_jpeg_info.argtypes = [vp, sz, ip, ip, ip, ip]
    lib.pf_decode_jpeg.argtypes = [vp, vp, sz, ip, ip, C.POINTER(vp), vp]
    lib.pf_decode_jpeg_batch.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), i, ip, ip, C.POINTER(vp)]
    lib.pf_set_frame.argtypes = [vp, vp, i, i, i, i, C.POINTER(C.c_ulonglong), ip]
    lib.pf_forget_frames.argtypes = [vp]
    lib.pf_set_option.argtypes = [vp, i, i]
    lib.pf_profile_enable.argtypes = [vp, i]
    lib.pf_profile_fetch.argtypes = [vp, C.c_char_p, sz, fp, ip, i, ip]
    lib.pf_host_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.pf_host_free.argtypes = [vp]
    lib.pf_host_alloc.restype = i
    lib.pf_host_free.restype = i
    lib.pf_comm_unique_id.argtypes = [vp, sz]
    lib.pf_rccl_version.argtypes = [ip]
    lib.pf_broadcast_weights.argtypes = [vp, vp, i, i, i, vp, sz, C.POINTER(sz), i, fp]
    lib.pf_comm_destroy.argtypes = [vp]
    lib.pf_batch_create.argtypes = [i, i, C.POINTER(vp)]
    lib.pf_batch_destroy.argtypes = [vp]
    lib.pf_batch_destroy.restype = None
    lib.pf_batch_last_error.argtypes = [vp]
    lib.pf_batch_last_error.restype = C.c_char_p
    lib.pf_batch_lanes.argtypes = [vp]
    lib.pf_batch_lane.argtypes = [vp, i]
    lib.pf_batch_lane.restype = vp
    lib.pf_batch_front.argtypes = [vp]
    lib.pf_batch_front.restype = vp
    lib.pf_batch_load_program.argtypes = [vp, i, vp, sz, i]
    lib.pf_batch_set_option.argtypes = [vp, i, i]
    lib.pf_batch_sync.argtypes = [vp]
    lib.pf_batch_run_frames.argtypes = [vp, vp, i, i, i, i, vp, i, f, f, f, i, vp, vp, vp, vp, i]
    for name in ("pf_batch_create", "pf_batch_lanes", "pf_batch_load_program", "pf_batch_set_option", "pf_batch_sync", "pf_batch_run_frames"):
        getattr(lib, name).restype = i
    for name in ("pf_comm_unique_id", "pf_rccl_version", "pf_broadcast_weights", "pf_comm_destroy"):
        getattr(lib, name).restype = i
    for name in ("pf_create", "pf_sync", "pf_load_program", "pf_landmark_forward", "pf_detector_forward",
                 "pf_read_tensor", "pf_detect", "pf_landmarks", "pf_run_frames", "pf_run_frames_planted",
                 "pf_profile_enable", "pf_profile_fetch", "pf_letterbox", "pf_nms_rows", "pf_crop_faces", "pf_set_frame", "pf_forget_frames", "pf_set_option"):
        getattr(lib, name).restype = i
    return lib


def load_library(path: Optional[str] = None):
    """dlopen the engine.  ``path`` defaults to the in-tree ``libpeppa_hip.so`` built by
    ``__graft_entry__.build()`` / ``python -m peppa_pig_face_landmark_amd.build``."""
    path = os.path.abspath(path or os.environ.get("PEPPA_HIP_LIBRARY", DEFAULT_LIBRARY))
    if path in _lib_cache:
        return _lib_cache[path]
    if not os.path.exists(path):
        raise PeppaHipError(
            f"{path} not found: build it with `python -m peppa_pig_face_landmark_amd.build` "
            "(hipcc --offload-arch=gfx950).  The engine has no CPU fallback.")
    lib = _declare(C.CDLL(path))
    _lib_cache[path] = lib
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class DeviceFrame:
    """A packed BGR uint8 frame that already lives in device memory (``Engine.imread`` / ``decode_jpeg``): accepted wherever the
    engine takes a frame.  ``shape`` mirrors the numpy array cv2.imread would have returned; ``numpy()`` is the host copy
    (for drawing).  The memory belongs to the engine and is reused by its next decode: consume the frame first (``FaceAna.run``
    copies it into its resident-frame slot on the device)."""

    def __init__(self, ptr: int, height: int, width: int, host: Optional[np.ndarray] = None):
        self.ptr, self.shape, self._host = int(ptr), (int(height), int(width), 3), host
        self.dtype = np.dtype(np.uint8)

    def numpy(self) -> np.ndarray:
        if self._host is None:
            raise PeppaHipError("this DeviceFrame was decoded without a host copy (imread(..., want_host=False))")
        return self._host


class Engine:
    """One engine == one GPU + one HIP stream (``pf_handle``)."""

    def __init__(self, device: int = 0, library: Optional[str] = None):
        self.lib = load_library(library)
        self.h = C.c_void_p()
        if self.lib.pf_create(int(device), C.byref(self.h)) != 0:
            msg = self.lib.pf_last_error(None)
            raise PeppaHipError("pf_create failed: " + (msg.decode() if msg else "unknown"))
        self.device = device
        self._programs = {}

    def pinned_empty(self, shape, dtype=np.uint8) -> np.ndarray:
        """Page-locked host array (pf_host_alloc): decode frames straight into it and pass it to run_frames* so the
        host->device copy is asynchronous.  Freed by close(); do not use the array afterwards."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        if self.lib.pf_host_alloc(nbytes, C.byref(p)) != 0 or not p.value:
            raise PeppaHipError("pf_host_alloc(%d bytes) failed" % nbytes)
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p)
        buf = (C.c_ubyte * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def run_frames_host_async(self, frames: np.ndarray, d_planted: int, rows: int, score_thres: float, iou_thres: float,
                              min_face: float, top_k: int, d_counts: int, d_boxes: int, d_kps: int, d_scores: int):
        """Host-resident frames (ideally from pinned_empty) in, device-resident results out, no synchronisation:
        the call returns after enqueueing the copy and the kernels on this engine's stream.  Planted detector rows
        (benchmark instrument) are passed as a device pointer."""
        F, H, W, _ = frames.shape
        assert frames.dtype == np.uint8 and frames.flags["C_CONTIGUOUS"]
        rc = self.lib.pf_run_frames_planted(self.h, _ptr(frames), PF_MEM_HOST | PF_MEM_ROWS_DEVICE, F, H, W,
                                            _ptr(d_planted) if d_planted else None, rows, score_thres, iou_thres,
                                            min_face, top_k, _ptr(d_counts), _ptr(d_boxes), _ptr(d_kps), _ptr(d_scores),
                                            PF_MEM_DEVICE)
        self._check(rc, "pf_run_frames_planted")

    def close(self):
        for p in getattr(self, "_pinned", []):
            self.lib.pf_host_free(p)
        self._pinned = []
        if getattr(self, "h", None) is not None and self.h:
            if not getattr(self, "_borrowed", False):      # a lane of a BatchEngine belongs to its pf_batch
                self.lib.pf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def version(self) -> str:
        return self.lib.pf_version().decode()

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.pf_last_error(self.h)
            raise PeppaHipError(f"{what} failed: " + (msg.decode() if msg else "unknown"))

    def sync(self):
        self._check(self.lib.pf_sync(self.h), "pf_sync")

    def load_program(self, slot: int, blob: bytes, max_batch: int):
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self.lib.pf_load_program(self.h, slot, C.cast(buf, C.c_void_p), len(blob), int(max_batch)),
                    "pf_load_program")
        self._programs[slot] = max_batch

    # ---- multi-GPU: one-time RCCL broadcast of the packed programs (include/peppa_hip.h) ----------------
    @staticmethod
    def comm_unique_id(library: Optional[str] = None) -> bytes:
        """Rank 0: a fresh RCCL unique id (128 bytes) to hand to the other ranks out of band."""
        lib = load_library(library)
        buf = C.create_string_buffer(PF_COMM_ID_BYTES)
        if lib.pf_comm_unique_id(C.cast(buf, C.c_void_p), PF_COMM_ID_BYTES) != 0:
            msg = lib.pf_last_error(None)
            raise PeppaHipError("pf_comm_unique_id failed: " + (msg.decode() if msg else "unknown"))
        return buf.raw

    def rccl_version(self) -> int:
        v = C.c_int(0)
        if self.lib.pf_rccl_version(C.byref(v)) != 0:
            msg = self.lib.pf_last_error(None)
            raise PeppaHipError("pf_rccl_version failed: " + (msg.decode() if msg else "unknown"))
        return v.value

    def broadcast_weights(self, unique_id: bytes, rank: int, world: int, slot: int, blob: Optional[bytes],
                          max_batch: int, capacity: int = 64 << 20) -> Tuple[bytes, float]:
        """Collective: rank 0 passes its packed program, the other ranks pass None; every rank ends up with the
        program loaded into `slot`.  Returns (blob bytes, device milliseconds of the ncclBroadcast)."""
        assert len(unique_id) == PF_COMM_ID_BYTES
        if rank == 0:
            assert blob is not None
            capacity = len(blob)
            buf = C.create_string_buffer(blob, len(blob))
        else:
            buf = C.create_string_buffer(capacity)
        n = C.c_size_t(len(blob) if rank == 0 else 0)
        ms = C.c_float(0.0)
        idb = C.create_string_buffer(unique_id, PF_COMM_ID_BYTES)
        self._check(self.lib.pf_broadcast_weights(self.h, C.cast(idb, C.c_void_p), int(rank), int(world), int(slot),
                                                  C.cast(buf, C.c_void_p), capacity, C.byref(n), int(max_batch), C.byref(ms)),
                    "pf_broadcast_weights")
        self._programs[slot] = max_batch
        return buf.raw[:n.value], float(ms.value)

    # ---- network seams ------------------------------------------------------------------------
    def landmark_forward(self, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """x: uint8 [B,S,S,3] or float32 [B,3,S,S] -> (loc_fix [B,196], score [B,98])."""
        if x.dtype == np.uint8:
            kind, batch = PF_INPUT_U8_NHWC, x.shape[0]
        elif x.dtype == np.float32:
            kind, batch = PF_INPUT_F32_NCHW, x.shape[0]
        else:
            raise TypeError("landmark input must be uint8 NHWC or float32 NCHW")
        x = np.ascontiguousarray(x)
        loc = np.empty((batch, 196), np.float32)
        score = np.empty((batch, 98), np.float32)
        self._check(self.lib.pf_landmark_forward(self.h, _ptr(x), kind, PF_MEM_HOST, batch, _ptr(loc), _ptr(score),
                                                 PF_MEM_HOST), "pf_landmark_forward")
        return loc, score

    def landmark_forward_device(self, d_input: int, kind: int, batch: int, d_loc: int = 0, d_score: int = 0):
        """Same on device pointers (bench): nothing crosses PCIe."""
        self._check(self.lib.pf_landmark_forward(self.h, _ptr(d_input), kind, PF_MEM_DEVICE, batch,
                                                 _ptr(d_loc) if d_loc else None, _ptr(d_score) if d_score else None,
                                                 PF_MEM_DEVICE), "pf_landmark_forward")

    def detector_forward(self, x: np.ndarray, rows: int = 15120) -> np.ndarray:
        if x.dtype == np.uint8:
            kind = PF_INPUT_U8_NHWC
        elif x.dtype == np.float32:
            kind = PF_INPUT_F32_NCHW
        else:
            raise TypeError("detector input must be uint8 NHWC or float32 NCHW")
        x = np.ascontiguousarray(x)
        batch = x.shape[0]
        out = np.empty((batch, rows, 16), np.float32)
        self._check(self.lib.pf_detector_forward(self.h, _ptr(x), kind, PF_MEM_HOST, batch, _ptr(out), PF_MEM_HOST),
                    "pf_detector_forward")
        return out

    def read_tensor(self, slot: int, tensor_id: int, batch: int, shape_hwc) -> np.ndarray:
        h, w, c = shape_hwc
        out = np.empty((batch, h, w, c), np.float32)
        self._check(self.lib.pf_read_tensor(self.h, slot, tensor_id, batch,
                                            out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "pf_read_tensor")
        return out

    # ---- video mode: resident frame + frame-difference gate ------------------------------------------
    def set_frame(self, image_bgr: np.ndarray):
        """Upload the frame once and keep it resident; returns the mean absolute difference to the
        previous resident frame exactly as FaceAna.diff_frames computes it (facer.py:111-113), or None
        when there is no previous frame of the same shape."""
        total = C.c_ulonglong(0)
        has_prev = C.c_int(0)
        if isinstance(image_bgr, DeviceFrame):
            img = image_bgr
            self._check(self.lib.pf_set_frame(self.h, C.c_void_p(img.ptr), PF_MEM_DEVICE, img.shape[0], img.shape[1], img.shape[1] * 3,
                                              C.byref(total), C.byref(has_prev)), "pf_set_frame")
        else:
            img = np.ascontiguousarray(image_bgr)
            assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
            self._check(self.lib.pf_set_frame(self.h, _ptr(img), PF_MEM_HOST, img.shape[0], img.shape[1], img.strides[0],
                                              C.byref(total), C.byref(has_prev)), "pf_set_frame")
        self._resident_shape = img.shape
        if not has_prev.value:
            return None
        return total.value / img.shape[0] / img.shape[1] / 3.0

    def forget_frames(self):
        self._check(self.lib.pf_forget_frames(self.h), "pf_forget_frames")
        self._resident_shape = None

    def _frame_args(self, image_bgr):
        """(pointer, mem, H, W, row_stride) for a host frame, or for the resident one when image is None."""
        if image_bgr is None:
            shp = getattr(self, "_resident_shape", None)
            if shp is None:
                raise PeppaHipError("no resident frame: call set_frame() first")
            return None, PF_MEM_RESIDENT, shp[0], shp[1], shp[1] * 3, None
        if isinstance(image_bgr, DeviceFrame):
            return C.c_void_p(image_bgr.ptr), PF_MEM_DEVICE, image_bgr.shape[0], image_bgr.shape[1], image_bgr.shape[1] * 3, image_bgr
        img = np.ascontiguousarray(image_bgr)
        assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
        return _ptr(img), PF_MEM_HOST, img.shape[0], img.shape[1], img.strides[0], img

    # ---- pipeline seams -----------------------------------------------------------------------
    def detect(self, image_bgr, score_thres: float, iou_thres: float, max_n: int = 1024) -> np.ndarray:
        """image_bgr: HxWx3 uint8, or None to use the frame stored by set_frame()."""
        ptr, mem, H, W, stride, _keep = self._frame_args(image_bgr)
        boxes = np.empty((max_n, 16), np.float32)
        n = C.c_int(0)
        self._check(self.lib.pf_detect(self.h, ptr, mem, H, W, stride,
                                       score_thres, iou_thres, boxes.ctypes.data_as(C.POINTER(C.c_float)), max_n,
                                       C.byref(n)), "pf_detect")
        return boxes[:n.value].copy()

    def landmarks(self, image_bgr, boxes: np.ndarray):
        """image_bgr: HxWx3 uint8, or None to use the frame stored by set_frame()."""
        ptr, mem, H, W, stride, _keep = self._frame_args(image_bgr)
        boxes = np.asarray(boxes)
        # float64 rows (tracked frames of FaceAna) keep their precision: the reference's box arithmetic is float64 then
        f64 = boxes.dtype == np.float64
        b = np.ascontiguousarray(np.asarray(boxes, np.float64 if f64 else np.float32)[:, :4])
        n = b.shape[0]
        kps = np.zeros((n, 98, 2), np.float32)
        scores = np.zeros((n, 98), np.float32)
        valid = np.zeros((n,), np.int32)
        if n:
            fn = self.lib.pf_landmarks_f64 if f64 else self.lib.pf_landmarks
            self._check(fn(self.h, ptr, mem, H, W,
                                              stride, b.ctypes.data_as(C.POINTER(C.c_double if f64 else C.c_float)), n,
                                              kps.ctypes.data_as(C.POINTER(C.c_float)),
                                              scores.ctypes.data_as(C.POINTER(C.c_float)),
                                              valid.ctypes.data_as(C.POINTER(C.c_int))), "pf_landmarks")
        return kps, scores, valid.astype(bool)

    def run_frames(self, frames: np.ndarray, score_thres: float, iou_thres: float, min_face: float, top_k: int,
                   planted_rows: Optional[np.ndarray] = None):
        fr = np.ascontiguousarray(frames)
        assert fr.dtype == np.uint8 and fr.ndim == 4 and fr.shape[3] == 3
        F, H, W, _ = fr.shape
        counts = np.zeros((F,), np.int32)
        boxes = np.zeros((F, top_k, 4), np.float32)
        kps = np.zeros((F, top_k, 98, 2), np.float32)
        scores = np.zeros((F, top_k, 98), np.float32)
        if planted_rows is None:
            rc = self.lib.pf_run_frames(self.h, _ptr(fr), PF_MEM_HOST, F, H, W, score_thres, iou_thres, min_face,
                                        top_k, _ptr(counts), _ptr(boxes), _ptr(kps), _ptr(scores), PF_MEM_HOST)
        else:
            pr = np.ascontiguousarray(planted_rows, np.float32)
            rc = self.lib.pf_run_frames_planted(self.h, _ptr(fr), PF_MEM_HOST, F, H, W, _ptr(pr), pr.shape[1],
                                                score_thres, iou_thres, min_face, top_k, _ptr(counts), _ptr(boxes),
                                                _ptr(kps), _ptr(scores), PF_MEM_HOST)
        self._check(rc, "pf_run_frames")
        return counts, boxes, kps, scores

    def run_frames_device(self, d_frames: int, F: int, H: int, W: int, score_thres: float, iou_thres: float,
                          min_face: float, top_k: int, d_planted: int = 0, rows: int = 0, d_counts: int = 0,
                          d_boxes: int = 0, d_kps: int = 0, d_scores: int = 0, out_mem: int = PF_MEM_DEVICE):
        """Device-resident frames; results to device buffers, or (out_mem = PF_MEM_HOST_PINNED) to page-locked host
        buffers from pinned_empty() -- asynchronous either way, complete after sync()."""
        rc = self.lib.pf_run_frames_planted(self.h, _ptr(d_frames), PF_MEM_DEVICE, F, H, W,
                                            _ptr(d_planted) if d_planted else None, rows, score_thres, iou_thres,
                                            min_face, top_k, _ptr(d_counts) if d_counts else None,
                                            _ptr(d_boxes) if d_boxes else None, _ptr(d_kps) if d_kps else None,
                                            _ptr(d_scores) if d_scores else None, out_mem)
        self._check(rc, "pf_run_frames_planted")

    # ---- stage-level seams -----------------------------------------------------------------------
    def letterbox(self, image_bgr: np.ndarray, out_hw=(384, 640)):
        """FaceDetector.preprocess (uint8 stage): -> (RGB uint8 [H,W,3], [scale, left, top])."""
        fptr, fmem, fh, fw, fstride, _keep = self._frame_args(image_bgr)
        out = np.empty((out_hw[0], out_hw[1], 3), np.uint8)
        info = np.zeros(3, np.float32)
        self._check(self.lib.pf_letterbox(self.h, fptr, fmem, fh, fw, fstride,
                                          out_hw[0], out_hw[1], _ptr(out), info.ctypes.data_as(C.POINTER(C.c_float))),
                    "pf_letterbox")
        return out, info

    def resize(self, image: np.ndarray, out_hw) -> np.ndarray:
        """cv2.resize(image, (out_w, out_h)) (INTER_LINEAR, uint8 HxWx3) on the GPU, bit-for-bit OpenCV's fixed point."""
        img = np.ascontiguousarray(image)
        assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
        out = np.empty((int(out_hw[0]), int(out_hw[1]), 3), np.uint8)
        self._check(self.lib.pf_resize(self.h, _ptr(img), PF_MEM_HOST, img.shape[0], img.shape[1], img.strides[0],
                                       int(out_hw[0]), int(out_hw[1]), _ptr(out)), "pf_resize")
        return out

    def jpeg_info(self, data: bytes) -> Tuple[int, int, int, int]:
        """(height, width, components, subsampling 0 | 444 | 422 | 420) of a JPEG stream the decoder accepts."""
        hh, ww, cc, ss = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        if self.lib.pf_jpeg_info(buf, len(data), C.byref(hh), C.byref(ww), C.byref(cc), C.byref(ss)) != 0:
            raise PeppaHipError("pf_jpeg_info: not a JPEG stream this decoder accepts")
        return hh.value, ww.value, cc.value, ss.value

    def decode_jpeg(self, data: bytes, want_host: bool = True):
        """cv2.imread for JPEG bytes (demo.py:76): Huffman decoding on the host, everything after it on the device.  Returns
        (device pointer of the packed BGR frame -- valid until the next decode --, height, width, host copy or None)."""
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        try:
            hh, ww, _, _ = self.jpeg_info(data)
        except PeppaHipError:      # let the decoder itself say why (its message names the unsupported feature)
            self._check(self.lib.pf_decode_jpeg(self.h, buf, len(data), None, None, None, None), "pf_decode_jpeg")
            raise
        out = np.empty((hh, ww, 3), np.uint8) if want_host else None
        d = C.c_void_p(0)
        h2, w2 = C.c_int(0), C.c_int(0)
        self._check(self.lib.pf_decode_jpeg(self.h, buf, len(data), C.byref(h2), C.byref(w2), C.byref(d),
                                            _ptr(out) if want_host else None), "pf_decode_jpeg")
        return d.value, h2.value, w2.value, out

    def decode_jpeg_batch(self, files, threads: int = 16):
        """n JPEG byte strings of one size -> (device pointer of [n][H][W][3] BGR, n, H, W) for run_frames_device: Huffman
        decoding of the files on `threads` host threads, the device stages once over the batch."""
        n = len(files)
        files = [bytes(d) for d in files]            # immutable: the pointers below stay valid for the call, no copies
        ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(d), C.c_void_p) for d in files])
        sizes = (C.c_size_t * n)(*[len(d) for d in files])
        hh, ww, d = C.c_int(0), C.c_int(0), C.c_void_p(0)
        self._check(self.lib.pf_decode_jpeg_batch(self.h, n, ptrs, sizes, int(threads), C.byref(hh), C.byref(ww), C.byref(d)),
                    "pf_decode_jpeg_batch")
        return d.value, n, hh.value, ww.value

    def run_jpeg_files(self, files, score_thres: float, iou_thres: float, min_face: float, top_k: int, threads: int = 16,
                       planted_rows: Optional[np.ndarray] = None):
        """``pf_decode_jpeg_batch`` + ``pf_run_frames`` on the decoded frames, synchronous, numpy results as ``run_frames``.

        The C ABI's batch decode is asynchronous and reports a Huffman stream the device's parallel entropy decoder could not
        synchronise (a valid file with a long stretch that never re-aligns) at the NEXT synchronising call, when the frames have
        already been consumed.  This wrapper owns the recovery the ABI leaves to its caller: the batch is decoded again with the
        host's Huffman loop (``PF_OPT_JPEG_ENTROPY`` = 1 for that one call) and the frames are run again -- no file that the
        host path decodes fails here."""
        files = [bytes(d) for d in files]
        F = len(files)
        counts = np.zeros((F,), np.int32)
        boxes = np.zeros((F, top_k, 4), np.float32)
        kps = np.zeros((F, top_k, 98, 2), np.float32)
        scores = np.zeros((F, top_k, 98), np.float32)

        pr = None if planted_rows is None else np.ascontiguousarray(planted_rows, np.float32)     # benchmark / test instrument, as run_frames

        def once():
            d, n, hh, ww = self.decode_jpeg_batch(files, threads)
            if pr is None:
                rc = self.lib.pf_run_frames(self.h, _ptr(d), PF_MEM_DEVICE, n, hh, ww, score_thres, iou_thres, min_face, top_k,
                                            _ptr(counts), _ptr(boxes), _ptr(kps), _ptr(scores), PF_MEM_HOST)
            else:
                rc = self.lib.pf_run_frames_planted(self.h, _ptr(d), PF_MEM_DEVICE, n, hh, ww, _ptr(pr), pr.shape[1], score_thres,
                                                    iou_thres, min_face, top_k, _ptr(counts), _ptr(boxes), _ptr(kps), _ptr(scores),
                                                    PF_MEM_HOST)
            self._check(rc, "pf_run_frames")
            self.sync()

        try:
            once()
        except PeppaHipError as e:
            if "did not synchronise" not in str(e):
                raise
            before = self.__dict__.get("_options", {}).get(PF_OPT_JPEG_ENTROPY, 0)
            self.set_option(PF_OPT_JPEG_ENTROPY, 1)
            try:
                once()
            finally:
                self.set_option(PF_OPT_JPEG_ENTROPY, before)
        return counts, boxes, kps, scores

    def imread(self, path_or_bytes, want_host: bool = True) -> "DeviceFrame":
        """cv2.imread(path) for baseline JPEG files, decoded into device memory (see decode_jpeg)."""
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
        d, hh, ww, host = self.decode_jpeg(bytes(data), want_host)
        return DeviceFrame(d, hh, ww, host)

    def nms_rows(self, rows: np.ndarray, scale: float, left: float, top: float, score_thres: float, iou_thres: float,
                 max_n: int = 1024) -> np.ndarray:
        """xywh2xyxy + py_nms + scale_coords on decoded rows [R,16]."""
        r = np.ascontiguousarray(rows, np.float32)
        kept = np.empty((max_n, 16), np.float32)
        n = C.c_int(0)
        self._check(self.lib.pf_nms_rows(self.h, r.ctypes.data_as(C.POINTER(C.c_float)), r.shape[0], float(scale),
                                         float(left), float(top), float(score_thres), float(iou_thres),
                                         kept.ctypes.data_as(C.POINTER(C.c_float)), max_n, C.byref(n)), "pf_nms_rows")
        return kept[:n.value].copy()

    def crop_faces(self, image_bgr: np.ndarray, boxes: np.ndarray, out_size: int = 256):
        """FaceLandmark.preprocess: -> (crops uint8 [n,S,S,3], params int32 [n,8])."""
        img = np.ascontiguousarray(image_bgr)
        f64 = np.asarray(boxes).dtype == np.float64
        b = np.ascontiguousarray(np.asarray(boxes, np.float64 if f64 else np.float32)[:, :4])
        n = b.shape[0]
        crops = np.zeros((n, out_size, out_size, 3), np.uint8)
        params = np.zeros((n, 8), np.int32)
        fn = self.lib.pf_crop_faces_f64 if f64 else self.lib.pf_crop_faces
        self._check(fn(self.h, _ptr(img), PF_MEM_HOST, img.shape[0], img.shape[1], img.strides[0],
                                           b.ctypes.data_as(C.POINTER(C.c_double if f64 else C.c_float)), n, out_size, _ptr(crops),
                                           params.ctypes.data_as(C.POINTER(C.c_int))), "pf_crop_faces")
        return crops, params

    # ---- video stream with the tracking state on the device (pf_track_frame) -----------------------------------------
    def track_frame(self, image_bgr: np.ndarray, score_thres: float, nms_iou_thres: float, min_face: float, top_k: int,
                    track_iou_thres: float = 0.5, smooth_box: float = 0.3, diff_thres: float = 5.0,
                    planted_rows: Optional[np.ndarray] = None):
        """FaceAna.run(image) for this engine's stream: returns (track boxes float64 [n,4], smoothed landmarks float64
        [n,98,2], scores float32 [n,98], detector_ran).  planted_rows (test instrument): decoded detector
        rows [R,16] that replace the detector network's own output when the gate runs the detector."""
        fptr, fmem, fh, fw, fstride, img = self._frame_args(image_bgr)
        n = C.c_int(0)
        ran = C.c_int(0)
        boxes = np.zeros((top_k, 4), np.float64)
        kps = np.zeros((top_k, 98, 2), np.float64)
        scores = np.zeros((top_k, 98), np.float32)
        outs = (C.byref(n), boxes.ctypes.data_as(C.POINTER(C.c_double)), kps.ctypes.data_as(C.POINTER(C.c_double)),
                scores.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ran))
        if planted_rows is None:
            rc = self.lib.pf_track_frame(self.h, fptr, fmem, fh, fw, fstride,
                                         float(score_thres), float(nms_iou_thres), float(min_face), int(top_k),
                                         float(track_iou_thres), float(smooth_box), float(diff_thres), 0, *outs)
        else:
            pr = np.ascontiguousarray(planted_rows, np.float32)
            rc = self.lib.pf_track_frame_planted(self.h, fptr, fmem, fh, fw, fstride,
                                                 pr.ctypes.data_as(C.POINTER(C.c_float)), pr.shape[0],
                                                 float(score_thres), float(nms_iou_thres), float(min_face), int(top_k),
                                                 float(track_iou_thres), float(smooth_box), float(diff_thres), *outs)
        self._check(rc, "pf_track_frame")
        self._resident_shape = img.shape
        k = n.value
        return boxes[:k].copy(), kps[:k].copy(), scores[:k].copy(), bool(ran.value)

    def track_reset(self):
        self._check(self.lib.pf_track_reset(self.h), "pf_track_reset")
        self._resident_shape = None

    def set_option(self, option: int, value: int):
        """PF_OPT_HIP_GRAPH (1): replay device-resident run_frames calls from a captured hipGraph."""
        self._check(self.lib.pf_set_option(self.h, int(option), int(value)), "pf_set_option")
        self.__dict__.setdefault("_options", {})[int(option)] = int(value)

    # ---- profiling ----------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._check(self.lib.pf_profile_enable(self.h, 1 if on else 0), "pf_profile_enable")

    def profile_fetch(self):
        cap = 256
        names = C.create_string_buffer(16384)
        ms = (C.c_float * cap)()
        cnt = (C.c_int * cap)()
        n = C.c_int(0)
        self._check(self.lib.pf_profile_fetch(self.h, names, 16384, ms, cnt, cap, C.byref(n)), "pf_profile_fetch")
        tags = [t for t in names.value.decode().split("\n") if t]
        return {tags[k]: (float(ms[k]), int(cnt[k])) for k in range(n.value)}


class BatchEngine:
    """``lanes`` engines (one HIP stream + one arena each) on ONE GPU behind one call: ``pf_batch_*`` of the C ABI.
    Every ``run_frames*`` call hands its frames to the lanes as contiguous slices; launches are asynchronous, so the lanes'
    kernels overlap on the device.  The reference has no batch path (face_landmark.py:119); this is the configuration
    ``bench.py`` measures."""

    def __init__(self, device: int = 0, lanes: int = 3, library: Optional[str] = None):
        self.lib = load_library(library)
        self.b = C.c_void_p()
        if self.lib.pf_batch_create(int(device), int(lanes), C.byref(self.b)) != 0:
            msg = self.lib.pf_batch_last_error(None)
            raise PeppaHipError("pf_batch_create failed: " + (msg.decode() if msg else "unknown"))
        self.device, self.lanes = device, int(lanes)
        self._lane_engines = {}
        self._host = Engine.__new__(Engine)          # page-locked allocations only (pf_host_alloc needs no handle)
        self._host.lib, self._host.h, self._host._borrowed = self.lib, None, True

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.pf_batch_last_error(self.b)
            raise PeppaHipError(f"{what} failed: " + (msg.decode() if msg else "unknown"))

    def lane(self, i: int) -> "Engine":
        """Lane ``i`` as an ``Engine`` (profiling, stage-level calls); the handle stays owned by the batch."""
        if i not in self._lane_engines:
            h = self.lib.pf_batch_lane(self.b, int(i))
            if not h:
                raise PeppaHipError("pf_batch_lane(%d): no such lane" % i)
            e = Engine.__new__(Engine)
            e.lib, e.h, e.device, e._programs, e._borrowed = self.lib, C.c_void_p(h), self.device, {}, True
            self._lane_engines[i] = e
        return self._lane_engines[i]

    def front(self) -> "Engine":
        """The engine that runs letterbox + detector + NMS of a whole call (``PF_OPT_BATCH_FRONT``), for profiling."""
        if "front" not in self._lane_engines:
            e = Engine.__new__(Engine)
            e.lib, e.h, e.device, e._programs, e._borrowed = self.lib, C.c_void_p(self.lib.pf_batch_front(self.b)), self.device, {}, True
            self._lane_engines["front"] = e
        return self._lane_engines["front"]

    def pinned_empty(self, shape, dtype=np.uint8) -> np.ndarray:
        return self._host.pinned_empty(shape, dtype)

    def load_program(self, slot: int, blob: bytes, max_batch_per_lane: int):
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self.lib.pf_batch_load_program(self.b, slot, C.cast(buf, C.c_void_p), len(blob), int(max_batch_per_lane)),
                    "pf_batch_load_program")

    def set_option(self, option: int, value: int):
        self._check(self.lib.pf_batch_set_option(self.b, int(option), int(value)), "pf_batch_set_option")

    def sync(self):
        self._check(self.lib.pf_batch_sync(self.b), "pf_batch_sync")

    def run_frames_device(self, d_frames: int, F: int, H: int, W: int, score_thres: float, iou_thres: float, min_face: float,
                          top_k: int, d_planted: int = 0, rows: int = 0, d_counts: int = 0, d_boxes: int = 0, d_kps: int = 0,
                          d_scores: int = 0, out_mem: int = PF_MEM_DEVICE):
        """Device-resident frames in; results to device buffers or page-locked host buffers (``out_mem``), no synchronisation."""
        rc = self.lib.pf_batch_run_frames(self.b, _ptr(d_frames), PF_MEM_DEVICE, F, H, W, _ptr(d_planted) if d_planted else None,
                                          rows, score_thres, iou_thres, min_face, top_k, _ptr(d_counts) if d_counts else None,
                                          _ptr(d_boxes) if d_boxes else None, _ptr(d_kps) if d_kps else None,
                                          _ptr(d_scores) if d_scores else None, out_mem)
        self._check(rc, "pf_batch_run_frames")

    def run_frames_host_async(self, frames: np.ndarray, d_planted: int, rows: int, score_thres: float, iou_thres: float,
                              min_face: float, top_k: int, d_counts: int, d_boxes: int, d_kps: int, d_scores: int):
        """Host-resident frames (ideally from pinned_empty) in, device-resident results out, no synchronisation: every lane
        copies its slice on its own stream, so one lane's copy overlaps the others' kernels.  Planted detector rows
        (benchmark instrument) are a device pointer."""
        F, H, W, _ = frames.shape
        assert frames.dtype == np.uint8 and frames.flags["C_CONTIGUOUS"]
        rc = self.lib.pf_batch_run_frames(self.b, _ptr(frames), PF_MEM_HOST | PF_MEM_ROWS_DEVICE, F, H, W,
                                          _ptr(d_planted) if d_planted else None, rows, score_thres, iou_thres, min_face, top_k,
                                          _ptr(d_counts), _ptr(d_boxes), _ptr(d_kps), _ptr(d_scores), PF_MEM_DEVICE)
        self._check(rc, "pf_batch_run_frames")

    def run_frames(self, frames: np.ndarray, score_thres: float, iou_thres: float, min_face: float, top_k: int,
                   planted_rows: Optional[np.ndarray] = None):
        """Host frames ``[F,H,W,3]`` uint8 in, numpy results out (synchronous): counts [F], boxes [F,top_k,4],
        landmarks [F,top_k,98,2], scores [F,top_k,98]; rows of a frame beyond its count are undefined."""
        frames = np.ascontiguousarray(frames, np.uint8)
        F, H, W, _ = frames.shape
        counts = np.zeros((F,), np.int32)
        boxes = np.zeros((F, top_k, 4), np.float32)
        kps = np.zeros((F, top_k, 98, 2), np.float32)
        scores = np.zeros((F, top_k, 98), np.float32)
        rows, pr = 0, None
        if planted_rows is not None:
            pr = np.ascontiguousarray(planted_rows, np.float32)
            rows = pr.shape[1]
        rc = self.lib.pf_batch_run_frames(self.b, _ptr(frames), PF_MEM_HOST, F, H, W, _ptr(pr), rows, score_thres, iou_thres,
                                          min_face, top_k, _ptr(counts), _ptr(boxes), _ptr(kps), _ptr(scores), PF_MEM_HOST)
        self._check(rc, "pf_batch_run_frames")
        return counts, boxes, kps, scores

    def close(self):
        self._host.close()
        for e in self._lane_engines.values():
            e.close()
        self._lane_engines = {}
        if getattr(self, "b", None) is not None and self.b:
            self.lib.pf_batch_destroy(self.b)
            self.b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
