"""Video ingest for the FaceAna loop: ``cv2.VideoCapture(path)`` of the reference's demo (demo.py:13-17,44-60) for Motion-JPEG AVI
files, with the frames decoded INTO DEVICE MEMORY by the engine's JPEG decoder (``pf_decode_jpeg`` / ``pf_decode_jpeg_batch``).

The reference hands every container to OpenCV / FFmpeg on the host.  This image has neither a codec library nor a video engine
binding (no rocDecode), so inter-frame codecs (H.264 & co.) stay out of reach; Motion-JPEG -- what USB cameras and many capture
tools write -- is a sequence of independent baseline JPEG images in a RIFF container, and those the engine decodes itself:
``MJPEGCapture`` walks the ``movi`` list, completes frames that rely on the MJPG convention of omitting the Huffman tables
(ITU T.81 Annex K.3 tables, inserted in front of the scan) and feeds them to the decoder.  ``read()`` mirrors
``VideoCapture.read()`` (one frame, a ``DeviceFrame`` that ``FaceAna.run`` accepts like an array); ``read_batch(n)`` decodes n
frames in one call for ``FrameBatchRunner`` / ``pf_run_frames``.  numpy + struct only."""
from __future__ import annotations

import struct
from typing import List, Optional, Tuple

import numpy as np

from . import _native

# ---- ITU T.81 Annex K.3: the "typical" Huffman tables every baseline encoder ships and MJPG streams leave out ------------------------
_DC_LUM_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
_DC_CHR_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
_DC_VALS = list(range(12))
_AC_LUM_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D]
_AC_LUM_VALS = [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xA1,
    0x08, 0x23, 0x42, 0xB1, 0xC1, 0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA,
    0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6,
    0xD7, 0xD8, 0xD9, 0xDA, 0xE1, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9,
    0xFA]
_AC_CHR_BITS = [0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77]
_AC_CHR_VALS = [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xA1, 0xB1, 0xC1, 0x09, 0x23, 0x33, 0x52, 0xF0, 0x15, 0x62, 0x72, 0xD1, 0x0A, 0x16, 0x24, 0x34, 0xE1, 0x25, 0xF1, 0x17, 0x18, 0x19,
    0x1A, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8,
    0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4,
    0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9,
    0xFA]


def standard_dht_segment() -> bytes:
    """One DHT marker segment holding the four Annex K.3 tables (classes/ids 0x00, 0x10, 0x01, 0x11), as encoders write them."""
    body = b""
    for tc_th, bits, vals in ((0x00, _DC_LUM_BITS, _DC_VALS), (0x10, _AC_LUM_BITS, _AC_LUM_VALS),
                              (0x01, _DC_CHR_BITS, _DC_VALS), (0x11, _AC_CHR_BITS, _AC_CHR_VALS)):
        assert sum(bits) == len(vals)
        body += bytes([tc_th]) + bytes(bits) + bytes(vals)
    return b"\xFF\xC4" + struct.pack(">H", len(body) + 2) + body


def complete_mjpeg_frame(jpeg: bytes) -> bytes:
    """An MJPG frame as a self-contained JPEG file: frames without any DHT segment get the standard tables in front of the
    first scan; everything else is returned as it is."""
    if jpeg[:2] != b"\xFF\xD8":
        raise ValueError("not a JPEG frame (no SOI marker)")
    pos = 2
    while pos + 4 <= len(jpeg):
        if jpeg[pos] != 0xFF:
            raise ValueError("corrupt JPEG header")
        m = jpeg[pos + 1]
        if m == 0xFF:                       # fill byte
            pos += 1
            continue
        if m == 0xC4:
            return jpeg                     # has its own tables
        if m == 0xDA:                       # start of scan and no DHT so far
            return jpeg[:pos] + standard_dht_segment() + jpeg[pos:]
        if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
            pos += 2
            continue
        pos += 2 + struct.unpack(">H", jpeg[pos + 2:pos + 4])[0]
    raise ValueError("JPEG frame without a scan")


def _chunks(buf: memoryview, start: int, end: int):
    pos = start
    while pos + 8 <= end:
        cid = bytes(buf[pos:pos + 4])
        size = struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        yield cid, pos + 8, size
        pos += 8 + size + (size & 1)


def parse_mjpeg_avi(data: bytes) -> Tuple[dict, List[Tuple[int, int]]]:
    """(info, [(offset, size) of every video frame]) of a RIFF AVI file whose first video stream is Motion-JPEG."""
    buf = memoryview(data)
    if len(data) < 12 or bytes(buf[0:4]) != b"RIFF" or bytes(buf[8:12]) != b"AVI ":
        raise ValueError("not a RIFF AVI file")
    info = {"width": 0, "height": 0, "fps": 0.0, "frames": 0, "fourcc": ""}
    frames: List[Tuple[int, int]] = []
    vid_id = [b"00"]          # chunk-id prefix of the first 'vids' stream: its position among the header's strl lists ("01dc" when audio comes first)
    n_strl = 0

    def walk_movi(start, end):
        for cid, off, size in _chunks(buf, start, end):
            if cid == b"LIST" and bytes(buf[off:off + 4]) == b"rec ":
                walk_movi(off + 4, off + size)
            elif cid[2:4] in (b"dc", b"db") and cid[0:2] == vid_id[0] and size > 0:
                frames.append((off, size))

    for cid, off, size in _chunks(buf, 12, len(data)):
        if cid != b"LIST":
            continue
        kind = bytes(buf[off:off + 4])
        if kind == b"hdrl":
            for c2, o2, s2 in _chunks(buf, off + 4, off + size):
                if c2 == b"avih" and s2 >= 40:
                    usec, = struct.unpack("<I", buf[o2:o2 + 4])
                    total, = struct.unpack("<I", buf[o2 + 16:o2 + 20])
                    w, h = struct.unpack("<II", buf[o2 + 32:o2 + 40])
                    info.update(width=int(w), height=int(h), frames=int(total), fps=(1e6 / usec if usec else 0.0))
                elif c2 == b"LIST" and bytes(buf[o2:o2 + 4]) == b"strl":
                    stream_no, n_strl = n_strl, n_strl + 1
                    if info["fourcc"]:
                        continue                      # a video stream has been found already
                    is_video = False
                    for c3, o3, s3 in _chunks(buf, o2 + 4, o2 + s2):
                        if c3 == b"strh" and s3 >= 8:
                            is_video = bytes(buf[o3:o3 + 4]) == b"vids"
                            if is_video:
                                vid_id[0] = b"%02d" % stream_no
                        elif c3 == b"strf" and is_video and s3 >= 20:
                            info["fourcc"] = bytes(buf[o3 + 16:o3 + 20]).decode("latin1")
        elif kind == b"movi":
            walk_movi(off + 4, off + size)
    if info["fourcc"].upper() not in ("MJPG", "JPEG", "AVRN", "LJPG") and frames:
        if bytes(buf[frames[0][0]:frames[0][0] + 2]) != b"\xFF\xD8":
            raise ValueError("AVI video stream is '%s', not Motion-JPEG: only MJPG streams have a device decoder (no codec "
                             "library in this environment)" % info["fourcc"])
    info["frames"] = len(frames)
    return info, frames


class MJPEGCapture:
    """``cv2.VideoCapture`` for Motion-JPEG AVI files (demo.py:13-17): ``isOpened()``, ``read() -> (ok, frame)``, ``get(prop)``,
    ``release()``.  With an engine (``_native.Engine``, e.g. ``FaceAna().engine``) frames are ``DeviceFrame`` objects decoded into
    device memory (``FaceAna.run`` takes them like arrays); ``read_batch(n)`` decodes n frames with one ``pf_decode_jpeg_batch``
    call and returns ``(device pointer, n, H, W)`` for ``run_frames_device``."""

    CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT, CAP_PROP_POS_FRAMES = 3, 4, 5, 7, 1

    def __init__(self, path_or_bytes, engine: Optional["_native.Engine"] = None, want_host: bool = True):
        self.engine, self.want_host = engine, want_host
        self._data = b""
        self._frames: List[Tuple[int, int]] = []
        self.info = {}
        self._pos = 0
        self.error = ""
        try:
            data = bytes(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
            self.info, self._frames = parse_mjpeg_avi(data)
            self._data = data
        except (OSError, ValueError) as e:
            self.error = str(e)             # like cv2: a capture that did not open, not an exception

    def isOpened(self) -> bool:
        return bool(self._frames)

    def get(self, prop: int) -> float:
        return float({self.CAP_PROP_FRAME_WIDTH: self.info.get("width", 0), self.CAP_PROP_FRAME_HEIGHT: self.info.get("height", 0),
                      self.CAP_PROP_FPS: self.info.get("fps", 0.0), self.CAP_PROP_FRAME_COUNT: len(self._frames),
                      self.CAP_PROP_POS_FRAMES: self._pos}.get(prop, 0.0))

    def frame_bytes(self, i: int) -> bytes:
        off, size = self._frames[i]
        return complete_mjpeg_frame(self._data[off:off + size])

    def read(self):
        """``(ok, frame)`` like ``cv2.VideoCapture.read``: a truncated or corrupt frame is ``(False, None)`` with the reason in
        ``self.error`` -- never an exception -- and the stream position moves past it."""
        if self._pos >= len(self._frames):
            return False, None
        i = self._pos
        self._pos += 1
        try:
            jpeg = self.frame_bytes(i)
            if self.engine is None:
                return True, jpeg           # no engine: the self-contained JPEG bytes (FaceAna.imread takes them)
            return True, self.engine.imread(jpeg, self.want_host)
        except (ValueError, _native.PeppaHipError) as e:
            self.error = "frame %d: %s" % (i, e)
            return False, None

    def read_batch(self, n: int, threads: int = 4):
        """Up to n frames, decoded into device memory by ONE pf_decode_jpeg_batch call: (device pointer [k][H][W][3] BGR, k, H, W),
        or None at the end of the file."""
        if self.engine is None:
            raise _native.PeppaHipError("MJPEGCapture.read_batch needs an engine")
        k = min(n, len(self._frames) - self._pos)
        if k <= 0:
            return None
        first = self._pos
        self._pos += k
        try:
            files = [self.frame_bytes(first + i) for i in range(k)]
            return self.engine.decode_jpeg_batch(files, threads)
        except (ValueError, _native.PeppaHipError) as e:       # one bad frame spoils the batch: report it like the end of the stream
            self.error = "frames %d..%d: %s" % (first, first + k - 1, e)
            return None

    def release(self):
        self._data, self._frames, self._pos = b"", [], 0


def write_mjpeg_avi(jpegs: List[bytes], width: int, height: int, fps: float = 25.0) -> bytes:
    """A minimal RIFF AVI file (one MJPG video stream, idx1 index) around ready-made JPEG frames -- for tests and for tools that
    want to hand still frames to the video path."""
    def chunk(cid: bytes, body: bytes) -> bytes:
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    def lst(kind: bytes, body: bytes) -> bytes:
        return chunk(b"LIST", kind + body)
    n = len(jpegs)
    biggest = max((len(j) for j in jpegs), default=0)
    avih = struct.pack("<14I", int(round(1e6 / fps)), int(biggest * fps), 0, 0x10, n, 0, 1, biggest, width, height, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1000, int(round(fps * 1000)), 0, n, biggest, 0xFFFFFFFF, 0) + struct.pack("<4h", 0, 0, width, height)
    strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, 24, b"MJPG", width * height * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx = b"", b""
    for j in jpegs:
        idx += b"00dc" + struct.pack("<III", 0x10, 4 + len(movi_body), len(j))
        movi_body += chunk(b"00dc", j)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    return b"RIFF" + struct.pack("<I", len(body)) + body
