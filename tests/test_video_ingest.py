"""Frame ingest beyond baseline JPEG stills: Motion-JPEG AVI files through the engine's JPEG decoder (the buildable
form of demo.py:13-17 -- no codec library in this image) and the host fallback of FaceAna.imread for everything else cv2.imread
opens (demo.py:76).  CPU tier on the SIMT emulator; the decoder itself is pinned bit-for-bit against libjpeg in test_jpeg_decode.py."""
import ctypes
import io
import types

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from peppa_pig_face_landmark_amd import video  # noqa: E402
from peppa_pig_face_landmark_amd.core.api.facer import FaceAna  # noqa: E402


def _photo(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([120 + 80 * np.sin(xx / (7.0 + c) + seed) * np.cos(yy / (5.0 + c)) for c in range(3)], -1)
    return np.clip(img + rng.normal(0, 4, img.shape), 0, 255).astype(np.uint8)


def _jpeg(rgb, **kw):
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _strip_dht(jpeg: bytes) -> bytes:
    out, pos = bytearray(jpeg[:2]), 2
    while True:
        m = jpeg[pos + 1]
        if m == 0xDA:
            return bytes(out) + jpeg[pos:]
        seg = 2 + int.from_bytes(jpeg[pos + 2:pos + 4], "big")
        if m != 0xC4:
            out += jpeg[pos:pos + seg]
        pos += seg


def test_standard_huffman_tables_are_the_ones_encoders_write():
    """libjpeg's default (non-optimised) tables ARE ITU T.81 K.3: the DHT segments of such a file concatenate to ours."""
    j = _jpeg(_photo(32, 32, 1), quality=85)
    tables, pos = {}, 2
    while j[pos + 1] != 0xDA:
        seg = 2 + int.from_bytes(j[pos + 2:pos + 4], "big")
        if j[pos + 1] == 0xC4:
            body, q = j[pos + 4:pos + seg], 0
            while q < len(body):
                n = sum(body[q + 1:q + 17])
                tables[body[q]] = bytes(body[q:q + 17 + n])
                q += 17 + n
        pos += seg
    ours = video.standard_dht_segment()[4:]
    assert b"".join(tables[k] for k in (0x00, 0x10, 0x01, 0x11)) == ours


def test_mjpeg_avi_frames_decode_bit_exact(emu_engine):
    h, w = 48, 64
    rgbs = [_photo(h, w, s) for s in range(3)]
    jpegs = [_jpeg(r, quality=88, subsampling=2) for r in rgbs]
    frames = [jpegs[0], _strip_dht(jpegs[1]), _strip_dht(jpegs[2])]          # MJPG streams usually omit the tables
    avi = video.write_mjpeg_avi(frames, w, h, fps=30.0)
    cap = video.MJPEGCapture(avi, engine=emu_engine)
    assert cap.isOpened() and cap.get(cap.CAP_PROP_FRAME_COUNT) == 3
    assert (cap.get(cap.CAP_PROP_FRAME_WIDTH), cap.get(cap.CAP_PROP_FRAME_HEIGHT)) == (w, h) and abs(cap.get(cap.CAP_PROP_FPS) - 30.0) < 1e-3
    for i in range(3):
        ok, frame = cap.read()
        assert ok
        ref = np.asarray(Image.open(io.BytesIO(jpegs[i])).convert("RGB"))[:, :, ::-1]      # libjpeg's pixels, BGR like cv2
        assert np.array_equal(frame.numpy(), ref)
    assert cap.read() == (False, None)
    # the batched form: three frames decoded by ONE pf_decode_jpeg_batch call
    cap2 = video.MJPEGCapture(avi, engine=emu_engine)
    d, n, hh, ww = cap2.read_batch(8, threads=2)
    assert (n, hh, ww) == (3, h, w) and cap2.read_batch(8) is None
    emu_engine.sync()
    got = np.ctypeslib.as_array(ctypes.cast(d, ctypes.POINTER(ctypes.c_ubyte)), shape=(3, h, w, 3))      # emulator: device memory is host memory
    for i in range(3):
        assert np.array_equal(got[i], np.asarray(Image.open(io.BytesIO(jpegs[i])).convert("RGB"))[:, :, ::-1])
    cap.release()


def test_capture_of_other_files_does_not_open(tmp_path):
    assert not video.MJPEGCapture(b"RIFF\x04\x00\x00\x00WAVE").isOpened()
    assert not video.MJPEGCapture(str(tmp_path / "missing.avi")).isOpened()
    raw = video.write_mjpeg_avi([b"\x00\x01\x02\x03" * 8], 4, 4).replace(b"MJPG", b"H264")
    cap = video.MJPEGCapture(raw)
    assert not cap.isOpened() and "H264" in cap.error
    with pytest.raises(ValueError):
        video.complete_mjpeg_frame(b"\x00\x00")


def test_imread_falls_back_to_the_host_for_what_the_device_decoder_refuses(emu_engine, tmp_path):
    """FaceAna.imread == cv2.imread(path) (demo.py:76): baseline JPEG -> DeviceFrame; progressive JPEG / PNG -> host-decoded BGR
    array; unreadable -> None."""
    fa = types.SimpleNamespace(engine=emu_engine)
    rgb = _photo(40, 56, 9)
    base = FaceAna.imread(fa, _jpeg(rgb, quality=90))
    assert hasattr(base, "ptr") and base.shape == (40, 56, 3)
    prog = _jpeg(rgb, quality=90, progressive=True)
    got = FaceAna.imread(fa, prog)
    assert isinstance(got, np.ndarray) and np.array_equal(got, np.asarray(Image.open(io.BytesIO(prog)).convert("RGB"))[:, :, ::-1])
    png = io.BytesIO()
    Image.fromarray(rgb).save(png, format="PNG")
    p = tmp_path / "face.png"
    p.write_bytes(png.getvalue())
    assert np.array_equal(FaceAna.imread(fa, str(p)), rgb[:, :, ::-1])
    assert FaceAna.imread(fa, b"not an image at all") is None
    assert FaceAna.imread(fa, str(tmp_path / "nope.jpg")) is None


def test_corrupt_frames_and_second_stream_videos_behave_like_cv2(emu_engine):
    """cv2.VideoCapture.read() answers (False, None) for a frame it cannot decode -- it does not raise -- and reads the video
    stream wherever it sits among the file's streams (round-4 advisor: '00dc' was hard-coded)."""
    h, w = 32, 48
    jpegs = [_jpeg(_photo(h, w, s), quality=85) for s in range(3)]
    broken = [jpegs[0], jpegs[1][:len(jpegs[1]) // 3], b"\x00\x01garbage" * 4]       # truncated scan, no SOI at all
    cap = video.MJPEGCapture(video.write_mjpeg_avi(broken, w, h), engine=emu_engine)
    assert cap.isOpened()
    ok0, f0 = cap.read()
    assert ok0 and f0.shape == (h, w, 3)
    for i in (1, 2):
        ok, frame = cap.read()
        assert (ok, frame) == (False, None) and ("frame %d" % i) in cap.error
    assert cap.get(cap.CAP_PROP_POS_FRAMES) == 3 and cap.read() == (False, None)
    cap2 = video.MJPEGCapture(video.write_mjpeg_avi(broken, w, h), engine=emu_engine)
    assert cap2.read_batch(8) is None and "frames 0..2" in cap2.error
    cap2.release()
    assert cap2.get(cap2.CAP_PROP_POS_FRAMES) == 0 and not cap2.isOpened()
    # the video as stream 01 behind an audio stream: frames are '01dc' chunks
    avi = video.write_mjpeg_avi(jpegs, w, h)
    strl_at = avi.index(b"strl") - 8
    strl_len = 8 + int.from_bytes(avi[strl_at + 4:strl_at + 8], "little")
    audio = bytearray(avi[strl_at:strl_at + strl_len])
    audio[audio.index(b"vids"):audio.index(b"vids") + 4] = b"auds"
    two = bytearray(avi[:strl_at] + bytes(audio) + avi[strl_at:])
    hdrl_at = two.index(b"hdrl") - 8
    for at in (4, hdrl_at + 4):                                                       # RIFF and hdrl LIST sizes grow by the new strl
        two[at:at + 4] = (int.from_bytes(two[at:at + 4], "little") + strl_len).to_bytes(4, "little")
    two = bytes(two).replace(b"00dc", b"01dc")
    cap3 = video.MJPEGCapture(two, engine=emu_engine)
    assert cap3.isOpened() and cap3.get(cap3.CAP_PROP_FRAME_COUNT) == 3
    ok, frame = cap3.read()
    assert ok and np.array_equal(frame.numpy(), np.asarray(Image.open(io.BytesIO(jpegs[0])).convert("RGB"))[:, :, ::-1])


def test_host_imread_applies_exif_orientation_like_cv2(emu_engine):
    """cv2.imread rotates by the EXIF orientation tag; the Pillow fallback has to do the same or rotated progressive files give other boxes."""
    fa = types.SimpleNamespace(engine=emu_engine)
    rgb = _photo(24, 40, 3)
    buf = io.BytesIO()
    exif = Image.Exif()
    exif[0x0112] = 6                      # rotate 90 degrees clockwise to display
    Image.fromarray(rgb).save(buf, format="JPEG", quality=92, progressive=True, exif=exif)
    got = FaceAna.imread(fa, buf.getvalue())
    assert isinstance(got, np.ndarray) and got.shape == (40, 24, 3)
    from PIL import ImageOps
    want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(buf.getvalue()))).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_mjpeg_avi_through_faceana_on_the_gpu(hip_library, student_weights, detector_weights):
    """demo.py:13-17 with the engine's ingest: every frame of a Motion-JPEG AVI decoded on the GPU (read() and read_batch()) is
    bit-identical with libjpeg's pixels, and FaceAna.run(frame) on the DeviceFrame equals FaceAna.run on the decoded array."""
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    from peppa_pig_face_landmark_amd.synth import make_frame
    h, w = 544, 960
    rgbs = [np.ascontiguousarray(make_frame(h, w, 3, seed=70 + i)[0][:, :, ::-1]) for i in range(4)]
    jpegs = [_jpeg(r, quality=90, subsampling=2) for r in rgbs]
    avi = video.write_mjpeg_avi([jpegs[0]] + [_strip_dht(j) for j in jpegs[1:]], w, h, fps=25.0)
    cfg = get_cfg()
    fa = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": student_weights}, library=hip_library)
    cap = video.MJPEGCapture(avi, engine=fa.engine)
    assert cap.isOpened() and int(cap.get(cap.CAP_PROP_FRAME_COUNT)) == 4
    for i in range(4):
        ok, frame = cap.read()
        assert ok
        ref = np.asarray(Image.open(io.BytesIO(jpegs[i])).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(frame.numpy(), ref), i
        fa.reset()
        got = fa.run(frame)
        fa.reset()
        want = fa.run(np.ascontiguousarray(ref))
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(a["box"], b["box"]) and np.array_equal(a["kps"], b["kps"])
    cap2 = video.MJPEGCapture(avi, engine=fa.engine, want_host=False)
    d, n, hh, ww = cap2.read_batch(4, threads=2)
    assert (n, hh, ww) == (4, h, w)
    for i in range(4):
        ref = np.asarray(Image.open(io.BytesIO(jpegs[i])).convert("RGB"))[:, :, ::-1]
        got = fa.engine.letterbox(_native_frame(d + i * h * w * 3, h, w), (384, 640))[0]
        assert np.array_equal(got, fa.engine.letterbox(np.ascontiguousarray(ref), (384, 640))[0]), i
    fa.engine.close()


def _native_frame(ptr, h, w):
    from peppa_pig_face_landmark_amd import _native
    return _native.DeviceFrame(ptr, h, w)
