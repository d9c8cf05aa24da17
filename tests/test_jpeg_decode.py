"""pf_decode_jpeg (frame ingest, SURVEY 8 next-row N2: cv2.imread at demo.py:76) against libjpeg's own decoder through PIL --
the same library and defaults cv2.imread uses (JDCT_ISLOW, fancy upsampling): the device decoder must be BIT-identical, for
every sampling grid it accepts, odd sizes (partial MCUs, replicated edge rows / columns), tiny images (the plain-replication
rule below three chroma columns), restart intervals, non-default quantisation and optimised Huffman tables, greyscale."""
import io
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")

from peppa_pig_face_landmark_amd import _native
from peppa_pig_face_landmark_amd.synth import make_frame


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    frame, _ = make_frame(max(h, 64), max(w, 64), 2, seed=seed)
    img = frame[:h, :w].astype(np.int16) + rng.integers(-12, 13, (h, w, 3))     # texture: every AC coefficient gets used
    return np.clip(img, 0, 255).astype(np.uint8)


def _encode(bgr, **kw):
    buf = io.BytesIO()
    PIL.fromarray(bgr[..., ::-1] if bgr.ndim == 3 else bgr).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _pil_decode(data):
    im = PIL.open(io.BytesIO(data))
    a = np.asarray(im.convert("RGB") if im.mode != "L" else im)
    return a[..., ::-1] if a.ndim == 3 else np.repeat(a[..., None], 3, axis=2)


CASES = [  # (h, w, save kwargs)
    (64, 96, dict(quality=90, subsampling=0)),
    (64, 96, dict(quality=90, subsampling=1)),
    (64, 96, dict(quality=90, subsampling=2)),
    (67, 101, dict(quality=75, subsampling=2)),            # partial MCUs, odd chroma edge
    (35, 53, dict(quality=60, subsampling=1)),
    (33, 47, dict(quality=95, subsampling=0, optimize=True)),
    (8, 5, dict(quality=85, subsampling=2)),               # three chroma columns
    (6, 4, dict(quality=85, subsampling=2)),               # two chroma columns: plain replication
    (1, 1, dict(quality=85, subsampling=2)),
    (120, 160, dict(quality=30, subsampling=2, optimize=True)),
    (120, 160, dict(quality=100, subsampling=2)),
]


def _check_cases(eng):
    for h, w, kw in CASES:
        img = _image(h, w, seed=h * 1000 + w)
        data = _encode(img, **kw)
        ref = _pil_decode(data)
        info = eng.jpeg_info(data)
        assert info[:2] == (h, w) and info[3] == {0: 444, 1: 422, 2: 420}[kw["subsampling"]]
        d, hh, ww, got = eng.decode_jpeg(data)
        assert (hh, ww) == (h, w) and d
        assert np.array_equal(got, ref), (h, w, kw, int(np.abs(got.astype(int) - ref).max()))
    grey = _image(50, 70, seed=9)[..., 1]
    data = _encode(grey, quality=80)
    _, _, _, got = eng.decode_jpeg(data)
    assert eng.jpeg_info(data)[2:] == (1, 0) and np.array_equal(got, _pil_decode(data))
    # restart intervals (Pillow >= 9.4 writes DRI on request; older versions ignore the keyword and the case is a repeat)
    img = _image(72, 88, seed=5)
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):
        try:
            data = _encode(img, quality=85, subsampling=2, **kw)
        except TypeError:
            continue
        _, _, _, got = eng.decode_jpeg(data)
        assert np.array_equal(got, _pil_decode(data)), kw
    # a batch of equally shaped files, entropy-decoded on several host threads
    imgs = [_image(40, 56, seed=40 + i) for i in range(5)]
    files = [_encode(im, quality=70 + 5 * i, subsampling=2) for i, im in enumerate(imgs)]
    singles = [eng.decode_jpeg(f)[3] for f in files]
    for f, one in zip(files, singles):
        assert np.array_equal(one, _pil_decode(f))
    d, n, hh, ww = eng.decode_jpeg_batch(files, threads=3)
    assert (n, hh, ww) == (5, 40, 56) and d
    for i, one in enumerate(singles):          # frame i of the batch, read through the detector's pre-processing kernel
        got = eng.letterbox(_native.DeviceFrame(d + i * hh * ww * 3, hh, ww), (48, 64))
        ref = eng.letterbox(one, (48, 64))
        assert np.array_equal(got[0], ref[0]), i
    with pytest.raises(_native.PeppaHipError, match="differs in size"):
        eng.decode_jpeg_batch([files[0], _encode(_image(48, 56, seed=1), quality=80, subsampling=2)])
    # files with restart markers: the Huffman stream itself can be decoded on the device (one thread per restart interval; chosen
    # automatically for big batches, forced here)
    eng.set_option(_native.PF_OPT_JPEG_ENTROPY, 2)
    eng.profile_enable(True)
    for (h, w), kw in (((67, 101), dict(subsampling=2, restart_marker_blocks=2)), ((64, 96), dict(subsampling=0, restart_marker_rows=1)),
                       ((35, 53), dict(subsampling=1, restart_marker_blocks=5)), ((50, 70), dict(subsampling=2, restart_marker_blocks=1, optimize=True)),
                       ((120, 160), dict(subsampling=2, restart_marker_rows=2, quality=35))):
        img = _image(h, w, seed=h + w)
        data = _encode(img, **dict(dict(quality=88), **kw))
        assert b"\xff\xdd" in data
        _, _, _, got = eng.decode_jpeg(data)
        assert np.array_equal(got, _pil_decode(data)), (h, w, kw)
    grey = _image(40, 72, seed=19)[..., 1]
    data = _encode(grey, quality=80, restart_marker_blocks=3)
    assert np.array_equal(eng.decode_jpeg(data)[3], _pil_decode(data))
    names = list(eng.profile_fetch())
    assert "jpeg_huffman" in names, names
    eng.profile_enable(False)
    # files WITHOUT restart markers (what cameras write): the stream is cut into 1024-bit sub-sequences that synchronise themselves
    # on the device (csrc/k_jpeg.h jpeg_sync_kernel); quality 100 noise has 64-coefficient blocks without end-of-block codes, the
    # slowest case to synchronise (threads walk on through the following sub-sequences)
    eng.set_option(_native.PF_OPT_JPEG_ENTROPY, 0)
    eng.profile_enable(True)
    eng.profile_fetch()
    for (h, w), kw in (((120, 160), dict(quality=100, subsampling=2)), ((64, 64), dict(quality=100, subsampling=0)),
                       ((97, 131), dict(quality=92, subsampling=1)), ((72, 104), dict(quality=30, subsampling=2))):
        data = _encode(_image(h, w, seed=3 * h + w), **kw)
        assert b"\xff\xdd" not in data
        _, _, _, got = eng.decode_jpeg(data)
        assert np.array_equal(got, _pil_decode(data)), (h, w, kw)
    for (h, w), q in (((200, 300), 80), ((203, 301), 95)):       # greyscale: one block per MCU; odd width = the per-pixel colour kernel
        data = _encode(_image(h, w, seed=h)[..., 1], quality=q)
        _, _, _, got = eng.decode_jpeg(data)
        assert np.array_equal(got, _pil_decode(data)), (h, w, q)
    names = list(eng.profile_fetch())
    assert "jpeg_subseq" in names and "jpeg_unpack" not in names, names
    # a stream that needs more rounds than were queued is NOT passed on: the synchronous call falls back to the host decoder
    eng.set_option(_native.PF_OPT_JPEG_SYNC_ROUNDS, 1)
    data = _encode(_image(120, 160, seed=520), quality=100, subsampling=2)
    _, _, _, got = eng.decode_jpeg(data)
    assert np.array_equal(got, _pil_decode(data))
    assert "jpeg_unpack" in list(eng.profile_fetch())
    # ... and the asynchronous batch call reports it at the next synchronisation
    eng.decode_jpeg_batch([data, data], threads=1)
    with pytest.raises(_native.PeppaHipError, match="did not synchronise"):
        eng.sync()
    eng.set_option(_native.PF_OPT_JPEG_SYNC_ROUNDS, 0)
    eng.profile_enable(False)
    eng.set_option(_native.PF_OPT_JPEG_ENTROPY, 2)
    # a batch that mixes both kinds: each file takes its own route
    imgs = [_image(48, 80, seed=60 + i) for i in range(4)]
    files = [_encode(im, quality=85, subsampling=2, **(dict(restart_marker_rows=1) if i % 2 else {})) for i, im in enumerate(imgs)]
    refs = [_pil_decode(f) for f in files]
    d, n, hh, ww = eng.decode_jpeg_batch(files, threads=2)
    for i, ref in enumerate(refs):
        got = eng.letterbox(_native.DeviceFrame(d + i * hh * ww * 3, hh, ww), (48, 80))
        assert np.array_equal(got[0], eng.letterbox(ref, (48, 80))[0]), i
    eng.set_option(_native.PF_OPT_JPEG_ENTROPY, 0)
    # refused, not approximated
    with pytest.raises(_native.PeppaHipError, match="progressive"):
        eng.decode_jpeg(_encode(img, quality=85, progressive=True))
    with pytest.raises(_native.PeppaHipError):
        eng.decode_jpeg(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    # the decoded frame is a device frame like any other: detector pre-processing reads it where it lies
    data = _encode(_image(270, 480, seed=3), quality=90, subsampling=2)
    frame = eng.imread(data)
    lb_dev, lb_host = eng.letterbox(frame), eng.letterbox(frame.numpy())
    assert np.array_equal(lb_dev[0], lb_host[0]) and np.array_equal(lb_dev[1], lb_host[1])


def _corrupt_scan_case(library):
    """Damaged entropy-coded data (no marker damage, so the file still takes the device's sub-sequence decoder): every thread's
    decode is bounded by its bit range and every block address by the block count -- the call returns (pixels are garbage, like
    libjpeg's 'corrupt data' warning path) or reports that the stream did not synchronise; it never hangs or writes out of range."""
    eng = _native.Engine(0, library)
    try:
        rng = np.random.default_rng(99)
        data = bytearray(_encode(_image(72, 104, seed=5), quality=80, subsampling=2))
        sos = bytes(data).find(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        for trial in range(6):
            bad = bytearray(data)
            for pos in rng.integers(start, len(bad) - 2, 25):
                bad[pos] = int(rng.integers(0, 255))          # never 0xFF: no new markers
            if trial == 5:
                bad = bad[:start + (len(bad) - start) // 2] + b"\xff\xd9"      # truncated scan
            try:
                _, h, w, got = eng.decode_jpeg(bytes(bad))
                assert got.shape == (72, 104, 3)
            except _native.PeppaHipError:
                pass
        _, _, _, got = eng.decode_jpeg(bytes(data))            # the engine is still usable
        assert np.array_equal(got, _pil_decode(bytes(data)))
    finally:
        eng.close()


def test_corrupt_scan_terminates_emulator(emu_library):
    _corrupt_scan_case(emu_library)


@pytest.mark.gpu
def test_corrupt_scan_terminates_gpu(hip_library):
    _corrupt_scan_case(hip_library)


def test_decode_matches_libjpeg_emulator(emu_library):
    eng = _native.Engine(0, emu_library)
    try:
        _check_cases(eng)
    finally:
        eng.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/figure/test1.jpg"), reason="reference checkout not present (GPU box)")
def test_reference_sample_image_emulator(emu_library):
    """The reference's own sample (figure/test1.jpg), as demo.py would cv2.imread it."""
    data = open("/root/reference/figure/test1.jpg", "rb").read()
    eng = _native.Engine(0, emu_library)
    try:
        _, h, w, got = eng.decode_jpeg(data)
        assert np.array_equal(got, _pil_decode(data)) and got.shape == (h, w, 3)
    finally:
        eng.close()


@pytest.mark.gpu
def test_decode_matches_libjpeg_gpu(hip_library):
    eng = _native.Engine(0, hip_library)
    try:
        _check_cases(eng)
        img = _image(1080, 1920, seed=77)
        data = _encode(img, quality=90, subsampling=2)
        _, _, _, got = eng.decode_jpeg(data)
        assert np.array_equal(got, _pil_decode(data))
        # the sub-sequence decoder's rounds race by design (a record may be rewritten while its successor reads it; the work lists
        # make somebody check again afterwards): slow-to-synchronise streams, many sub-sequences, many repetitions
        for (h, w), kw in (((480, 640), dict(quality=100, subsampling=2)), ((360, 480), dict(quality=100, subsampling=0)),
                           ((1080, 1920), dict(quality=97, subsampling=2))):
            data = _encode(_image(h, w, seed=h + w), **kw)
            ref = _pil_decode(data)
            eng.profile_enable(True)
            eng.profile_fetch()
            for rep in range(12):
                _, _, _, got = eng.decode_jpeg(data)
                assert np.array_equal(got, ref), (h, w, kw, rep)
            names = list(eng.profile_fetch())
            eng.profile_enable(False)
            assert "jpeg_subseq" in names, names
    finally:
        eng.close()


def _facade_case(library, student_weights, detector_weights):
    """demo.py:76-86 with the engine's ingest: FaceAna.run(facer.imread(jpg)) equals FaceAna.run(<libjpeg-decoded array>)."""
    from Skps import FaceAna
    from peppa_pig_face_landmark_amd.core.api.facer import get_cfg
    cfg = get_cfg()
    cfg["Skps"]["Detect"]["input_shape"] = [96, 160, 3]
    cfg["Skps"]["Keypoints"]["input_shape"] = [64, 64, 3]
    cfg["Skps"]["Engine"]["dtype"] = "f32"
    frame, _ = make_frame(270, 480, 3, seed=11, face_w=330, face_h=430)
    data = _encode(frame, quality=92, subsampling=2)
    a = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": student_weights}, library=library)
    b = FaceAna(cfg=cfg, weights={"detector": detector_weights, "keypoints": student_weights}, library=library)
    try:
        dev = a.imread(data)
        assert dev.shape == (270, 480, 3) and np.array_equal(dev.numpy(), _pil_decode(data))
        ra = a.run(dev)
        rb = b.run(np.ascontiguousarray(_pil_decode(data)))
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            for key in ("box", "kps", "scores"):
                assert np.array_equal(np.asarray(x[key]), np.asarray(y[key])), key
        a.reset()
        dev2 = a.imread(data, want_host=False)          # second decode reuses the buffers; no host copy requested
        assert len(a.run(dev2)) == len(ra)
    finally:
        a.engine.close()
        b.engine.close()


def test_faceana_runs_on_decoded_device_frame_emulator(emu_library, student_weights, detector_weights):
    _facade_case(emu_library, student_weights, detector_weights)


@pytest.mark.gpu
def test_faceana_runs_on_decoded_device_frame_gpu(hip_library, student_weights, detector_weights):
    _facade_case(hip_library, student_weights, detector_weights)


def _malformed(library, trials):
    rng = np.random.default_rng(2024)
    base = [_encode(_image(40, 56, seed=70), quality=85, subsampling=2),
            _encode(_image(33, 47, seed=71), quality=60, subsampling=0, optimize=True),
            _encode(_image(48, 64, seed=72), quality=85, subsampling=2, restart_marker_blocks=2),
            _encode(_image(24, 40, seed=73)[..., 0], quality=80)]
    eng = _native.Engine(0, library)
    outcomes = {"ok": 0, "error": 0}
    try:
        for trial in range(trials):
            data = bytearray(base[trial % len(base)])
            kind = trial % 3
            if kind == 0:                              # truncation
                data = data[:int(rng.integers(2, len(data)))]
            elif kind == 1:                            # a few flipped bytes anywhere (headers, tables, scan, markers)
                for _ in range(int(rng.integers(1, 6))):
                    data[int(rng.integers(2, len(data)))] = int(rng.integers(0, 256))
            else:                                      # corrupted segment lengths / marker codes in the header region
                i = int(rng.integers(2, min(len(data), 600)))
                data[i] = int(rng.integers(0, 256))
                data = data[:int(rng.integers(len(data) // 2, len(data) + 1))]
            eng.set_option(_native.PF_OPT_JPEG_ENTROPY, 2 if trial % 2 else 1)
            try:
                _, h, w, got = eng.decode_jpeg(bytes(data))
                assert got.shape == (h, w, 3)
                outcomes["ok"] += 1
            except _native.PeppaHipError:
                outcomes["error"] += 1
    finally:
        eng.close()
    assert outcomes["ok"] > trials // 12 and outcomes["error"] > trials // 12, outcomes


def test_malformed_files_fail_cleanly_emulator(emu_library):
    """Ingest reads untrusted bytes: truncated, bit-flipped and length-corrupted files must end in a decoded frame or a
    PeppaHipError -- never in a crash or a hang (host parser / Huffman decoder bounds; both entropy routes)."""
    _malformed(emu_library, 240)


@pytest.mark.gpu
def test_malformed_files_fail_cleanly_gpu(hip_library):
    _malformed(hip_library, 120)


# ---- over-subscribed Huffman tables (DHT) ----------------------------------------------------------------------------------
def _dht_file(counts, tc_th=0x00):
    """SOI + one DHT segment whose 16 code-length counts are `counts` (+ that many symbol bytes) + EOI."""
    nsym = min(sum(counts), 256)
    body = bytes([tc_th]) + bytes(counts) + bytes(i & 0xFF for i in range(nsym))
    return b"\xff\xd8" + b"\xff\xc4" + (len(body) + 2).to_bytes(2, "big") + body + b"\xff\xd9"


def _oversubscribed_dhts():
    files = []
    one = [0] * 16
    one[0] = 255                                   # 255 codes of length 1 (two exist): the advisor's 130 KB overrun
    files.append(_dht_file(one))
    for l in range(1, 10):                         # lengths 1..9 index the 9-bit lookahead table
        for extra in (1, 3):
            c = [0] * 16
            c[l - 1] = min((1 << l) + extra, 255)
            files.append(_dht_file(c))
            files.append(_dht_file(c, tc_th=0x10))  # the same counts as an AC table
    c = [0] * 16
    c[0], c[1] = 2, 1                              # length 1 full, then one more code of length 2
    files.append(_dht_file(c))
    return files


def test_oversubscribed_huffman_tables_are_refused_emulator(emu_library):
    """A DHT whose counts exceed the code space of a length used to index look[] past its end before the check ran
    (round-2 advisor finding, jpeg.inl JpegHuff::build): every such table must be refused, through every entry point that
    parses headers."""
    eng = _native.Engine(0, emu_library)
    try:
        for data in _oversubscribed_dhts():
            with pytest.raises(_native.PeppaHipError):
                eng.jpeg_info(data)
            with pytest.raises(_native.PeppaHipError):
                eng.decode_jpeg(data)
    finally:
        eng.close()


def test_oversubscribed_huffman_tables_under_asan(tmp_path):
    """The same files through an AddressSanitizer build of the emulator flavour, in a subprocess (the sanitizer runtime has to be
    preloaded): a write past look[] aborts the child with an ASan report instead of silently landing in fast_ac / the heap."""
    import subprocess
    import sys
    from tests.simt_emu import build_emu
    rt = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"
    if not (build_emu.available() and os.path.exists(rt)):
        pytest.skip("no host clang / ASan runtime")
    lib = build_emu.build_emu(flavour="asan")
    blob = tmp_path / "dhts.bin"
    files = _oversubscribed_dhts()
    with open(blob, "wb") as f:
        for d in files:
            f.write(len(d).to_bytes(4, "little") + d)
    child = (
        "import ctypes as C, sys\n"
        "lib = C.CDLL(sys.argv[1])\n"
        "lib.pf_jpeg_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4\n"
        "raw = open(sys.argv[2], 'rb').read(); i = 0; n = 0\n"
        "while i < len(raw):\n"
        "    ln = int.from_bytes(raw[i:i + 4], 'little'); d = raw[i + 4:i + 4 + ln]; i += 4 + ln\n"
        "    buf = (C.c_ubyte * ln).from_buffer_copy(d); v = [C.c_int() for _ in range(4)]\n"
        "    assert lib.pf_jpeg_info(buf, ln, *[C.byref(x) for x in v]) != 0\n"
        "    n += 1\n"
        "print('refused', n)\n")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, "-c", child, lib, str(blob)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ("refused %d" % len(files)) in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
