"""Frame-ingest timing: pf_decode_jpeg (host Huffman + device IDCT / upsampling / colour) against libjpeg on
the host (PIL, the library cv2.imread uses) followed by the upload of the decoded frame.  One 1080p 4:2:0 JPEG, one stream.
usage: python tools/bench_jpeg.py [--quality 90] [--n 50]   -> one JSON line"""
import argparse
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from peppa_pig_face_landmark_amd import _native  # noqa: E402
from peppa_pig_face_landmark_amd.synth import make_frame  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quality", type=int, default=90)
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--content", default="photo", choices=["photo", "noise"])
    ap.add_argument("--restart-rows", type=int, default=0,
                    help="write a restart marker every N MCU rows: such files have their Huffman stream decoded on the device")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    if args.content == "noise":      # worst case for the entropy stage: every block carries ~35 non-zero coefficients
        frame, _ = make_frame(1080, 1920, 8, seed=1)
        frame = np.clip(frame.astype(np.int16) + rng.integers(-6, 7, frame.shape), 0, 255).astype(np.uint8)
    else:                            # photo-like statistics: smooth large-scale structure + faint sensor noise (~0.1 byte per pixel at q90)
        low = rng.integers(0, 256, (68, 120, 3), dtype=np.uint8)
        frame = np.asarray(Image.fromarray(low).resize((1920, 1080), Image.BICUBIC)).astype(np.int16)
        frame = np.clip(frame + rng.integers(-2, 3, frame.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    kw = dict(restart_marker_rows=args.restart_rows) if args.restart_rows else {}
    Image.fromarray(frame[..., ::-1]).save(buf, format="JPEG", quality=args.quality, subsampling=2, **kw)
    data = buf.getvalue()
    eng = _native.Engine(0)
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1]
    _, _, _, got = eng.decode_jpeg(data)
    assert np.array_equal(got, ref), "decoder differs from libjpeg"
    for _ in range(3):
        eng.decode_jpeg(data, want_host=False)
    t0 = time.perf_counter()
    for _ in range(args.n):
        eng.decode_jpeg(data, want_host=False)
    t_dev = (time.perf_counter() - t0) / args.n
    t0 = time.perf_counter()
    for _ in range(args.n):
        a = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    t_pil = (time.perf_counter() - t0) / args.n
    bgr = np.ascontiguousarray(a[..., ::-1])
    t0 = time.perf_counter()
    for _ in range(args.n):
        eng.set_frame(bgr)
    t_up = (time.perf_counter() - t0) / args.n
    # batch ingest: 32 files per call, Huffman decoding on T host threads, device stages once over the batch
    batch = {}
    files = [data] * 32
    files = [data] * 96
    for T in (1, 16, 32, 48, 96):
        eng.decode_jpeg_batch(files, threads=T)
        t0 = time.perf_counter()
        reps = max(2, args.n // 10)
        for _ in range(reps):
            eng.decode_jpeg_batch(files, threads=T)
        dt = (time.perf_counter() - t0) / reps
        batch[str(T)] = {"ms_per_96_frames": round(dt * 1e3, 2), "frames_per_s": round(96 / dt, 1)}
    print(json.dumps({"jpeg_bytes": len(data), "quality": args.quality, "frame": "1920x1080 4:2:0", "content": args.content, "restart_marker_rows": args.restart_rows,
                      "entropy_decoding": "device (one thread per restart interval) for the batches, host for the single file" if args.restart_rows else "host threads", "batch_by_host_threads": batch,
                      "host_cores": os.cpu_count(),
                      "pf_decode_jpeg_ms": round(t_dev * 1e3, 3), "frames_per_s_one_stream": round(1.0 / t_dev, 1),
                      "libjpeg_host_decode_ms": round(t_pil * 1e3, 3), "host_frame_upload_and_gate_ms": round(t_up * 1e3, 3),
                      "bit_identical_with_libjpeg": True}))
    eng.close()


if __name__ == "__main__":
    main()
