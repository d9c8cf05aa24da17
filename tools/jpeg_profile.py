"""Per-stage HIP-event times of pf_decode_jpeg_batch on N copies of one 1080p 4:2:0 file (run on the GPU box).
usage: python tools/jpeg_profile.py [--n 32] [--quality 90] [--content noise|photo]"""
import argparse, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402
from peppa_pig_face_landmark_amd import _native  # noqa: E402
from peppa_pig_face_landmark_amd.synth import make_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--quality", type=int, default=90)
ap.add_argument("--content", default="noise")
ap.add_argument("--reps", type=int, default=6)
args = ap.parse_args()
rng = np.random.default_rng(0)
if args.content == "noise":
    frame, _ = make_frame(1080, 1920, 8, seed=1)
    frame = np.clip(frame.astype(np.int16) + rng.integers(-6, 7, frame.shape), 0, 255).astype(np.uint8)
else:
    low = rng.integers(0, 256, (68, 120, 3), dtype=np.uint8)
    frame = np.asarray(Image.fromarray(low).resize((1920, 1080), Image.BICUBIC)).astype(np.int16)
    frame = np.clip(frame + rng.integers(-2, 3, frame.shape), 0, 255).astype(np.uint8)
buf = io.BytesIO()
Image.fromarray(frame[..., ::-1]).save(buf, format="JPEG", quality=args.quality, subsampling=2)
data = buf.getvalue()
eng = _native.Engine(0)
files = [data] * args.n
for _ in range(2):
    eng.decode_jpeg_batch(files, threads=4)
eng.sync()
t0 = time.perf_counter()
for _ in range(args.reps):
    eng.decode_jpeg_batch(files, threads=4)
eng.sync()
wall = (time.perf_counter() - t0) / args.reps
eng.profile_enable(True)
eng.profile_fetch()
for _ in range(args.reps):
    eng.decode_jpeg_batch(files, threads=4)
eng.sync()
prof = eng.profile_fetch()
print(json.dumps({"files": args.n, "jpeg_bytes": len(data), "content": args.content, "quality": args.quality,
                  "wall_ms_per_batch": round(wall * 1e3, 3),
                  "stage_ms_per_batch": {k: round(v[0] / args.reps, 4) for k, v in prof.items() if k.startswith("jpeg")}}))
eng.close()
