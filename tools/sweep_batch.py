"""Throughput of the landmark program vs batch size (device-resident inputs).  Dev tool."""
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench_support as bs  # noqa: E402
from peppa_pig_face_landmark_amd._native import Engine  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    blobs = bs.build_programs("landmark", dtype)
    res = {}
    for B in (8, 16, 32, 64, 128, 256, 512):
        bs.load_programs(eng, blobs, "landmark", B, 1)
        wl = bs.LandmarkWorkload(eng, dev, B, seed=1)
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        n = max(4, 2048 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            wl.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[B] = B * n / dt
        print("B=%4d  %9.1f faces/s  %8.3f ms/step" % (B, res[B], dt / n * 1e3), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "sweep_%s.json" % dtype), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
