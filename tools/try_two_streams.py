"""Experiment: does running two halves of a step on two engines (two HIP streams) concurrently help?"""
import os, sys, threading, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench_support as bs
from peppa_pig_face_landmark_amd._native import Engine

def run(nlanes, frames_total, steps=20):
    dev = torch.device("cuda", 0)
    blobs = bs.build_programs("pipeline", "f32s")
    lanes = []
    per = frames_total // nlanes
    for i in range(nlanes):
        eng = Engine(0)
        bs.load_programs(eng, blobs, "pipeline", per * 8, per)
        lanes.append(bs.PipelineWorkload(eng, dev, per, 8, seed=7 + i))
    def work(wl, n):
        for _ in range(n):
            wl.step()
        wl.eng.sync()
    for wl in lanes: work(wl, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(wl, steps)) for wl in lanes]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("lanes=%d frames/step=%d  %.1f faces/s  %.3f ms/step" % (nlanes, frames_total, frames_total * 8 * steps / dt, dt / steps * 1e3), flush=True)
    for wl in lanes: wl.eng.close()

for nl, fr in ((1, 32), (2, 32), (1, 64), (2, 64), (4, 64), (1, 128), (2, 128)):
    run(nl, fr)
