"""Benchmark plumbing shared by bench.py and tools/: program packing, RCCL weight broadcast, device-resident
synthetic workloads.  Lives beside bench.py, OUTSIDE the product package: it imports torch (device memory and
torch.distributed -- plumbing), which the engine and its ctypes shim never do; every kernel on the timed path is
the HIP engine's."""
from __future__ import annotations

import os
import time
from typing import Dict, Optional, Tuple

import numpy as np

from peppa_pig_face_landmark_amd import _native
from peppa_pig_face_landmark_amd.graph.random_init import random_student_weights
from peppa_pig_face_landmark_amd.graph.student import build_student_program

try:  # the detector program builder arrives with the full pipeline
    from peppa_pig_face_landmark_amd.graph.detector import build_detector_program, random_detector_weights
except ImportError:  # pragma: no cover
    build_detector_program = None
    random_detector_weights = None


def pipeline_available() -> bool:
    return build_detector_program is not None


def build_programs(workload: str, dtype: str, model: str = "student", **student_kw) -> Dict[int, bytes]:
    if model == "teacher":
        from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program, random_teacher_weights
        blobs = {_native.PF_NET_LANDMARK: build_teacher_program(random_teacher_weights(2), 256, dtype, **{k: v for k, v in student_kw.items() if k == "one_product"})[0]}
    else:
        blobs = {_native.PF_NET_LANDMARK: build_student_program(random_student_weights(0), 256, dtype, **student_kw)[0]}
    if workload == "pipeline":
        blobs[_native.PF_NET_DETECTOR] = build_detector_program(random_detector_weights(1), (384, 640), dtype)[0]
    return blobs


def broadcast_blobs(blobs: Optional[Dict[int, bytes]], dev, rank: int) -> Tuple[Dict[int, bytes], float]:
    """One-time weight distribution: rank 0's packed programs -> every rank, as uint8 HBM tensors
    over RCCL (torch.distributed 'nccl' backend == RCCL on ROCm).  Returns (blobs, milliseconds)."""
    import torch
    import torch.distributed as dist

    meta = torch.zeros(8, dtype=torch.int64, device=dev)
    if rank == 0:
        slots = sorted(blobs)
        meta[0] = len(slots)
        for i, s in enumerate(slots):
            meta[1 + 2 * i] = s
            meta[2 + 2 * i] = len(blobs[s])
    dist.broadcast(meta, 0)
    n = int(meta[0].item())
    out: Dict[int, bytes] = {}
    on_gpu = torch.device(dev).type == "cuda"
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    tensors = []
    for i in range(n):
        slot, size = int(meta[1 + 2 * i].item()), int(meta[2 + 2 * i].item())
        if rank == 0:
            t = torch.frombuffer(bytearray(blobs[slot]), dtype=torch.uint8).to(dev)
        else:
            t = torch.empty(size, dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0)
        tensors.append((slot, t))
    if on_gpu:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    for slot, t in tensors:
        out[slot] = t.cpu().numpy().tobytes()
    return out, ms


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(script: str, argv, n: int, extra_env=None) -> int:
    """`python script --gpus N ...` without a launcher: re-exec the same command line under
    torch.distributed.run with one rank per GPU on this node (the form the driver uses for N > 1) and pass
    its exit code through.  Rank 0's single JSON line reaches stdout unchanged."""
    import os
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    if extra_env:
        env.update(extra_env)
    return subprocess.run(cmd, env=env).returncode


def broadcast_programs_rccl(eng: "_native.Engine", blobs: Optional[Dict[int, bytes]], slots, rank: int, world: int,
                            exchange_id) -> Tuple[Dict[int, bytes], float, int]:
    """One-time weight distribution through the engine's own C ABI (pf_broadcast_weights -> ncclBroadcast on the
    engine's stream, HBM to HBM over xGMI).  `exchange_id(id_or_None) -> id` moves rank 0's 128-byte RCCL unique
    id to every rank out of band.  Returns (blobs, total device ms of the payload broadcasts, total bytes)."""
    uid = exchange_id(_native.Engine.comm_unique_id() if rank == 0 else None)
    out: Dict[int, bytes] = {}
    ms_total, nbytes = 0.0, 0
    for slot in slots:
        blob, ms = eng.broadcast_weights(uid, rank, world, slot, blobs[slot] if rank == 0 else None, max_batch=1)
        out[slot] = blob
        ms_total += ms
        nbytes += len(blob)
    return out, ms_total, nbytes


def shard_frames(n_frames: int, rank: int, world: int):
    """Frame f is processed by rank f mod world: independent units, no data-path collective."""
    return list(range(rank, n_frames, world))


def load_programs(eng: "_native.Engine", blobs: Dict[int, bytes], workload: str, faces: int, frames: int):
    import os
    if os.environ.get("PEPPA_BENCH_NO_GUARD"):       # measurement aid only: what the always-on f32s range guard costs
        eng.set_option(_native.PF_OPT_RANGE_CHECK, 0)
    eng.load_program(_native.PF_NET_LANDMARK, blobs[_native.PF_NET_LANDMARK], faces)
    if workload == "pipeline":
        eng.load_program(_native.PF_NET_DETECTOR, blobs[_native.PF_NET_DETECTOR], frames)


def synthetic_crops(n: int, size: int, seed: int) -> np.ndarray:
    """uint8 [n,size,size,3]: smooth blobs + noise (SURVEY 8d set B), 8 distinct images tiled."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    base = []
    for _ in range(min(n, 8)):
        img = np.full((size, size, 3), 40.0, np.float32)
        for c in range(3):
            for _ in range(6):
                cx, cy = rng.uniform(0, size, 2)
                sig = rng.uniform(8.0, 40.0) * size / 256.0
                img[:, :, c] += rng.uniform(40.0, 200.0) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig * sig))
        img += rng.normal(0.0, 6.0, img.shape)
        base.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    reps = (n + len(base) - 1) // len(base)
    return np.stack((base * reps)[:n])


class LandmarkWorkload:
    """BASELINE configs[1]: B pre-cropped 256x256 faces, device resident, outputs stay on device."""

    def __init__(self, eng, dev, batch: int, seed: int):
        import torch
        self.eng, self.batch = eng, batch
        self.crops = torch.from_numpy(synthetic_crops(batch, 256, seed)).to(dev)
        self.loc = torch.empty((batch, 196), dtype=torch.float32, device=dev)
        self.score = torch.empty((batch, 98), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()

    def step(self):
        self.eng.landmark_forward_device(self.crops.data_ptr(), _native.PF_INPUT_U8_NHWC, self.batch,
                                         self.loc.data_ptr(), self.score.data_ptr())

    def sync(self):
        self.eng.sync()

    def check(self):
        import torch
        self.eng.sync()
        assert bool(torch.isfinite(self.loc).all()) and bool(torch.isfinite(self.score).all()), "non-finite landmarks"

    def profile(self, steps: int):
        self.eng.sync()
        self.eng.profile_enable(True)
        for _ in range(steps):
            self.step()
        self.eng.sync()
        prof = self.eng.profile_fetch()
        self.eng.profile_enable(False)
        return prof


class PipelineWorkload:
    """BASELINE configs[2]: F 1080p frames x 8 faces per step, everything device resident.

    Because the trained detector weights are unavailable, detection *semantics* are pinned by planted
    candidates: the letterbox + detector network + decode run for real on random-init
    weights (their time is inside the step), then NMS consumes a decoded-row tensor in which every
    face has 24 jittered candidates -- NMS must reduce them to exactly the 8 boxes -- followed by
    top-k, crop/resize, Student@256, heat-map decode and back-projection."""

    H, W, ROWS = 1080, 1920, 15120

    def __init__(self, eng, dev, frames: int, faces_per_frame: int, seed: int, graph: bool = True,
                 frame_hw: Tuple[int, int] = (1080, 1920)):
        import torch
        from peppa_pig_face_landmark_amd.synth import make_frame, make_frame_grid, plant_rows
        self.eng, self.F, self.K = eng, frames, faces_per_frame
        self.graph = bool(graph)
        self.H, self.W = frame_hw                            # (2160, 3840) x 32 faces = BASELINE config 5 / SURVEY C5
        eng.set_option(_native.PF_OPT_HIP_GRAPH, 1 if graph else 0)   # replay the step from a captured hipGraph
        self.frames, self.rows = self.synth_inputs(dev, frames, faces_per_frame, seed, frame_hw)
        self.unique_frames = frames
        n = frames * faces_per_frame
        self.counts = torch.zeros((frames,), dtype=torch.int32, device=dev)
        self.boxes = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        self.kps = torch.zeros((n, 98, 2), dtype=torch.float32, device=dev)
        self.scores = torch.zeros((n, 98), dtype=torch.float32, device=dev)
        # results leave the device inside the step, like FaceAna.run() returning numpy (facer.py:84-96): page-locked
        # host buffers, device->host copies enqueued on the engine's stream (part of the captured graph)
        self.h_counts = eng.pinned_empty((frames,), np.int32)
        self.h_boxes = eng.pinned_empty((n, 4), np.float32)
        self.h_kps = eng.pinned_empty((n, 98, 2), np.float32)
        self.h_scores = eng.pinned_empty((n, 98), np.float32)
        for a in (self.h_counts, self.h_boxes, self.h_kps, self.h_scores):
            a[...] = 0
        torch.cuda.synchronize()

    @staticmethod
    def synth_inputs(dev, frames: int, faces_per_frame: int, seed: int, frame_hw):
        """(frames uint8 [F,H,W,3], planted rows f32 [F,15120,16]) on the device: two base scenes tiled over the slots, every
        slot with its own +-3 of pixel noise so a step reads F distinct frames from HBM."""
        import torch
        from peppa_pig_face_landmark_amd.synth import make_frame, make_frame_grid, plant_rows
        H, W = frame_hw
        base_frames, base_rows = [], []
        for i in range(min(frames, 2)):
            if faces_per_frame == 32:
                fr, boxes = make_frame_grid(H, W, 8, 4, seed=seed + i)
            else:
                fr, boxes = make_frame(H, W, faces_per_frame, seed=seed + i)
            base_frames.append(fr)
            base_rows.append(plant_rows(boxes, (H, W), PipelineWorkload.ROWS, (384, 640), 24, seed=seed + i))
        reps = (frames + len(base_frames) - 1) // len(base_frames)
        frames_t = torch.from_numpy(np.stack((base_frames * reps)[:frames])).to(dev)
        # every frame slot gets its own pixels (+-3 of per-frame noise on top of the two base scenes), so the step
        # reads F distinct frames from HBM instead of two cache-resident ones (96 x 6.2 MB = 597 MB > the 256 MB L3)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1000 + seed)
        for i in range(frames):
            nz = torch.randint(-3, 4, frames_t[i].shape, dtype=torch.int16, device=dev, generator=gen)
            frames_t[i] = (frames_t[i].to(torch.int16) + nz).clamp_(0, 255).to(torch.uint8)
        rows_t = torch.from_numpy(np.stack((base_rows * reps)[:frames])).to(dev)
        return frames_t, rows_t

    def step(self):
        self.eng.run_frames_device(self.frames.data_ptr(), self.F, self.H, self.W, 0.5, 0.3, 1600.0, self.K,
                                   d_planted=self.rows.data_ptr(), rows=self.ROWS, d_counts=self.h_counts.ctypes.data,
                                   d_boxes=self.h_boxes.ctypes.data, d_kps=self.h_kps.ctypes.data,
                                   d_scores=self.h_scores.ctypes.data, out_mem=_native.PF_MEM_HOST_PINNED)

    def enable_host_frames(self):
        """Frame ingest seam (next-row N2): the same frames in page-locked HOST memory (pf_host_alloc)."""
        self.host_frames = self.eng.pinned_empty((self.F, self.H, self.W, 3), np.uint8)
        self.host_frames[...] = self.frames.cpu().numpy()

    def step_host(self):
        """One step with the frames crossing PCIe inside the call (host -> device copy on this lane's stream)."""
        self.eng.run_frames_host_async(self.host_frames, self.rows.data_ptr(), self.ROWS, 0.5, 0.3, 1600.0, self.K,
                                       self.counts.data_ptr(), self.boxes.data_ptr(), self.kps.data_ptr(),
                                       self.scores.data_ptr())

    def enable_jpeg_frames(self, quality: int = 90, restart_rows: int = 0):
        """Frame ingest seam (next-row N2), file side: this lane's frames as baseline 4:2:0 JPEG files (PIL / libjpeg encodes
        them once, outside any timed region), optionally with a restart marker every `restart_rows` MCU rows (such files have
        their Huffman stream decoded on the device).  Returns the total size of the files."""
        import io
        from PIL import Image
        fr = self.frames.cpu().numpy()
        self.jpegs = []
        for i in range(self.F):
            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(fr[i][..., ::-1])).save(buf, format="JPEG", quality=quality, subsampling=2,
                                                                        **(dict(restart_marker_rows=restart_rows) if restart_rows else {}))
            self.jpegs.append(buf.getvalue())
        return sum(len(j) for j in self.jpegs)

    def step_jpeg(self, threads: int):
        """One step from JPEG FILES: pf_decode_jpeg_batch (Huffman on `threads` host threads, the rest on this lane's stream)
        straight into pf_run_frames on the decoded device frames."""
        d, n, h, w = self.eng.decode_jpeg_batch(self.jpegs, threads)
        assert (n, h, w) == (self.F, self.H, self.W)
        self.eng.run_frames_device(d, self.F, h, w, 0.5, 0.3, 1600.0, self.K, d_planted=self.rows.data_ptr(), rows=self.ROWS,
                                   d_counts=self.h_counts.ctypes.data, d_boxes=self.h_boxes.ctypes.data,
                                   d_kps=self.h_kps.ctypes.data, d_scores=self.h_scores.ctypes.data,
                                   out_mem=_native.PF_MEM_HOST_PINNED)

    def latency_p50(self, frames: int, reps: int = 40):
        """p50 / p99 wall time (ms) of one synchronous call on `frames` frames (submit -> results complete)."""
        import time
        frames = min(frames, self.F)
        ts = []
        for _ in range(reps + 5):
            t0 = time.perf_counter()
            self.eng.run_frames_device(self.frames.data_ptr(), frames, self.H, self.W, 0.5, 0.3, 1600.0, self.K,
                                       d_planted=self.rows.data_ptr(), rows=self.ROWS, d_counts=self.counts.data_ptr(),
                                       d_boxes=self.boxes.data_ptr(), d_kps=self.kps.data_ptr(),
                                       d_scores=self.scores.data_ptr())
            self.eng.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[5:])
        return ts[len(ts) // 2], ts[min(len(ts) - 1, int(len(ts) * 0.99))]

    def check(self, compare_eager: bool = True):
        """The timed steps deliver to the page-locked host buffers (the latency / PCIe probes to the device ones).
        compare_eager=False: after steps fed from JPEG files (lossy: their landmarks are not those of the resident frames)."""
        self.eng.sync()
        if os.environ.get("PEPPA_DBG"):      # timing ablations of the -DPF_ABLATE=1 library compute garbage
            return
        assert bool((self.h_counts == self.K).all()), "NMS did not return the planted faces: %s" % self.h_counts.tolist()
        assert bool(np.isfinite(self.h_kps).all()) and bool(np.isfinite(self.h_scores).all()), "non-finite landmarks"
        assert float(np.abs(self.h_kps).max()) > 0.0, "results never reached the host buffers"
        # the timed steps replay a captured hipGraph: the SAME inputs launched eagerly (graph off, results into the device-side
        # buffers) must give bit-identical counts / boxes / landmarks / scores -- a stale or mis-captured graph cannot pass
        if self.graph and compare_eager:
            self.eng.set_option(_native.PF_OPT_HIP_GRAPH, 0)
            try:
                self.eng.run_frames_device(self.frames.data_ptr(), self.F, self.H, self.W, 0.5, 0.3, 1600.0, self.K,
                                           d_planted=self.rows.data_ptr(), rows=self.ROWS, d_counts=self.counts.data_ptr(),
                                           d_boxes=self.boxes.data_ptr(), d_kps=self.kps.data_ptr(), d_scores=self.scores.data_ptr())
                self.eng.sync()
            finally:
                self.eng.set_option(_native.PF_OPT_HIP_GRAPH, 1)
            for name, host, dev_t in (("counts", self.h_counts, self.counts), ("boxes", self.h_boxes, self.boxes),
                                      ("landmarks", self.h_kps, self.kps), ("scores", self.h_scores, self.scores)):
                assert np.array_equal(host, dev_t.cpu().numpy()), "graph replay and eager launch disagree on " + name

    profile = LandmarkWorkload.profile
    sync = LandmarkWorkload.sync


class BatchPipelineWorkload:
    """BASELINE configs[2] on the product's multi-lane runner (``_native.BatchEngine`` = ``pf_batch_*`` of the C ABI,
    the engine behind ``FrameBatchRunner``): ONE ``pf_batch_run_frames`` call per step; the library splits the frames
    into contiguous per-lane slices and runs them on its own HIP streams.  This file only synthesises the inputs, owns
    the result buffers and compares -- the orchestration that produces the headline is the package's.
    (34.9 k / 39.2 k / 40.4 k / 39.1 k faces/s at 1 / 2 / 3 / 4 lanes of 32 frames in round 1.)"""

    ROWS = 15120

    def __init__(self, batch, dev, frames: int, faces_per_frame: int, seed: int, lanes: int,
                 graph: bool = True, frame_hw: Tuple[int, int] = (1080, 1920), front: bool = True):
        import torch
        assert frames % lanes == 0
        self.batch, self.F, self.K, self.L = batch, frames, faces_per_frame, lanes
        self.front = bool(front)
        self.front_prof = {}
        batch.set_option(_native.PF_OPT_BATCH_FRONT, 1 if front else 0)
        self.per = frames // lanes
        self.H, self.W = frame_hw
        self.graph = bool(graph)
        batch.set_option(_native.PF_OPT_HIP_GRAPH, 1 if graph else 0)
        # the same per-lane scenes as the per-engine workload used to build (seed + 101 * lane), concatenated in lane order
        parts = [PipelineWorkload.synth_inputs(dev, self.per, faces_per_frame, seed + 101 * i, frame_hw) for i in range(lanes)]
        self.frames = torch.cat([p[0] for p in parts])
        self.rows = torch.cat([p[1] for p in parts])
        del parts
        self.unique_frames = frames
        n = frames * faces_per_frame
        self.counts = torch.zeros((frames,), dtype=torch.int32, device=dev)
        self.boxes = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        self.kps = torch.zeros((n, 98, 2), dtype=torch.float32, device=dev)
        self.scores = torch.zeros((n, 98), dtype=torch.float32, device=dev)
        self.h_counts = batch.pinned_empty((frames,), np.int32)
        self.h_boxes = batch.pinned_empty((n, 4), np.float32)
        self.h_kps = batch.pinned_empty((n, 98, 2), np.float32)
        self.h_scores = batch.pinned_empty((n, 98), np.float32)
        for a in (self.h_counts, self.h_boxes, self.h_kps, self.h_scores):
            a[...] = 0
        torch.cuda.synchronize()

    def _run(self, to_host: bool):
        c, b, k, s = ((self.h_counts.ctypes.data, self.h_boxes.ctypes.data, self.h_kps.ctypes.data, self.h_scores.ctypes.data) if to_host
                      else (self.counts.data_ptr(), self.boxes.data_ptr(), self.kps.data_ptr(), self.scores.data_ptr()))
        self.batch.run_frames_device(self.frames.data_ptr(), self.F, self.H, self.W, 0.5, 0.3, 1600.0, self.K,
                                     d_planted=self.rows.data_ptr(), rows=self.ROWS, d_counts=c, d_boxes=b, d_kps=k, d_scores=s,
                                     out_mem=_native.PF_MEM_HOST_PINNED if to_host else _native.PF_MEM_DEVICE)

    def step(self):
        self._run(True)

    def sync(self):
        self.batch.sync()

    def check(self, compare_eager: bool = True):
        self.batch.sync()
        if os.environ.get("PEPPA_DBG"):
            return
        assert bool((self.h_counts == self.K).all()), "NMS did not return the planted faces: %s" % self.h_counts.tolist()
        assert bool(np.isfinite(self.h_kps).all()) and bool(np.isfinite(self.h_scores).all()), "non-finite landmarks"
        assert float(np.abs(self.h_kps).max()) > 0.0, "results never reached the host buffers"
        if self.graph and compare_eager:      # graph replay == the same inputs launched eagerly, bit for bit
            self.batch.set_option(_native.PF_OPT_HIP_GRAPH, 0)
            try:
                self._run(False)
                self.batch.sync()
            finally:
                self.batch.set_option(_native.PF_OPT_HIP_GRAPH, 1)
            for name, host, dev_t in (("counts", self.h_counts, self.counts), ("boxes", self.h_boxes, self.boxes),
                                      ("landmarks", self.h_kps, self.kps), ("scores", self.h_scores, self.scores)):
                assert np.array_equal(host, dev_t.cpu().numpy()), "graph replay and eager launch disagree on " + name

    # ---- probes (never the headline) ------------------------------------------------------------------------------------------
    def _lane_args(self, i: int):
        f0 = i * self.per
        return dict(d_planted=self.rows[f0:].data_ptr(), rows=self.ROWS, d_counts=self.counts[f0:].data_ptr(),
                    d_boxes=self.boxes[f0 * self.K:].data_ptr(), d_kps=self.kps[f0 * self.K:].data_ptr(),
                    d_scores=self.scores[f0 * self.K:].data_ptr())

    def profile(self, steps: int):
        """Per-kernel HIP-event times.  Front mode (PF_OPT_BATCH_FRONT, the default): `steps` whole batch calls with profiling on
        the front engine (letterbox + detector + NMS of ALL frames: ``self.front_prof``) and on lane 0 (crop + landmarks of its
        slice: the return value); profiling serialises a handle's launches, the other lanes run alongside.  Per-lane mode: lane 0
        running ALONE on its slice, everything in the return value and ``front_prof`` empty."""
        self.batch.sync()
        eng = self.batch.lane(0)
        self.front_prof = {}
        if self.front:
            fr = self.batch.front()
            fr.profile_enable(True)
            eng.profile_enable(True)
            for _ in range(steps):
                self._run(False)
                self.batch.sync()
            self.front_prof = fr.profile_fetch()
            prof = eng.profile_fetch()
            fr.profile_enable(False)
            eng.profile_enable(False)
            return prof
        eng.profile_enable(True)
        for _ in range(steps):
            eng.run_frames_device(self.frames.data_ptr(), self.per, self.H, self.W, 0.5, 0.3, 1600.0, self.K, **self._lane_args(0))
        eng.sync()
        prof = eng.profile_fetch()
        eng.profile_enable(False)
        return prof

    def one_lane_rate(self, steps: int):
        """faces/s of ONE lane (one engine, one stream: what a plain pf_run_frames user gets) on its slice, graph replay."""
        import time
        self.batch.sync()
        eng = self.batch.lane(0)
        for _ in range(2):
            eng.run_frames_device(self.frames.data_ptr(), self.per, self.H, self.W, 0.5, 0.3, 1600.0, self.K, **self._lane_args(0))
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_frames_device(self.frames.data_ptr(), self.per, self.H, self.W, 0.5, 0.3, 1600.0, self.K, **self._lane_args(0))
        eng.sync()
        return self.per * self.K * steps / (time.perf_counter() - t0)

    def latency_p50(self, frames: int, reps: int = 40):
        import time
        self.batch.sync()
        eng = self.batch.lane(0)
        frames = min(frames, self.per)
        ts = []
        for _ in range(reps + 5):
            t0 = time.perf_counter()
            eng.run_frames_device(self.frames.data_ptr(), frames, self.H, self.W, 0.5, 0.3, 1600.0, self.K, **self._lane_args(0))
            eng.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[5:])
        return ts[len(ts) // 2], ts[min(len(ts) - 1, int(len(ts) * 0.99))]

    def enable_host_frames(self):
        self.host_frames = self.batch.pinned_empty((self.F, self.H, self.W, 3), np.uint8)
        self.host_frames[...] = self.frames.cpu().numpy()

    def step_host(self):
        self.batch.run_frames_host_async(self.host_frames, self.rows.data_ptr(), self.ROWS, 0.5, 0.3, 1600.0, self.K,
                                         self.counts.data_ptr(), self.boxes.data_ptr(), self.kps.data_ptr(), self.scores.data_ptr())

    def enable_jpeg_frames(self, quality: int = 90, restart_rows: int = 0):
        import io
        from PIL import Image
        fr = self.frames.cpu().numpy()
        self.jpegs = []
        for i in range(self.F):
            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(fr[i][..., ::-1])).save(buf, format="JPEG", quality=quality, subsampling=2,
                                                                        **(dict(restart_marker_rows=restart_rows) if restart_rows else {}))
            self.jpegs.append(buf.getvalue())
        return sum(len(j) for j in self.jpegs)

    def step_jpeg(self, threads: int):
        """Every lane decodes its files (pf_decode_jpeg_batch) and runs them from its own host thread (the decode call blocks on
        the host-side stage; ctypes releases the GIL), so one lane's host work overlaps the others' kernels."""
        from concurrent.futures import ThreadPoolExecutor
        if not hasattr(self, "_pool"):
            self._pool = ThreadPoolExecutor(self.L)

        def one(i):
            eng = self.batch.lane(i)
            f0 = i * self.per
            d, n, h, w = eng.decode_jpeg_batch(self.jpegs[f0:f0 + self.per], threads)
            assert (n, h, w) == (self.per, self.H, self.W)
            eng.run_frames_device(d, self.per, h, w, 0.5, 0.3, 1600.0, self.K, d_planted=self.rows[f0:].data_ptr(), rows=self.ROWS,
                                  d_counts=self.h_counts[f0:].ctypes.data, d_boxes=self.h_boxes[f0 * self.K:].ctypes.data,
                                  d_kps=self.h_kps[f0 * self.K:].ctypes.data, d_scores=self.h_scores[f0 * self.K:].ctypes.data,
                                  out_mem=_native.PF_MEM_HOST_PINNED)
        list(self._pool.map(one, range(self.L)))

    def close(self):
        self.batch.close()
