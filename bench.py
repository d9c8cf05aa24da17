#!/usr/bin/env python
"""Benchmark of the FaceAna hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM when the timed region starts:
  --workload pipeline : BASELINE.json configs[2] -- F 1080p frames x 8 planted faces each through
                        letterbox -> detector net -> decode -> NMS -> top-k -> crop/resize ->
                        Student@256 -> heat-map decode -> back-projection   (default when built)
  --workload landmark : BASELINE.json configs[1] -- 256 pre-cropped 256x256 faces through Student@256
Frames / faces shard across ranks with no data-path collective (weak scaling: per-rank work is
fixed); the only collective is the one-time RCCL broadcast of the packed weights from rank 0.
Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_FACE = {"student": 2.968, "teacher": 11.9}   # 1 484.1 M MAC; Teacher from README 5.53 GiMAC
GFLOP_DETECTOR = 0.34        # yolov5n-0.5 @384x640 per frame (upstream figure)
# dense MFMA peaks (MI355X_MICROARCH.md).  "f32s" = f32 tensors, split-precision convs: every product
# is 3 v_mfma_f32_16x16x32_f16 instructions (hi*hi + hi*lo + lo*hi), so it is priced against the f16 pipe.
PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "f32s": 2500.0}
MFMA_INSTR_PER_PRODUCT = {"f32": 1, "f16": 1, "f32s": 3}
HERO_TAG = "conv3x3_c128_n128_64x64"          # up2.conv2 (model.py:165-172): 40.7 % of all MACs
HERO_FLOP_PER_FACE = 2.0 * 64 * 64 * 128 * 128 * 9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "landmark", "pipeline"])
    ap.add_argument("--batch", type=int, default=256, help="faces per step per GPU (landmark workload)")
    ap.add_argument("--frames", type=int, default=96, help="1080p frames per step per GPU (pipeline workload)")
    ap.add_argument("--lanes", type=int, default=3, help="concurrent HIP streams (engines) per GPU sharing a step's frames")
    ap.add_argument("--faces-per-frame", type=int, default=8)
    ap.add_argument("--frame-hw", type=int, nargs=2, default=[1080, 1920], metavar=("H", "W"),
                    help="frame size; --frame-hw 2160 3840 --faces-per-frame 32 --model teacher = BASELINE config 5")
    ap.add_argument("--dtype", default="f32s", choices=["f32", "f32s", "f16"],
                    help="f32: exact v_mfma_f32 convs; f32s (default): f32 tensors + split-precision 3xf16 MFMA convs "
                         "(same accuracy); f16: f16 storage fast mode (parity not claimed)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--model", default="student", choices=["student", "teacher"],
                    help="landmark regressor: Student (headline) or Teacher/HRNet-W18 (BASELINE config 5 model)")
    ap.add_argument("--no-probes", action="store_true", help="skip the call-latency and PCIe-inclusive probes (keeps a rocprofv3 "
                    "--stats run of this command to launches of ONE batch size, so its per-kernel averages are comparable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-faces", type=int, default=48)
    ap.add_argument("--dump-profile", default="", help="write the full per-kernel HIP-event table (JSON) here")
    return ap.parse_args()


def cpu_baseline(workload: str, n_faces: int):
    """Reference-shaped CPU path timed on this box's host cores: the torch-CPU oracle (stand-in for
    onnxruntime-CPU, which is not installed) run exactly like face_landmark.py:40-48 -- one face at
    a time, batch 1, float32 -- on a bounded sample.  Baseline, not target."""
    import torch
    from oracle import landmark_net as ln
    from oracle import synth_weights as sw

    W = ln.to_torch(sw.student_weights())
    crops = sw.smooth_blob_images(8, 256, seed=99)
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        ln.student_forward(W, x[:1])
        t0 = time.perf_counter()
        for i in range(n_faces):
            ln.student_forward(W, x[i % 8:i % 8 + 1])
        dt = time.perf_counter() - t0
    return {"value": round(n_faces / dt, 2), "unit": "faces/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d faces, Student@256 landmark forward only, batch=1 python loop (face_landmark.py:40-48 shape), "
                      "torch-CPU f32 oracle as stand-in for onnxruntime-CPU; %.1f s" % (n_faces, dt)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ   # launched by torch.distributed.run
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout at communicator creation; the contract is ONE JSON line on
        # stdout, so stdout is pointed at stderr while the communicator comes up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from peppa_pig_face_landmark_amd import build as pbuild
    from peppa_pig_face_landmark_amd._native import Engine, PF_INPUT_U8_NHWC
    from peppa_pig_face_landmark_amd import bench_support as bs

    if rank == 0:
        pbuild.build_hip()
    if use_dist:
        dist.barrier()
    eng = Engine(local_rank)

    workload = args.workload
    if workload == "auto":
        workload = "pipeline" if bs.pipeline_available() else "landmark"

    # ---- weights: packed on rank 0, broadcast once over RCCL / xGMI -----------------------------
    t0 = time.time()
    blobs = bs.build_programs(workload, args.dtype, args.model) if rank == 0 else None
    bcast_ms = 0.0
    if use_dist:
        blobs, bcast_ms = bs.broadcast_blobs(blobs, dev, rank)
    faces_per_step = args.batch if workload == "landmark" else args.frames * args.faces_per_frame
    lanes = args.lanes if workload == "pipeline" else 1
    if lanes == 1:
        bs.load_programs(eng, blobs, workload, faces_per_step, args.frames)
    setup_s = time.time() - t0

    # ---- synthetic inputs, resident in HBM ----------------------------------------------------------
    if workload == "landmark":
        state = bs.LandmarkWorkload(eng, dev, args.batch, seed=1234 + rank)
    elif lanes == 1:
        state = bs.PipelineWorkload(eng, dev, args.frames, args.faces_per_frame, seed=7 + rank, graph=not args.no_graph,
                                    frame_hw=tuple(args.frame_hw))
    else:
        eng.close()
        state = bs.MultiLanePipeline(lambda: Engine(local_rank), blobs, dev, args.frames, args.faces_per_frame,
                                     seed=7 + rank, lanes=lanes, graph=not args.no_graph, frame_hw=tuple(args.frame_hw))
        eng = state.lanes[0].eng

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        state.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    barrier()
    state.check()

    # ---- per-kernel device time (HIP events on the engine's own stream), dominant kernel roofline ---
    prof = state.profile(3)
    hero_ms, hero_n = prof.get(HERO_TAG, (0.0, 0))
    faces_per_launch = faces_per_step // lanes      # the profiled lane processes 1/lanes of the step
    roofline = None
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_hero_kernel.json")
    if args.dtype in ("f32", "f32s") and os.path.exists(pmc_path):
        # HBM bytes of the hero launch from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in
        # separate runs, gfx950 FETCH correction applied), scaled to this run's faces per launch
        with open(pmc_path) as f:
            pmc = json.load(f)
        traffic = int(pmc["traffic_bytes_per_launch"] * (faces_per_step // (args.lanes if workload == "pipeline" else 1)) / pmc["faces_per_launch"])
    if hero_n:
        avg_ms = hero_ms / hero_n
        achieved = HERO_FLOP_PER_FACE * faces_per_launch / (avg_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "%s %s" % ("conv3x3_halo_split_kernel<128,4,2>" if args.dtype == "f32s" else "conv_gemm_kernel<%s,128,128>" % args.dtype, HERO_TAG),
                    "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic,
                    "algorithmic_bytes": int(2 * 64 * 64 * 128 * 4 * faces_per_launch) if args.dtype != "f16" else int(2 * 64 * 64 * 128 * 2 * faces_per_launch),
                    "faces_per_launch": faces_per_launch,
                    "avg_launch_ms": round(avg_ms, 4), "launches": hero_n,
                    "executed_mfma_tflops": round(achieved * MFMA_INSTR_PER_PRODUCT[args.dtype], 2),
                    "executed_mfma_frac": round(achieved * MFMA_INSTR_PER_PRODUCT[args.dtype] / PEAK_TFLOPS[args.dtype], 4)}

    if args.dump_profile and rank == 0:
        with open(args.dump_profile, "w") as f:
            json.dump({"steps": 3, "faces_per_step": faces_per_step, "dtype": args.dtype, "workload": workload,
                       "kernels": {k: {"ms_per_step": v[0] / 3, "launches_per_step": v[1] / 3}
                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}, f, indent=1)
    latency = None
    if workload == "pipeline" and not args.no_probes:
        l1 = state.latency_p50(1)
        lN = state.latency_p50(args.frames // lanes)
        latency = {"single_frame_call_ms_p50": round(l1[0], 4), "single_frame_call_ms_p99": round(l1[1], 4),
                   "lane_batch_call_ms_p50": round(lN[0], 4), "lane_batch_frames": args.frames // lanes,
                   "ms_per_frame_p50_at_lane_batch": round(lN[0] / (args.frames // lanes), 4),
                   "note": "synchronous pf_run_frames call on one stream, device-resident frames"}
    # ---- PCIe-inclusive rate (never the headline): same step with the frames in page-locked HOST memory ----------
    pcie = None
    if workload == "pipeline" and rank == 0 and world == 1 and hasattr(state, "enable_host_frames") and not args.no_probes:
        state.enable_host_frames()
        hs = max(2, min(args.steps, 8))
        state.step_host()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(hs):
            state.step_host()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        state.check()
        pcie = {"faces_per_s": round(faces_per_step * hs / dt, 1), "frames_per_s": round(args.frames * hs / dt, 1),
                "h2d_GBps": round(args.frames * hs * args.frame_hw[0] * args.frame_hw[1] * 3 / dt / 1e9, 2), "steps": hs,
                "note": "frames handed over in pf_host_alloc (page-locked) host memory, copied inside the call on each lane's "
                        "stream; results stay on the device (9.5 KB/frame)"}
    ms_per_step = elapsed / args.steps * 1e3
    faces_total = faces_per_step * world * args.steps
    value = faces_total / elapsed
    out = {
        "metric": "faces/sec (whole node), %s@256" % args.model.capitalize() + ((" %dpx%d-face full pipeline" % (args.frame_hw[0], args.faces_per_frame)) if workload == "pipeline" else " landmark-only"),
        "value": round(value, 1), "unit": "faces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "f16": "f16", "f32s": "f32 (tensors f32; convs = 3x f16-MFMA split precision, f32 accumulate)"}[args.dtype],
        "data": "synthetic",
        "config": {"workload": ("%s full pipeline: %d x %dx%d frames x %d planted faces per GPU per step" % (
                       "configs[2]" if tuple(args.frame_hw) == (1080, 1920) else "configs[4]-shaped", args.frames, args.frame_hw[1], args.frame_hw[0], args.faces_per_frame))
                   if workload == "pipeline" else ("configs[1] landmark-only: %d pre-cropped 256x256 faces per GPU per step" % args.batch),
                   "faces_per_step_per_gpu": faces_per_step, "parallelism": "frame-sharded x%d GPUs, %d HIP streams per GPU, no data-path collective" % (world, lanes),
                   "weights": "synthetic (reference .onnx blobs absent), RCCL broadcast of %.1f MB in %.2f ms%s" % (
                       sum(len(b) for b in blobs.values()) / 1e6, bcast_ms,
                       (" = %.1f GB/s" % (sum(len(b) for b in blobs.values()) / 1e9 / (bcast_ms * 1e-3))) if bcast_ms > 0 else "")},
        "roofline": roofline,
        "cpu_baseline": None,
        "extra": {"ms_per_frame": round(ms_per_step / args.frames, 4) if workload == "pipeline" else None,
                  "algorithmic_tflops": round(value * GFLOP_PER_FACE[args.model] / 1e3, 2),
                  "frac_of_conv_roofline": round(value / world * GFLOP_PER_FACE[args.model] / 1e3 / PEAK_TFLOPS[args.dtype], 4),
                  "latency": latency,
                  "pcie_inclusive": pcie,
                  "setup_s": round(setup_s, 2),
                  "kernel_ms_per_lane_step": {k: round(v[0] / 3, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]}},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(workload, args.cpu_faces)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if hasattr(state, "close"):
        state.close()
    else:
        eng.close()


if __name__ == "__main__":
    main()
