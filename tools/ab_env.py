"""A/B timing of kernel variants selected by environment variables (run on the GPU box).
usage: python tools/ab_env.py "<tag substring,tag substring,...>" "NAME=VAL NAME2=VAL" "NAME=VAL" ...
Each remaining argument is one variant: a space-separated list of environment assignments ("-" = none).
Prints whole-pipeline faces/s and the per-lane-step ms of every kernel tag containing one of the substrings."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from peppa_pig_face_landmark_amd import build as _b  # noqa: E402
# PEPPA_DBG exists in the ablation flavour of the library only (-DPF_ABLATE=1); build it in-tree BEFORE going to the GPU box
os.environ["PEPPA_HIP_LIBRARY"] = _b.build_hip(ablate=True, verbose=False)
subs = sys.argv[1].split(",")
extra = os.environ.get("AB_BENCH_ARGS", "").split()
os.makedirs("gpurun_out", exist_ok=True)
for vi, spec in enumerate(sys.argv[2:]):
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split():
            k, v = kv.split("=", 1)
            env[k] = v
    out = "gpurun_out/ab_%d.json" % vi
    r = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-probes",
                        "--dump-profile", out] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        k = json.load(open(out))["kernels"]
        sel = {n: round(v["ms_per_step"], 4) for n, v in k.items() if any(s in n for s in subs)}
        print("%-40s %8.0f faces/s  serial %.3f ms  %s" % (spec, d["value"], d["extra"]["lane_step_ms_serial"], sel), flush=True)
        for l in r.stderr.splitlines():
            if l.startswith("[hero_pipe") or l.startswith("[sepup_pipe") or l.startswith("[det_"):     # cycle accounting of the ablation build (PEPPA_DBG & 64)
                print("    " + l, flush=True)
    except Exception as e:  # noqa: BLE001
        print(spec, "FAILED", e, r.stderr[-800:], flush=True)
