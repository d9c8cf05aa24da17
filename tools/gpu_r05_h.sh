#!/bin/bash
# per-kernel A/B of compiler-flag variants of the whole library (tools/_variants/lib<name>.so) against the production build
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run21}; shift
VARS="prod $@"
for v in $VARS; do
  L=$PWD/peppa_pig_face_landmark_amd/libpeppa_hip.so; [ $v != prod ] && L=$PWD/tools/_variants/lib$v.so
  PEPPA_HIP_LIBRARY=$L timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probes --dump-profile gpurun_out/${T}_kernel_table_$v.json > gpurun_out/${T}_bench_$v.json 2> gpurun_out/${T}_bench_$v.err
done
python - $T $VARS <<'PY'
import json, sys
T, vs = sys.argv[1], sys.argv[2:]
tabs = {v: json.load(open("gpurun_out/%s_kernel_table_%s.json" % (T, v)))["kernels"] for v in vs}
for v in vs:
    d = json.loads(open("gpurun_out/%s_bench_%s.json" % (T, v)).read().strip().splitlines()[-1])
    print("%-10s %8.0f faces/s  serial lane step %.4f ms" % (v, d["value"], d["extra"]["lane_step_ms_serial"]))
print("%-40s" % "kernel" + "".join("%10s" % v for v in vs))
for n, r in sorted(tabs["prod"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    row = [tabs[v].get(n, {}).get("ms_per_step", float("nan")) for v in vs]
    if max(abs(x - row[0]) for x in row) > 0.003: print("%-40s" % n + "".join("%10.4f" % x for x in row))
PY
