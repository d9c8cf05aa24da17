#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run19}
for v in prod noslp; do
  L=$PWD/peppa_pig_face_landmark_amd/libpeppa_hip.so; [ $v = noslp ] && L=$PWD/tools/_variants/libnoslp.so
  PEPPA_HIP_LIBRARY=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probes --dump-profile gpurun_out/${T}_kernel_table_$v.json > gpurun_out/${T}_bench_$v.json 2> gpurun_out/${T}_bench_$v.err
  tail -c 200 gpurun_out/${T}_bench_$v.err | grep -i "error\|assert"
done
python - <<PY
import json
a=json.load(open("gpurun_out/${T}_kernel_table_prod.json"))["kernels"]; b=json.load(open("gpurun_out/${T}_kernel_table_noslp.json"))["kernels"]
da=json.loads(open("gpurun_out/${T}_bench_prod.json").read().strip().splitlines()[-1]); db=json.loads(open("gpurun_out/${T}_bench_noslp.json").read().strip().splitlines()[-1])
print("prod", da["value"], da["extra"]["lane_step_ms_serial"], "| noslp", db["value"], db["extra"]["lane_step_ms_serial"])
for n in a:
    if n in b and abs(b[n]["ms_per_step"]-a[n]["ms_per_step"]) > 0.004: print("%-40s %.4f -> %.4f" % (n, a[n]["ms_per_step"], b[n]["ms_per_step"]))
PY
