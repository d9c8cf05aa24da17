#!/bin/bash
# round 4, GPU session 31: fuse_up with an 80 KB tier (two workgroups per CU on the 36-channel branch)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run31}
( timeout 600 python -m pytest tests/test_basic_chain.py -x -q -m gpu -k "fuse_up" 2>&1 | tail -3 ) | tee gpurun_out/${T}_pytest.log
timeout 400 python bench.py --model teacher --workload landmark --batch 256 --steps 10 --warmup 2 --no-cpu-baseline --dump-profile gpurun_out/${T}_teacher_b256_kernel_table.json > gpurun_out/${T}_bench_teacher_landmark_b256.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_teacher_landmark_b256.json").read().strip().splitlines()[-1])
print("TEACHER landmark-only b256", d["value"], d["ms_per_step"])
k=json.load(open("gpurun_out/${T}_teacher_b256_kernel_table.json"))["kernels"]
print({n: (round(v["ms_per_step"],3), v.get("launches_per_step")) for n,v in k.items() if "fuse" in n}, "sum", sum(v["ms_per_step"] for v in k.values()))
PY
