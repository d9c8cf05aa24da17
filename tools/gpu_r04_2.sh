#!/bin/bash
# round 4, GPU session 2: workgroup-level detector kernels -- parity, then bench + per-kernel table
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run2}
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_round2.py tests/test_batch_runner.py -x -q 2>&1 | tail -8 ) | tee gpurun_out/${T}_pytest_pipeline.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "sustained", d["extra"]["sustained"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
for n,v in k.items():
    print("%-50s %.4f %d" % (n, v["ms_per_step"], v["launches_per_step"]))
PY
