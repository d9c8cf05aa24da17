#!/bin/bash
# round 4, GPU session 42: hero conv with a three-stage weight ring (k_hero.h): parity, kernel table
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run42}
( timeout 900 python -m pytest tests/test_gpu_landmark.py tests/test_basic_chain.py tests/test_gpu_pipeline.py -x -q -m gpu -k "student or narrow or planted or run_frames" 2>&1 | tail -4 ) | tee gpurun_out/${T}_pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probes --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in k.items() if "argmax" in n or "conv3x3_c128" in n or "sepup_c280" in n})
PY
