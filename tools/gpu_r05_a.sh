#!/bin/bash
# round 5, session A: the race-net tests (new), the whole GPU tier, the default bench with the kernel table (start-of-round numbers on this round's box)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run1}
( timeout 900 python -m pytest tests/test_gpu_race_net.py tests/test_batch_runner.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) | tee gpurun_out/${T}_pytest_race_net.log
( timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_race_net.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) | tee gpurun_out/${T}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("other", d["extra"].get("other_configs"))
print("argmax", d["extra"]["dense_kernels"].get("conv1x1_argmax_c128_n98_64x64"))
PY
