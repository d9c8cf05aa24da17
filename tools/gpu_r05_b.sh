#!/bin/bash
# round 5, session B: the block kernels of k_mbx.h -- parity on the GPU, then the default bench with the kernel table
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run2}
( timeout 900 python -m pytest tests/test_fused_blocks.py tests/test_gpu_race_net.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) | tee gpurun_out/${T}_pytest_mbx.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
for n,v in k.items():
    if n.startswith("mbx") or n.startswith("expdw") or n.startswith("conv1x1_c") or n in ("fc","gap"): print("%-40s %.4f ms  x%.0f" % (n, v["ms_per_step"], v["launches_per_step"]))
PY
