#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}
( timeout 1500 python -m pytest tests/test_fused_blocks.py tests/test_gpu_landmark.py tests/test_gpu_race_net.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) | tee gpurun_out/${T}_pytest.log
for i in 1 2 3; do
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probes --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || tail -5 gpurun_out/${T}_bench.err
python - <<PY | tee -a gpurun_out/${T}_summary.txt
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1]); e=d["extra"]
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], "front", e.get("front_step_ms_serial"), "step serial", e.get("step_ms_serial"), "overlap", e["lanes_overlap"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in k.items() if n.startswith("mbx")})
PY
done
