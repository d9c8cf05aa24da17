#!/bin/bash
# round 6: whole GPU tier + bench (+ A/B flag)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) | tee gpurun_out/${T}_pytest_gpu.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || tail -5 gpurun_out/${T}_bench.err
python - <<PY | tee -a gpurun_out/${T}_summary.txt
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1]); e=d["extra"]
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], "front", e.get("front_step_ms_serial"), "step serial", e.get("step_ms_serial"), "overlap", e["lanes_overlap"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in list(k.items())[:14]})
PY
done
