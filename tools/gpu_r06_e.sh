#!/bin/bash
# round 6: A/B of a bench flag (e.g. --no-fc-pairs) after the parity tests
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}; FLAG=${2:---no-fc-pairs}
( timeout 1200 python -m pytest tests/test_gpu_landmark.py tests/test_gpu_race_net.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) | tee gpurun_out/${T}_pytest.log
for mode in new old new old; do
  extra=""; [ $mode = old ] && extra="$FLAG"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes $extra --dump-profile gpurun_out/${T}_kernel_table_$mode.json > gpurun_out/${T}_bench_$mode.json 2> gpurun_out/${T}_bench_$mode.err || tail -5 gpurun_out/${T}_bench_$mode.err
  python - <<PY | tee -a gpurun_out/${T}_ab.txt
import json
d=json.loads(open("gpurun_out/${T}_bench_$mode.json").read().strip().splitlines()[-1]); e=d["extra"]
k=json.load(open("gpurun_out/${T}_kernel_table_$mode.json"))["kernels"]
print("$mode $extra", "VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], "step serial", e.get("step_ms_serial"), "fc", {n: (round(v["ms_per_step"],4), v["launches_per_step"]) for n,v in k.items() if n in ("fc","gap","scse")})
PY
done
