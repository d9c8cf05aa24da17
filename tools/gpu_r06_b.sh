#!/bin/bash
# round 6, session B: the front engine (detector + NMS once per step) against the per-lane flow
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_run2}
( timeout 900 python -m pytest tests/test_batch_runner.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) | tee gpurun_out/${T}_pytest_batch.log
for mode in front nofront front nofront; do
  extra=""; [ $mode = nofront ] && extra="--no-front"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes $extra --dump-profile gpurun_out/${T}_kernel_table_$mode.json > gpurun_out/${T}_bench_$mode.json 2> gpurun_out/${T}_bench_$mode.err || tail -5 gpurun_out/${T}_bench_$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_$mode.json").read().strip().splitlines()[-1])
e=d["extra"]
print("$mode", "VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], "front", e.get("front_step_ms_serial"), "step serial", e.get("step_ms_serial"), "overlap", e["lanes_overlap"], "fwd frac", e.get("executed_mfma_frac_forward"))
PY
done
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_front.json").read().strip().splitlines()[-1])
print(d["extra"]["front_kernel_ms_per_step"]); print(d["extra"]["hbm_ops"].keys())
PY
