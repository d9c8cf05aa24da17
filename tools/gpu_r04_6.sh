#!/bin/bash
# round 4, GPU session 6: parity of the changed kernels, bench, 1-lane kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run6}
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_round2.py tests/test_batch_runner.py tests/test_range_guard.py -x -q 2>&1 | tail -15 ) | tee gpurun_out/${T}_pytest_pipeline.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in k.items() if any(s in n for s in ("unit_","c3_","stem","fc","gap"))})
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-probes --no-cpu-baseline --no-kernel-table --lanes 1 --frames 32 > /tmp/prof1.out 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py /tmp/prof1 gpurun_out/${T}_rocprofv3_kernel_stats_1lane.md > /dev/null && grep -E "det_|stem|verdict" gpurun_out/${T}_rocprofv3_kernel_stats_1lane.md
