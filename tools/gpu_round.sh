#!/bin/bash
# One GPU-box session: parity tests, default bench (JSON + per-kernel table), rocprofv3 kernel stats of the same command.
# usage: tools/gpu_round.sh <tag> [pytest-args...]
TAG=${1:-r02}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15 ) > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 --dump-profile gpurun_out/${TAG}_kernel_table.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("VALUE", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"] and d["roofline"]["frac"], d["roofline"] and d["roofline"]["avg_launch_ms"])
    print("serial lane-step ms", d["extra"]["lane_step_ms_serial"], "sustained", d["extra"]["sustained"])
    print("hbm_ops", {k:(v["ms_per_step"], v["achieved_GBps"]) for k,v in (d["extra"]["hbm_ops"] or {}).items()})
    print("dense", {k:(v["ms_per_lane_step"], v["executed_mfma_frac"]) for k,v in (d["extra"]["dense_kernels"] or {}).items()})
    print("bcast", d["extra"]["weight_broadcast"])
    print("cpu", d["cpu_baseline"])
except Exception as e:
    print("bench parse failed", e)
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-probes --no-cpu-baseline --lanes 1 --frames 32 > /tmp/prof_${TAG}.out 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py /tmp/prof_${TAG} gpurun_out/${TAG}_rocprofv3_kernel_stats_1lane.md && head -12 gpurun_out/${TAG}_rocprofv3_kernel_stats_1lane.md
