#!/bin/bash
# round 4, GPU session 1: baseline bench of the round-3 build, lane sweep (both directions), 3-lane vs 1-lane kernel traces
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04_run1
timeout 400 python bench.py --steps 20 --warmup 3 --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 600 gpurun_out/${T}_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1]); print('VALUE', d['value'], d['ms_per_step'], d['extra']['lane_step_ms_serial'])"
timeout 900 tools/sweep_lanes.sh gpurun_out/${T}_sweep_lanes.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof3 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table > /tmp/prof3.out 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table --lanes 1 --frames 32 > /tmp/prof1.out 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof6 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table --lanes 6 --frames 96 > /tmp/prof6.out 2>&1
cd $GRAFT_REPO_ROOT
python tools/lane_trace.py /tmp/prof3 /tmp/prof1 gpurun_out/${T}_lane_trace_3lanes.md
python tools/lane_trace.py /tmp/prof6 gpurun_out/${T}_lane_trace_6lanes.md
tail -3 /tmp/prof3.out
