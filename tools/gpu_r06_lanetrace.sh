#!/bin/bash
# round 6: kernel trace of the FINAL build, two lanes + front engine against one lane of 48 frames (tools/lane_trace.py: kernels in flight, stretch under contention)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_run26}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof3 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table > /tmp/prof3.out 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table --lanes 1 --frames 48 > /tmp/prof1.out 2>&1
cd $GRAFT_REPO_ROOT
python tools/lane_trace.py /tmp/prof3 /tmp/prof1 gpurun_out/${T}_lane_trace_2lanes_front.md
head -40 gpurun_out/${T}_lane_trace_2lanes_front.md
tail -2 /tmp/prof3.out
