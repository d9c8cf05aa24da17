#!/bin/bash
# round 5, GPU session 59: kernel trace of the FINAL build, three lanes against one (tools/lane_trace.py: kernels in flight, stretch under contention)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run59}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof3 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table > /tmp/prof3.out 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table --lanes 1 --frames 32 > /tmp/prof1.out 2>&1
cd $GRAFT_REPO_ROOT
python tools/lane_trace.py /tmp/prof3 /tmp/prof1 gpurun_out/${T}_lane_trace_3lanes.md
head -40 gpurun_out/${T}_lane_trace_3lanes.md
tail -2 /tmp/prof3.out
