#!/bin/bash
# round 4, GPU session 8: per-phase cycles of lm_front / det_unit (ablation build), hardware-queue count, fused front on/off
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run8}
AB_BENCH_ARGS="--lanes 1 --frames 32" timeout 600 python tools/ab_env.py "lm_front,stem_block" "PEPPA_DBG=4096" 2>&1 | tail -10 | tee gpurun_out/${T}_phase_cycles.txt
q() { python bench.py --steps 12 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%8.0f faces/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
( echo -n "fuse-front 1, default queues : "; q --fuse-front 1
  echo -n "fuse-front 0, default queues : "; q --fuse-front 0
  echo -n "fuse-front 0, GPU_MAX_HW_QUEUES=8, 3 lanes: "; GPU_MAX_HW_QUEUES=8 q --fuse-front 0
  echo -n "fuse-front 0, GPU_MAX_HW_QUEUES=8, 4 lanes: "; GPU_MAX_HW_QUEUES=8 q --fuse-front 0 --lanes 4 --frames 128
  echo -n "fuse-front 0, GPU_MAX_HW_QUEUES=8, 6 lanes: "; GPU_MAX_HW_QUEUES=8 q --fuse-front 0 --lanes 6 --frames 96
  echo -n "fuse-front 0, GPU_MAX_HW_QUEUES=2, 3 lanes: "; GPU_MAX_HW_QUEUES=2 q --fuse-front 0
  echo -n "fuse-front 0, default queues (repeat): "; q --fuse-front 0 ) | tee gpurun_out/${T}_queues_and_front.txt
