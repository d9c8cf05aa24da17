#!/bin/bash
# whole-pipeline faces/s for (frames per step, lanes) splits, with and without hipGraph replay -- run on the GPU box
# usage: tools/sweep_lanes.sh <out.txt> ["F L" ...]
OUT=${1:-gpurun_out/sweep_lanes.txt}; shift
CFGS=("$@")
[ ${#CFGS[@]} -eq 0 ] && CFGS=("96 3" "96 6" "96 12" "48 3" "48 6" "24 3" "192 6" "192 12")
: > $OUT
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  for g in "" "--no-graph"; do
    python bench.py --frames $1 --lanes $2 --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-kernel-table $g 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('frames %4d lanes %2d (%2d per lane) %-10s -> %8.0f faces/s  %.2f ms/step' % ($1, $2, $1//$2, '$g' or 'graph', d['value'], d['ms_per_step']))" | tee -a $OUT
  done
done
