#!/bin/bash
# whole-pipeline faces/s for several (frames per step, lanes) splits -- run on the GPU box
for cfg in "96 3" "96 2" "144 3" "192 3" "192 4" "128 2" "256 4" "192 2"; do
  set -- $cfg
  python bench.py --frames $1 --lanes $2 --steps 10 --warmup 3 --no-probes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('frames %4d lanes %d -> %8.0f faces/s  %.2f ms/step  serial lane-step %.2f ms' % ($1, $2, d['value'], d['ms_per_step'], d['extra']['lane_step_ms_serial']))"
done
