#!/bin/bash
# round 6, session C: lane shapes with the front engine (the detector no longer rides in the lanes)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_run3}
for cfg in "96 2" "96 3" "96 4" "96 6" "128 4" "192 3" "192 6" "64 2"; do
  set -- $cfg
  timeout 300 python bench.py --frames $1 --lanes $2 --steps 12 --warmup 4 --no-cpu-baseline --no-probes --no-kernel-table > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python - <<PY | tee -a gpurun_out/${T}_sweep_lanes_front.txt
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("frames %4d lanes %2d (%d per lane) -> %8.0f faces/s  %.2f ms/step" % ($1, $2, $1//$2, d["value"], d["ms_per_step"]))
PY
done
