#!/bin/bash
# round 6: lanes 2 x 48 against 3 x 32 frames (front engine on), same box, alternating
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}
for rep in 1 2 3; do for L in 3 2 4; do
  timeout 600 python bench.py --steps 30 --warmup 5 --lanes $L --no-cpu-baseline --no-probes --no-kernel-table > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - <<PY | tee -a gpurun_out/${T}_lanes_ab.txt
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("lanes $L", "VALUE", d["value"], "ms/step", d["ms_per_step"])
PY
done; done
