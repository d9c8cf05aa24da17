#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}
for rep in 1 2; do for cfg in "96 2" "128 2" "64 2" "96 3" "128 4" "64 1 --batch-engine"; do
  set -- $cfg; F=$1; L=$2; shift; shift
  timeout 600 python bench.py --steps 20 --warmup 5 --frames $F --lanes $L $* --no-cpu-baseline --no-probes --no-kernel-table > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - <<PY | tee -a gpurun_out/${T}_shapes.txt
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("frames $F lanes $L $*", "VALUE", d["value"], "ms/step", d["ms_per_step"])
PY
done; done
