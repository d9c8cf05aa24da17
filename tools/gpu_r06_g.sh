#!/bin/bash
# round 6: same-box A/B of bench flags: "new" = defaults, "old" = the flags given
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}; shift
for mode in new old new old new old; do
  extra=""; [ $mode = old ] && extra="$*"
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probes $extra > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - <<PY | tee -a gpurun_out/${T}_ab.txt
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); e=d["extra"]
print("$mode [$extra]", "VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], "step serial", e.get("step_ms_serial"), "overlap", e["lanes_overlap"])
PY
done
