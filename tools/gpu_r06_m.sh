#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}
( timeout 1500 python -m pytest tests/test_gpu_landmark.py tests/test_gpu_race_net.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|three products|Error" | tail -6 ) | tee gpurun_out/${T}_pytest.log
for mix in "" "hero" "hero,head" "" "hero" "hero,head"; do
  extra=""; [ -n "$mix" ] && extra="--mix $mix"
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probes $extra --dump-profile /tmp/kt.json > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - <<PY | tee -a gpurun_out/${T}_mix_ab.txt
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); e=d["extra"]
k=json.load(open("/tmp/kt.json"))["kernels"]
print("mix [$mix]", "VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], {n: round(v["ms_per_step"],4) for n,v in k.items() if n in ("conv3x3_c128_n128_64x64","conv1x1_argmax_c128_n98_64x64")})
PY
done
