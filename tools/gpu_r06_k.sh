#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_runX}; shift
( timeout 1500 python -m pytest tests/test_gpu_landmark.py tests/test_gpu_race_net.py tests/test_batch_runner.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) | tee gpurun_out/${T}_pytest.log
for mode in new old new old new old; do
  extra=""; [ $mode = old ] && extra="$*"
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probes $extra --dump-profile /tmp/kt.json > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - <<PY | tee -a gpurun_out/${T}_ab.txt
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); e=d["extra"]
k=json.load(open("/tmp/kt.json"))["kernels"]
print("$mode [$extra]", "VALUE", d["value"], "ms/step", d["ms_per_step"], "lane serial", e["lane_step_ms_serial"], "step serial", e.get("step_ms_serial"), {n: round(v["ms_per_step"],4) for n,v in k.items() if n in ("fc","gap","sepup_c296_n256_32x32")})
PY
done
