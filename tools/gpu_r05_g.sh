#!/bin/bash
# round 5, session G: quick A/B of the block kernels against the layer-wise blocks (determinism, parity, kernel tables)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run17}
python tools/mbx_determinism.py peppa_pig_face_landmark_amd/libpeppa_hip.so 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee gpurun_out/${T}_mbx_determinism.txt
( timeout 900 python -m pytest tests/test_fused_blocks.py tests/test_gpu_race_net.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 ) | tee gpurun_out/${T}_pytest_mbx.log
for v in default off; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probes --mbx $v --dump-profile gpurun_out/${T}_kernel_table_$v.json > gpurun_out/${T}_bench_$v.json 2> gpurun_out/${T}_bench_$v.err
  tail -c 200 gpurun_out/${T}_bench_$v.err | grep -i "error\|assert"
  python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_$v.json").read().strip().splitlines()[-1])
k=json.load(open("gpurun_out/${T}_kernel_table_$v.json"))["kernels"]
g={n: round(x["ms_per_step"],4) for n,x in k.items() if n.startswith("mbx") or n.startswith("expdw") and "16x16" in n or n.startswith("conv1x1_c") and "16x16" in n}
print("$v VALUE", d["value"], "serial", d["extra"]["lane_step_ms_serial"], "group", round(sum(g.values()),4))
print("   ", g)
PY
done 2>&1 | tee gpurun_out/${T}_mbx_ab.txt
