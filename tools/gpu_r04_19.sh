#!/bin/bash
# round 4, GPU session 19: channel-pair depthwise epilogue (v_pk_fma_f32) in the 16 x 16 expand+depthwise kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run21}
( timeout 1500 python -m pytest tests/test_gpu_landmark.py -x -q -m gpu -k "student or fused" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) | tee gpurun_out/${T}_pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"))
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in k.items() if "expdw" in n or "conv1x1" in n})
print("expdw sum", sum(v["ms_per_step"] for n,v in k.items() if "expdw" in n or "conv1x1" in n))
PY
