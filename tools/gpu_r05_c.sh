#!/bin/bash
# round 5, session C: block kernels (k_mbx.h) -- GPU parity, bench with kernel table, per-phase cycles of the ablation build
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05_run4}
python tools/mbx_determinism.py peppa_pig_face_landmark_amd/libpeppa_hip.so 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee gpurun_out/${T}_mbx_determinism.txt
( timeout 900 python -m pytest tests/test_fused_blocks.py tests/test_gpu_race_net.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 ) | tee gpurun_out/${T}_pytest_mbx.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
tot=0
for n,v in k.items():
    if n.startswith("mbx") or n.startswith("expdw") or n.startswith("conv1x1_c") or n in ("fc","gap"):
        print("%-40s %.4f ms  x%.0f" % (n, v["ms_per_step"], v["launches_per_step"])); tot+=v["ms_per_step"]
print("group total", round(tot,4))
PY
python tools/ab_env.py "mbx" "PEPPA_DBG=64" 2>&1 | grep "det_mbx" | tee gpurun_out/${T}_mbx_phase_cycles.txt
