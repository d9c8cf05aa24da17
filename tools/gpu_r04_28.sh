#!/bin/bash
# round 4, GPU session 28: HRNet fuse sums in one launch per output branch (PF_OP_FUSEUP): Teacher parity, kernel table, config-5 shape
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run28}
( timeout 1500 python -m pytest tests/test_basic_chain.py tests/test_gpu_landmark.py tests/test_gpu_pipeline.py -x -q -m gpu -k "bottleneck or teacher or Teacher or c5" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) | tee gpurun_out/${T}_pytest.log
timeout 400 python bench.py --model teacher --workload landmark --batch 256 --steps 10 --warmup 2 --no-cpu-baseline --dump-profile gpurun_out/${T}_teacher_b256_kernel_table.json > gpurun_out/${T}_bench_teacher_landmark_b256.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_teacher_landmark_b256.json").read().strip().splitlines()[-1])
print("TEACHER landmark-only b256", d["value"], d["ms_per_step"])
k=json.load(open("gpurun_out/${T}_teacher_b256_kernel_table.json"))["kernels"]
print({n: (round(v["ms_per_step"],3), v.get("launches_per_step")) for n,v in k.items() if "fuse" in n or "add_up" in n or "conv1x1" in n}, "sum", sum(v["ms_per_step"] for v in k.values()))
PY
for cfg in "3 6" "3 12" "3 24"; do
  set -- $cfg
  timeout 600 python bench.py --model teacher --frame-hw 2160 3840 --faces-per-frame 32 --frames $2 --lanes $1 --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-kernel-table > gpurun_out/${T}_bench_c5_teacher_l$1_f$2.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_c5_teacher_l$1_f$2.json').read().strip().splitlines()[-1]); print('C5 teacher lanes $1 frames $2:', d['value'], d['ms_per_step'])"
done | tee gpurun_out/${T}_c5_frames_sweep.txt
