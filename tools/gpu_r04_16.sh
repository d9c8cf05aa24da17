#!/bin/bash
# round 4, GPU session 16: XCD-aware tile order (det_unit / det_c3 / hr_bottleneck): parity, kernel tables, PMC of a unit launch
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run16}
( timeout 1500 python -m pytest tests/test_gpu_detector.py tests/test_gpu_landmark.py tests/test_gpu_pipeline.py tests/test_batch_runner.py tests/test_video_ingest.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) | tee gpurun_out/${T}_pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"))
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in k.items() if "det_" in n})
print("det sum", sum(v["ms_per_step"] for n,v in k.items() if "det_" in n))
PY
timeout 400 python bench.py --model teacher --workload landmark --batch 256 --steps 10 --warmup 2 --no-cpu-baseline --dump-profile gpurun_out/${T}_teacher_b256_kernel_table.json > gpurun_out/${T}_bench_teacher_landmark_b256.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_teacher_landmark_b256.json").read().strip().splitlines()[-1])
print("TEACHER landmark-only b256", d["value"], d["ms_per_step"])
k=json.load(open("gpurun_out/${T}_teacher_b256_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],3) for n,v in k.items() if "bottleneck" in n or "hr_" in n}, "sum", sum(v["ms_per_step"] for v in k.values()))
PY
timeout 400 python bench.py --model teacher --frame-hw 2160 3840 --faces-per-frame 32 --frames 6 --steps 10 --warmup 2 --no-cpu-baseline --no-probes > gpurun_out/${T}_bench_c5_teacher_f6.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_c5_teacher_f6.json').read().strip().splitlines()[-1]); print('C5 teacher 4K x 32, 6 frames/step', d['value'], d['ms_per_step'])"
