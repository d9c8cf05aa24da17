#!/bin/bash
# round 4, GPU session 7: fused encoder front end (lm_front) + persistent StemBlock kernel: parity, bench, A/B of the register cap
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run7}
( timeout 1200 python -m pytest tests/test_gpu_landmark.py tests/test_gpu_pipeline.py tests/test_batch_runner.py tests/test_fused_blocks.py -x -q -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) | tee gpurun_out/${T}_pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
k=json.load(open("gpurun_out/${T}_kernel_table.json"))["kernels"]
print({n: round(v["ms_per_step"],4) for n,v in list(k.items())[:24]})
PY
timeout 300 python bench.py --workload landmark --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_landmark.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_landmark.json').read().strip().splitlines()[-1]); print('LANDMARK-ONLY', d['value'], d['ms_per_step'])"
AB_BENCH_ARGS="--lanes 1 --frames 32" timeout 600 python tools/ab_env.py "lm_front,stem_block" "-" "PEPPA_DBG=8192" 2>&1 | tail -4 | tee gpurun_out/${T}_lm_front_ab.txt
