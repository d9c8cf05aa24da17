#!/bin/bash
# GPU box: landmark parity tests + one profiled bench run; prints throughput and the N slowest kernel tags
python -m pytest tests/test_gpu_landmark.py -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dump-profile gpurun_out/prof_q.json > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_q.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
k=json.load(open("gpurun_out/prof_q.json"))["kernels"]
print({n: round(v["ms_per_step"],3) for n,v in list(k.items())[:${1:-30}]})
print("sum", round(sum(v["ms_per_step"] for v in k.values()),3))
PY
