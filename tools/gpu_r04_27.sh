#!/bin/bash
# round 4, GPU session 27: config-5 shape at 24 frames per step; Student headline re-check on the committed build
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run27}
for cfg in "3 24" "2 24" "3 12"; do
  set -- $cfg
  timeout 600 python bench.py --model teacher --frame-hw 2160 3840 --faces-per-frame 32 --frames $2 --lanes $1 --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-kernel-table > gpurun_out/${T}_bench_c5_teacher_l$1_f$2.json 2>gpurun_out/${T}_c5.err
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_c5_teacher_l$1_f$2.json').read().strip().splitlines()[-1]); print('C5 teacher lanes $1 frames $2:', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/${T}_c5.err
done | tee gpurun_out/${T}_c5_frames_sweep.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probes --no-kernel-table > gpurun_out/${T}_bench.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1]); print('STUDENT', d['value'], d['ms_per_step'])"
