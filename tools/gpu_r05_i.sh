#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
for v in prod noslp; do
  L=$PWD/peppa_pig_face_landmark_amd/libpeppa_hip.so; [ $v = noslp ] && L=$PWD/tools/_variants/libnoslp.so
  PEPPA_HIP_LIBRARY=$L timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probes --no-kernel-table 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
