#!/bin/bash
# round 4, GPU session 33: host threads per lane of the JPEG-file ingest probe
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run33}
for thr in 4 8 16 2; do
  timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-table --sustain-s 0 --jpeg-threads $thr > gpurun_out/${T}_bench_jpeg_t$thr.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_jpeg_t$thr.json').read().strip().splitlines()[-1]); print('threads $thr: value', d['value'], 'jpeg', d['extra']['jpeg_ingest']['no_restart_markers']['faces_per_s'], 'pcie', d['extra']['pcie_inclusive']['faces_per_s'])"
done | tee gpurun_out/${T}_jpeg_threads_sweep.txt
