#!/bin/bash
# round 4, GPU session 9: tile sweep of the detector unit / C3 kernels (ablation build), counters of the new kernels and of the hero
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run9}
AB_BENCH_ARGS="--lanes 1 --frames 32" timeout 1200 python tools/ab_env.py "unit_s1,c3_c128_t2_48,c3_c192" "-" "PEPPA_DET_TILE=2,20" "PEPPA_DET_TILE=4,20" "PEPPA_DET_TILE=1,40" "PEPPA_DET_TILE=2,40" "PEPPA_DET_TILE=4,40" "PEPPA_DET_TILE=6,10" "PEPPA_DET_TILE=3,10" "PEPPA_DET_TILE=6,20" "PEPPA_DET_TILE=8,20" "PEPPA_DET_TILE=2,80" 2>&1 | grep -v "^    \[" | tee gpurun_out/${T}_det_tile_sweep.txt
