#!/bin/bash
# round 4, GPU session 10: does giving every lane its own share of the CUs (hipExtStreamCreateWithCUMask) beat time-sharing?
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run10}
( AB_BENCH_ARGS="--lanes 3 --frames 96" timeout 900 python tools/ab_env.py "hero_none" "-" "PEPPA_CU_PARTITION=3,1" "PEPPA_CU_PARTITION=3,2" "PEPPA_CU_PARTITION=3,3" 2>&1 | grep -v "^    \["
  AB_BENCH_ARGS="--lanes 2 --frames 64" timeout 600 python tools/ab_env.py "hero_none" "-" "PEPPA_CU_PARTITION=2,1" "PEPPA_CU_PARTITION=2,2" 2>&1 | grep -v "^    \["
  AB_BENCH_ARGS="--lanes 4 --frames 128" timeout 600 python tools/ab_env.py "hero_none" "-" "PEPPA_CU_PARTITION=4,1" "PEPPA_CU_PARTITION=4,2" 2>&1 | grep -v "^    \[" ) | tee gpurun_out/${T}_cu_partition.txt
