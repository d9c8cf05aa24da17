#!/bin/bash
# round 4, GPU session 20: where the 16 x 16 expand + depthwise launches spend their time (ablation build, PEPPA_DBG bits)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run20}
timeout 900 python tools/ab_env.py "expdw5x5d2,expdw5x5d1_c112,expdw3x3d1_c112,conv1x1_c960" "-" "PEPPA_DBG=16" "PEPPA_DBG=256" "PEPPA_DBG=512" "PEPPA_DBG=32" "PEPPA_DBG=1024" "PEPPA_DBG=1056" "PEPPA_DBG=1840" 2>&1 | tail -12 | tee gpurun_out/${T}_expdw_ablations.txt
