#!/bin/bash
# guard cost: production (words one per 128-byte line) vs the same sources with the words packed (stride 1), twice; then the
# ablation build with the guard on / off (PEPPA_DBG=32768 switches only the guard off) as the bound
export TMPDIR=/tmp
bash tools/gpu_r05_h.sh r05_run28a stride1
bash tools/gpu_r05_h.sh r05_run28b stride1
python tools/ab_env.py "sepup_c280,mbx,unit_s,mbconv,stem,conv3x3_c128,argmax" - PEPPA_DBG=32768 - PEPPA_DBG=32768 2>&1 | grep -v "^    \["
