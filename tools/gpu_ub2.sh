#!/bin/bash
mkdir -p gpurun_out
./tools/_ub/ub_sepup 256 | grep -E "up|shipped|VCOL" | tee -a gpurun_out/$1_ub_sepup_ablate.txt
for d in 4 8 12 28 31; do ./tools/_ub/ub_sepup_ablate 256 $d 2>&1 | grep -E "ABLATION|shipped|VCOL" | tee -a gpurun_out/$1_ub_sepup_ablate.txt; done
