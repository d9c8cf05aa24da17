#!/bin/bash
# runs a prebuilt microbenchmark (tools/_ub/<name>, built here by hipcc: see the .hip file's header) on the GPU box
mkdir -p gpurun_out
for i in 1 2; do ./tools/_ub/$1 ${2:-256} 2>&1 | tee -a gpurun_out/${3:-ub}_$1.txt; done
