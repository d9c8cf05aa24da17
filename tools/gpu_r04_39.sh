#!/bin/bash
# round 4, GPU session 39: SQ / memory counters of the Teacher's remaining dominant kernels (evidence for the next round)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run39}
TEA="--model teacher --workload landmark --batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-probes --no-kernel-table"
timeout 600 python tools/pmc_kernel.py basic_block_kernel --out=${T}_pmc_basic_block_sq $TEA > /dev/null 2>&1
timeout 300 python tools/pmc_kernel.py basic_block_kernel --counters=FETCH_SIZE,WRITE_SIZE --out=${T}_pmc_basic_block_mem $TEA > /dev/null 2>&1
timeout 600 python tools/pmc_kernel.py basic_chain_kernel --out=${T}_pmc_basic_chain_sq $TEA > /dev/null 2>&1
python - <<PY
import json
for f in ("gpurun_out/${T}_pmc_basic_block_sq.json", "gpurun_out/${T}_pmc_basic_block_mem.json", "gpurun_out/${T}_pmc_basic_chain_sq.json"):
    try:
        d=json.load(open(f))
        for kk, v in d.items(): print(kk[:60], {c: round(x) for c, x in v.items()})
    except Exception as e: print(f, e)
PY
