#!/bin/bash
# round 6, final GPU session: the whole GPU test tier, the default bench (+ CPU baseline), landmark-only, rocprofv3 kernel stats (1 lane), counters
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_run20}
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) | tee gpurun_out/${T}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source"))
print("jpeg", d["extra"]["jpeg_ingest"]); print("pcie", d["extra"]["pcie_inclusive"]); print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
print("other", d["extra"]["other_configs"])
PY
timeout 300 python bench.py --workload landmark --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_landmark.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_landmark.json').read().strip().splitlines()[-1]); print('LANDMARK-ONLY', d['value'], d['ms_per_step'])"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-probes --no-cpu-baseline --no-kernel-table --lanes 1 --frames 32 > /tmp/prof1.out 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py /tmp/prof1 gpurun_out/${T}_rocprofv3_kernel_stats_1lane.md > /dev/null && head -12 gpurun_out/${T}_rocprofv3_kernel_stats_1lane.md
timeout 600 python tools/pmc_kernel.py conv3x3_hero_kernel --out=${T}_pmc_hero_sq > /dev/null 2>&1
timeout 300 python tools/pmc_kernel.py conv3x3_hero_kernel --counters=FETCH_SIZE,WRITE_SIZE --out=${T}_pmc_hero_mem > /dev/null 2>&1
python - <<PY
import json
a=json.load(open("gpurun_out/${T}_pmc_hero_sq.json")); b=json.load(open("gpurun_out/${T}_pmc_hero_mem.json"))
k=[x for x in a if "conv3x3_hero_kernel" in x][0]
rec=dict(a[k]); rec.update(b.get(k, {}))
json.dump({"_meta": {"round": 6, "faces_per_launch": 256, "tool": "tools/pmc_kernel.py (one rocprofv3 --pmc pass per counter group; FETCH_SIZE / WRITE_SIZE in KB)"}, k: rec}, open("gpurun_out/${T}_pmc_hero.json", "w"), indent=1)
print("hero", {c: rec.get(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA")})
PY
timeout 300 python tools/pmc_groups.py gpurun_out/${T}_pmc_groups.json
timeout 300 python tools/pmc_kernel.py "mbx_kernel<16, 5, 10, 5, 2, 3>" --out=${T}_pmc_mbxS_sq > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/${T}_pmc_mbxS_sq.json')); [print(k[:60], {c: int(v) for c, v in r.items()}) for k, r in d.items()]"
timeout 300 python tools/pmc_kernel.py pw_head_kernel --out=${T}_pmc_pw_head_sq > /dev/null 2>&1
timeout 300 python tools/pmc_all.py gpurun_out/${T}_pmc_all_kernels.json > gpurun_out/${T}_pmc_all_kernels.txt 2>&1; head -14 gpurun_out/${T}_pmc_all_kernels.txt
timeout 400 python bench.py --model teacher --frame-hw 2160 3840 --faces-per-frame 32 --frames 12 --steps 8 --warmup 2 --no-cpu-baseline --no-probes > gpurun_out/${T}_bench_c5_teacher_f12.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_c5_teacher_f12.json').read().strip().splitlines()[-1]); print('C5 teacher frames 12:', d['value'], d['ms_per_step'], d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac']))"
