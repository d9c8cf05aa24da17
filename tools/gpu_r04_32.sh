#!/bin/bash
# round 4, final GPU session: the whole GPU test tier, the default bench, rocprofv3 kernel stats (1 lane), counters of the hero and of the new detector kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04_run32}
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) | tee gpurun_out/${T}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "one lane", d["extra"].get("one_lane_faces_per_s"), "sustained", d["extra"]["sustained"])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source"))
print("jpeg", d["extra"]["jpeg_ingest"]); print("pcie", d["extra"]["pcie_inclusive"]); print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
PY
timeout 300 python bench.py --workload landmark --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_landmark.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_landmark.json').read().strip().splitlines()[-1]); print('LANDMARK-ONLY', d['value'], d['ms_per_step'])"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-probes --no-cpu-baseline --no-kernel-table --lanes 1 --frames 32 > /tmp/prof1.out 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py /tmp/prof1 gpurun_out/${T}_rocprofv3_kernel_stats_1lane.md > /dev/null && head -8 gpurun_out/${T}_rocprofv3_kernel_stats_1lane.md
timeout 600 python tools/pmc_kernel.py conv3x3_hero_kernel --out=${T}_pmc_hero_sq > /dev/null 2>&1
timeout 300 python tools/pmc_kernel.py conv3x3_hero_kernel --counters=FETCH_SIZE,WRITE_SIZE --out=${T}_pmc_hero_mem > /dev/null 2>&1
DET="--workload pipeline --steps 2 --warmup 1 --no-cpu-baseline --no-probes --no-kernel-table --lanes 1 --frames 32"
timeout 600 python tools/pmc_kernel.py det_ --out=${T}_pmc_det_sq $DET > /dev/null 2>&1
timeout 300 python tools/pmc_kernel.py det_ --counters=FETCH_SIZE,WRITE_SIZE --out=${T}_pmc_det_mem $DET > /dev/null 2>&1
python - <<PY
import json
a=json.load(open("gpurun_out/${T}_pmc_hero_sq.json")); b=json.load(open("gpurun_out/${T}_pmc_hero_mem.json"))
k=[x for x in a if "conv3x3_hero_kernel" in x][0]
rec=dict(a[k]); rec.update(b.get(k, {}))
json.dump({"_meta": {"round": 4, "faces_per_launch": 256, "tool": "tools/pmc_kernel.py (one rocprofv3 --pmc pass per counter group; FETCH_SIZE / WRITE_SIZE in KB)"}, k: rec}, open("gpurun_out/${T}_pmc_hero.json", "w"), indent=1)
print("hero", {c: rec.get(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA")})
d=json.load(open("gpurun_out/${T}_pmc_det_mem.json"))
for kk, v in d.items(): print(kk[:60], v)
PY
# Teacher: landmark-only table, config-5 shape at 6 / 12 / 24 frames per step, counters of the round's Teacher kernels
timeout 400 python bench.py --model teacher --workload landmark --batch 256 --steps 10 --warmup 2 --no-cpu-baseline --dump-profile gpurun_out/${T}_teacher_b256_kernel_table.json > gpurun_out/${T}_bench_teacher_landmark_b256.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_teacher_landmark_b256.json').read().strip().splitlines()[-1]); print('TEACHER landmark-only b256', d['value'], d['ms_per_step'], d['roofline'].get('kernel'))"
for fr in 6 12 24; do
  timeout 600 python bench.py --model teacher --frame-hw 2160 3840 --faces-per-frame 32 --frames $fr --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-kernel-table > gpurun_out/${T}_bench_c5_teacher_f$fr.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_c5_teacher_f$fr.json').read().strip().splitlines()[-1]); print('C5 teacher frames $fr:', d['value'], d['ms_per_step'])"
done | tee gpurun_out/${T}_c5_frames_sweep.txt
TEA="--model teacher --workload landmark --batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-probes --no-kernel-table"
timeout 600 python tools/pmc_kernel.py hr_bottleneck_kernel --out=${T}_pmc_hrb_sq $TEA > /dev/null 2>&1
timeout 300 python tools/pmc_kernel.py hr_bottleneck_kernel --counters=FETCH_SIZE,WRITE_SIZE --out=${T}_pmc_hrb_mem $TEA > /dev/null 2>&1
timeout 300 python tools/pmc_kernel.py fuse_up_kernel --counters=FETCH_SIZE,WRITE_SIZE --out=${T}_pmc_fuse_up_mem $TEA > /dev/null 2>&1
python - <<PY
import json
for f in ("gpurun_out/${T}_pmc_hrb_sq.json", "gpurun_out/${T}_pmc_hrb_mem.json", "gpurun_out/${T}_pmc_fuse_up_mem.json"):
    try:
        d=json.load(open(f))
        for kk, v in d.items(): print(kk[:70], {c: v.get(c) for c in list(v)[:8]})
    except Exception as e: print(f, e)
PY
