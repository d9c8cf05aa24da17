#!/bin/bash
# round 6, session A: start-of-round baseline + how the detector's kernels scale with frames per launch (32 vs 96, one lane)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06_run1}
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-profile gpurun_out/${T}_kernel_table.json > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], "ms/step", d["ms_per_step"], "serial", d["extra"]["lane_step_ms_serial"], "sustained", d.get("sustained_value"), "fwd frac", d["extra"].get("executed_mfma_frac_forward"))
PY
for F in 32 96; do
timeout 300 python bench.py --lanes 1 --frames $F --steps 6 --warmup 2 --no-cpu-baseline --no-probes --dump-profile gpurun_out/${T}_kernel_table_1lane_f$F.json > gpurun_out/${T}_bench_1lane_f$F.json 2>/dev/null
done
python - <<PY
import json, re
a=json.load(open("gpurun_out/${T}_kernel_table_1lane_f32.json"))["kernels"]; b=json.load(open("gpurun_out/${T}_kernel_table_1lane_f96.json"))["kernels"]
det=lambda t: bool(re.search(r"(24x40|48x80|12x20|stem_block|letterbox|nms)", t))
sa=sum(v["ms_per_step"] for t,v in a.items() if det(t)); sb=sum(v["ms_per_step"] for t,v in b.items() if det(t))
la=sum(v["ms_per_step"] for t,v in a.items() if not det(t)); lb=sum(v["ms_per_step"] for t,v in b.items() if not det(t))
print("detector group: 32 frames %.3f ms, 96 frames %.3f ms (x%.2f for 3x the frames)" % (sa, sb, sb/sa))
print("landmark group: 256 faces %.3f ms, 768 faces %.3f ms (x%.2f)" % (la, lb, lb/la))
for t in sorted(b, key=lambda t: -b[t]["ms_per_step"]):
    if t in a: print("%-45s %8.4f %8.4f  x%.2f" % (t, a[t]["ms_per_step"], b[t]["ms_per_step"], b[t]["ms_per_step"]/a[t]["ms_per_step"]))
PY
