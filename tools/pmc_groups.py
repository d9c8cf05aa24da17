"""FETCH_SIZE / WRITE_SIZE of the kernels behind the Student's stage 3-5 blocks -> profiles-style JSON keyed by bench.py's profile tags.
usage (GPU box): python tools/pmc_groups.py <out.json> [bench args for the mbx policy, e.g. --mbx recompute]
Runs tools/pmc_kernel.py twice (one rocprofv3 --pmc pass per counter) on the landmark workload (256 faces, one lane)."""
import json, os, re, subprocess, sys
out, extra = sys.argv[1], sys.argv[2:]
tmp = "pmc_groups_tmp"
cmd = [sys.executable, "tools/pmc_kernel.py", "_kernel", "--counters=FETCH_SIZE,WRITE_SIZE", "--out=" + tmp,
       "--workload", "landmark", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-table"] + extra
subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
raw = json.load(open("gpurun_out/%s.json" % tmp))
CEXP = {(3, 5): None}      # (KS, NTO) does not determine the expanded width: read it from the shapes below
SHAPES = {  # (KS, NTO, K, DIL) -> (cin, cexp, cout) of the Student's blocks at 256 x 256
    (3, 5, 3, 1): (80, 200, 80), (3, 7, 3, 1): (80, 480, 112), (4, 7, 3, 1): (112, 672, 112), (4, 10, 5, 1): (112, 672, 160), (5, 10, 5, 2): (160, 960, 160)}
tags = {}
for name, rec in raw.items():
    m = re.match(r"void mbx_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", name) or re.match(r"mbx_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        nw, ks, nto, k, dil, mode = map(int, m.groups())
        cin, cexp, cout = SHAPES[(ks, nto, k, dil)]
        tag = "mbx%s%dx%dd%d_c%d_m%d_n%d_16x16" % (["", "A", "B", "S"][mode], k, k, dil, cin, cexp, cout if mode in (0, 2) else 0)
        # the block's algorithmic bytes: its input once + its output once (the residual is the input); a launch that is half of a
        # block (squeeze pass, or its partner) is priced with the WHOLE block's bytes so that the ratios of a block's launches add up
        tags[tag] = dict(rec, kernel=name, waves=nw, faces_per_launch=256, algorithmic_bytes_per_face=(cin + cout) * 256 * 4)
        continue
    m = re.search(r"conv_gemm_split_kernel<(\d+), (\d+), \d+, \d+, 1, 0, 0, 1, 0, (\d+)>", name)
    if m and int(m.group(3)) in (15, 21, 30):
        n, kdepth = int(m.group(2)), int(m.group(3)) * 32
        tag = "conv1x1_c%d_n%d_16x16" % (kdepth, n)
        tags[tag] = dict(rec, kernel=name, faces_per_launch=256, algorithmic_bytes_per_face=(kdepth + 2 * n) * 256 * 4)   # the map in, the output out, the residual in
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump({"_meta": {"tool": "tools/pmc_groups.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each (KB); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE on gfx950 "
                             "(MI355X_MICROARCH.md, HBM); largest launch of each kernel, 256 faces", "bench_args": extra}, "tags": tags}, open(out, "w"), indent=1)
for t, r in sorted(tags.items()):
    tr = (2 * r.get("FETCH_SIZE", 0) + r.get("WRITE_SIZE", 0)) * 1024 / 256
    print("%-40s traffic %.2f MB/face  (block algorithmic %.2f MB/face)" % (t, tr / 1e6, r["algorithmic_bytes_per_face"] / 1e6))
