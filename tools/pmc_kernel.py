"""Collects SQ counters for kernels whose name contains a substring (one rocprofv3 pass per counter group).
usage: python tools/pmc_kernel.py <name-substring> [bench args...]   (run on the GPU box, writes gpurun_out/pmc_*.json)"""
import csv, glob, json, os, subprocess, sys, collections
sub = sys.argv[1]
args = sys.argv[2:]
counters = None
out_name = None
while args and (args[0].startswith("--counters=") or args[0].startswith("--out=")):
    if args[0].startswith("--out="):                 # gpurun_out/<name>.json instead of pmc_<substring>.json
        out_name = args.pop(0).split("=", 1)[1]
    else:                                            # one rocprofv3 pass per listed counter (e.g. FETCH_SIZE,WRITE_SIZE)
        counters = args.pop(0).split("=", 1)[1].split(",")
bench_args = args or ["--workload", "landmark", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--lanes", "1"]
GROUPS = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS"],
          ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA", "SQ_INSTS_SMEM"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"],
          ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_F16", "GRBM_GUI_ACTIVE"]]
if counters:
    GROUPS = [[c] for c in counters]
os.environ["TMPDIR"] = "/tmp"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for gi, grp in enumerate(GROUPS):
    d = f"/tmp/pmc_{gi}"
    cmd = ["rocprofv3", "--pmc", *grp, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath("bench.py"), *bench_args]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.getcwd())
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("group", gi, "no output", r.stderr[-400:])
        continue
    for fn in files:
        for row in csv.DictReader(open(fn)):
            if sub in row["Kernel_Name"]:
                res[row["Kernel_Name"][:120]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: max(v) for c, v in cs.items()} for k, cs in res.items()} if counters else \
      {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}   # --counters: largest launch (full batch)
for k, cs in out.items():
    print(k)
    for c, v in cs.items():
        print(f"   {c:24s} {v:16.0f}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/%s.json" % (out_name or "pmc_" + sub.replace('<', '_').replace(',', '_').replace(' ', '')[:40]), "w"), indent=1)
