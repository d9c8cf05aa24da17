"""VALU / matrix-pipe / LDS occupancy of EVERY kernel of a bench run (run on the GPU box): one rocprofv3 --pmc pass per counter
group, averaged over the launches of a kernel.  usage: python tools/pmc_all.py <out.json> [bench args...]"""
import csv, glob, json, os, subprocess, sys, collections
out = sys.argv[1]
bench_args = sys.argv[2:] or ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-probes", "--no-kernel-table", "--lanes", "1", "--frames", "32"]
GROUPS = [["SQ_WAVES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"],
          ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_F16", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD"]]
os.environ["TMPDIR"] = "/tmp"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for gi, grp in enumerate(GROUPS):
    d = f"/tmp/pmc_all_{gi}"
    r = subprocess.run(["rocprofv3", "--pmc", *grp, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath("bench.py"), *bench_args],
                       capture_output=True, text=True)
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("group", gi, "no output", r.stderr[-400:]); continue
    for fn in files:
        for row in csv.DictReader(open(fn)):
            res[row["Kernel_Name"][:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
table = {}
CLK_MHZ = 2400.0          # busy time = busy cycles per SIMD (or per CU for the LDS) / clock; a launch's wall time under the counter passes is
                          # NOT its real duration (GRBM_GUI_ACTIVE includes the profiler's serialisation), so busy TIMES are reported and
                          # compared with the durations of bench.py's kernel table
for k, cs in res.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    n = len(next(iter(cs.values())))
    if k.startswith("void at::") or k.startswith("__amd") or not c.get("SQ_WAVES"):
        continue
    table[k] = {"launches": n, "valu_busy_us": round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / 1024.0 / CLK_MHZ, 1),
                "mfma_busy_us": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / CLK_MHZ, 1),
                "lds_busy_us": round(4.0 * c.get("SQ_ACTIVE_INST_LDS", 0) / 256.0 / CLK_MHZ, 1),
                "valu_per_wave": round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)),
                "salu_per_wave": round(c.get("SQ_INSTS_SALU", 0) / max(c.get("SQ_WAVES", 1), 1)), "waves": round(c.get("SQ_WAVES", 0))}
json.dump({"_meta": {"bench_args": bench_args, "note": "per launch, averaged over the launches of a kernel: busy time of the VALU (4 cycles per instruction) and of the matrix pipe "
                     "averaged over the 1024 SIMDs, of the LDS averaged over the 256 CUs, at 2.4 GHz -- compare with the launch durations of bench.py's kernel table"},
           "kernels": table}, open(out, "w"), indent=1)
print("%-84s %4s %9s %9s %9s %10s %8s" % ("kernel", "n", "VALU us", "MFMA us", "LDS us", "VALU/wave", "waves"))
for k, t in sorted(table.items(), key=lambda kv: -kv[1]["valu_busy_us"] * kv[1]["launches"])[:48]:
    print("%-84s %4d %9.1f %9.1f %9.1f %10d %8d" % (k[:84], t["launches"], t["valu_busy_us"], t["mfma_busy_us"], t["lds_busy_us"], t["valu_per_wave"], t["waves"]))
