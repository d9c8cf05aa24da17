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
for k, cs in res.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    n = len(next(iter(cs.values())))
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0: continue
    simd_cycles = gui * 1024.0                         # 256 CUs x 4 SIMDs
    table[k] = {"launches": n, "gpu_cycles": round(gui), "valu_busy": round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles, 3),
                "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles, 3),
                "lds_busy": round(4.0 * c.get("SQ_ACTIVE_INST_LDS", 0) / (gui * 256.0), 3),
                "valu_per_wave": round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)),
                "salu_per_wave": round(c.get("SQ_INSTS_SALU", 0) / max(c.get("SQ_WAVES", 1), 1)), "waves": round(c.get("SQ_WAVES", 0))}
json.dump({"_meta": {"bench_args": bench_args, "note": "busy = fraction of the launch's GPU cycles (GRBM_GUI_ACTIVE) x 1024 SIMDs; VALU at 4 cycles per instruction"}, "kernels": table},
          open(out, "w"), indent=1)
for k, t in sorted(table.items(), key=lambda kv: -kv[1]["gpu_cycles"] * kv[1]["launches"])[:45]:
    print("%-92s x%-3d %8d cyc  valu %4.0f%%  mfma %4.0f%%  lds %4.0f%%  valu/wave %6d" % (k[:92], t["launches"], t["gpu_cycles"], 100 * t["valu_busy"], 100 * t["mfma_busy"], 100 * t["lds_busy"], t["valu_per_wave"]))
