"""Condense a rocprofv3 output directory into a small per-kernel summary (for profiles/).

    python tools/rocprof_summary.py <rocprof_dir> <out.md> [--pmc]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    d, out = sys.argv[1], sys.argv[2]
    lines = []
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    for path in stats:
        lines.append("## kernel stats (%s)\n" % os.path.basename(path))
        lines.append("| kernel | calls | total ms | avg us | min us | max us | % |")
        lines.append("|---|---|---|---|---|---|---|")
        with open(path) as f:
            rows = list(csv.DictReader(f))
        for r in rows[:60]:
            g = lambda *ks: next((r[k] for k in ks if k in r), "")
            tot = float(g("TotalDurationNs", "Total Duration(ns)") or 0)
            lines.append("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (
                short(g("Name", "KernelName")), g("Calls"), tot / 1e6, float(g("AverageNs", "Average(ns)") or 0) / 1e3,
                float(g("MinNs", "Min(ns)") or 0) / 1e3, float(g("MaxNs", "Max(ns)") or 0) / 1e3, g("Percentage")))
        lines.append("")
    ctr = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for path in ctr:
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r.get("Kernel_Name", r.get("KernelName", "?")))
                agg[k][r.get("Counter_Name", "?")] += float(r.get("Counter_Value", 0) or 0)
                cnt[(k, r.get("Counter_Name", "?"))] += 1
        lines.append("## counters (%s): per-kernel SUM over dispatches and per-dispatch mean\n" % os.path.basename(path))
        lines.append("| kernel | counter | dispatches | sum | mean/dispatch |")
        lines.append("|---|---|---|---|---|")
        for k in sorted(agg, key=lambda kk: -sum(agg[kk].values()))[:40]:
            for c, v in agg[k].items():
                n = cnt[(k, c)]
                lines.append("| %s | %s | %d | %.4g | %.4g |" % (k, c, n, v, v / max(n, 1)))
        lines.append("")
    # hero kernel (up2.conv2 = the only kxk launch of the 128x128 tile): per-dispatch numbers
    hero = "conv_gemm_split_kernel<128, 128, 4, 2, 3, 0>" if any("split_kernel<128, 128, 4, 2, 3, 0>" in open(p).read() for p in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)) else "conv_gemm_kernel<float, 128, 128, 2, 2, 3>"
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        durs = []
        with open(path) as f:
            for r in csv.DictReader(f):
                if hero in r.get("Kernel_Name", ""):
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if durs:
            lines.append("## hero kernel %s: %d dispatches, avg %.2f us, min %.2f us, max %.2f us\n" % (
                hero, len(durs), sum(durs) / len(durs), min(durs), max(durs)))
    for path in ctr:
        vals = defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                if hero in r.get("Kernel_Name", ""):
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in vals.items():
            lines.append("## hero kernel %s: %s per dispatch = %s (raw counter units, KB)\n" % (hero, c, ", ".join("%.1f" % x for x in v)))
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", out, "from", len(stats), "stats files and", len(ctr), "counter files")


if __name__ == "__main__":
    main()
