"""What bounds a multi-lane step?  Reduce a rocprofv3 --kernel-trace of `bench.py --lanes N` to
  * the fraction of wall time with 0 / 1 / 2 / 3+ kernels in flight (steady-state window only),
  * per kernel: launches, mean duration under contention vs. mean duration in a one-lane trace,
  * CU fill per launch: workgroups in the grid / workgroups the chip can hold at once (256 CUs x the
    per-CU residency the launch's VGPR / LDS / wave footprint allows).

    python tools/lane_trace.py <rocprof_dir_multi_lane> [<rocprof_dir_one_lane>] <out.md>

The steady-state window is the middle 60 % of the dispatches of the most expensive kernel (warm-up,
graph capture and the eager re-check at the end fall outside it)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

N_CU = 256


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    return name[:96]


def load(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append(r)
    out = []
    for r in rows:
        g = lambda k, dflt=0: int(float(r.get(k, dflt) or dflt))
        wg = max(1, g("Workgroup_Size_X", 1) * g("Workgroup_Size_Y", 1) * g("Workgroup_Size_Z", 1))
        grid = max(1, g("Grid_Size_X", 1) * g("Grid_Size_Y", 1) * g("Grid_Size_Z", 1))
        out.append({"name": short(r["Kernel_Name"]), "t0": g("Start_Timestamp"), "t1": g("End_Timestamp"), "queue": r.get("Queue_Id", "?"),
                    "wg": wg, "n_wg": (grid + wg - 1) // wg, "lds": g("LDS_Block_Size"), "vgpr": g("VGPR_Count") + g("Accum_VGPR_Count"),
                    "sgpr": g("SGPR_Count")})
    out.sort(key=lambda r: r["t0"])
    return out


def residency(r):
    """Workgroups of this launch one CU can hold (MI355X_MICROARCH.md: 512 registers per lane and SIMD in steps of 8, 160 KiB LDS,
    32 waves per CU)."""
    waves = (r["wg"] + 63) // 64
    alloc = max(8, (r["vgpr"] + 7) // 8 * 8)
    wps = min(8, 512 // alloc) if alloc <= 512 else 1
    by_reg = max(1, (4 * wps) // waves)
    by_wave = max(1, 32 // waves)
    by_lds = (160 * 1024) // r["lds"] if r["lds"] > 0 else 64
    return max(1, min(by_reg, by_wave, by_lds, 16))


def window(rows):
    tot = defaultdict(int)
    for r in rows:
        tot[r["name"]] += r["t1"] - r["t0"]
    hero = max(tot, key=tot.get)
    h = [r for r in rows if r["name"] == hero]
    lo, hi = int(len(h) * 0.2), max(int(len(h) * 0.8), int(len(h) * 0.2) + 1)
    return hero, h[lo]["t0"], h[min(hi, len(h) - 1)]["t1"]


def stats(rows, w0, w1):
    per = defaultdict(lambda: [0, 0.0, 0, 0.0])
    for r in rows:
        if r["t0"] >= w0 and r["t1"] <= w1:
            p = per[r["name"]]
            p[0] += 1
            p[1] += (r["t1"] - r["t0"]) / 1e3
            p[2] = r["n_wg"]
            p[3] = r["n_wg"] / float(N_CU * residency(r))
    return per


def main():
    args = sys.argv[1:]
    multi, out = args[0], args[-1]
    single = args[1] if len(args) == 3 else None
    rows = load(multi)
    hero, w0, w1 = window(rows)
    inw = [r for r in rows if r["t1"] > w0 and r["t0"] < w1]
    ev = []
    for r in inw:
        ev.append((max(r["t0"], w0), 1))
        ev.append((min(r["t1"], w1), -1))
    ev.sort()
    depth, last, hist = 0, w0, defaultdict(int)
    for t, dlt in ev:
        hist[min(depth, 4)] += t - last
        last = t
        depth += dlt
    hist[min(depth, 4)] += w1 - last
    wall = float(w1 - w0)
    busy = sum((min(r["t1"], w1) - max(r["t0"], w0)) for r in inw)
    per = stats(rows, w0, w1)
    alone = {}
    if single:
        srows = load(single)
        _, s0, s1 = window(srows)
        alone = stats(srows, s0, s1)
    L = ["# multi-lane kernel trace, steady-state window of %.1f ms (%d dispatches, %d queues)\n" % (
        wall / 1e6, len(inw), len({r["queue"] for r in inw})),
         "window = middle 60 %% of the dispatches of `%s`\n" % hero,
         "| kernels in flight | share of wall time |", "|---|---|"]
    for k in range(5):
        L.append("| %s | %.1f %% |" % (str(k) if k < 4 else "4+", 100.0 * hist[k] / wall))
    L.append("\nsum of kernel durations / wall = %.2f (mean number of kernels in flight)\n" % (busy / wall))
    L.append("| kernel | launches | mean us (lanes) | mean us (alone) | stretch | workgroups | CU fill (grid / resident capacity) | share of summed kernel time |")
    L.append("|---|---|---|---|---|---|---|---|")
    tot = sum(p[1] for p in per.values())
    for name, p in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
        m = p[1] / p[0]
        a = alone.get(name)
        am = a[1] / a[0] if a and a[0] else None
        L.append("| %s | %d | %.1f | %s | %s | %d | %.2f | %.1f %% |" % (
            name, p[0], m, ("%.1f" % am) if am else "-", ("%.2f" % (m / am)) if am else "-", p[2], p[3], 100.0 * p[1] / tot))
    if alone:
        both = [(n, p) for n, p in per.items() if n in alone and alone[n][0]]
        s_l = sum(p[1] / p[0] * alone[n][0] for n, p in both)     # contention time for one lane-step's worth of launches
        s_a = sum(alone[n][1] for n, p in both)
        L.append("\nall kernels, one-lane launch mix: %.2f ms alone -> %.2f ms under contention (x%.2f)\n" % (s_a / 1e3, s_l / 1e3, s_l / max(s_a, 1e-9)))
        small = [(n, p) for n, p in both if p[3] < 0.5]
        L.append("launches filling < 50 %% of the chip: %d kernels, %.1f %% of the summed kernel time under contention\n" % (
            len(small), 100.0 * sum(p[1] for _, p in small) / tot))
    with open(out, "w") as f:
        f.write("\n".join(L) + "\n")
    print("\n".join(L[:14]))


if __name__ == "__main__":
    main()
