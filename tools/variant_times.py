"""Times the per-tag kernels of one bench step for each library under variants/ (kernel experiments)."""
import glob, json, os, subprocess, sys
tagsub = sys.argv[1] if len(sys.argv) > 1 else "mbconv"
for lib in sorted(glob.glob("variants/lib_*.so")):
    env = dict(os.environ, PEPPA_HIP_LIBRARY=os.path.abspath(lib))
    out = f"gpurun_out/prof_{os.path.basename(lib)}.json"
    r = subprocess.run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--dump-profile", out],
                       env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        k = json.load(open(out))["kernels"]
        print(os.path.basename(lib), round(d["value"]), {n: round(v["ms_per_step"], 4) for n, v in k.items() if tagsub in n})
    except Exception as e:
        print(lib, "failed", e, r.stderr[-500:])
