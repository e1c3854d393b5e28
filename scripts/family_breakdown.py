"""per-family and per-shape GEMM milliseconds of the bench step (for scripts/lib_multi.py A/B runs)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "3", "--dump-prof", "/tmp/pl.csv"], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print("ms", d["ms_per_step"], " ".join(f"{f['family'][-14:]}={f['ms_per_step']}" for f in d["roofline"]["families"]))
import csv, collections
rows = list(csv.DictReader(open("/tmp/pl.csv")))
steps = sum(1 for x in rows if x["family"] == "kd_loss")
g = collections.defaultdict(float)
for x in rows:
    if x["family"].startswith("gemm"): g[(x["family"][-8:], x["t0"], x["t1"], x["t2"], x["t3"])] += float(x["ms"]) / steps
for k, v in sorted(g.items(), key=lambda kv: -kv[1])[:14]: print("  ", k, round(v, 3))
