"""per-kernel averages of every counter in a rocprofv3 --pmc pass directory"""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60] + "|grid" + r.get("Grid_Size", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if "attn" not in k and "gemm" not in k: continue
    print(k, " ".join(f"{c}={acc[k][c] / n[k][c]:.4g}" for c in sorted(acc[k])))
