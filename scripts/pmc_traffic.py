"""Average HBM traffic per launch of the GEMM kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream
-> doubled; WRITE_SIZE is exact; both are in KiB."""
import csv, glob, json, sys
def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    rows = list(csv.DictReader(open(f[0])))
    vals = {}
    for r in rows:
        if r["Counter_Name"] != counter: continue
        vals.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return vals
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in ("gemm_lcp_kernel", "gemm_lc_kernel"):
    fk = [v for k, vs in fetch.items() if fam in k for v in vs]
    wk = [v for k, vs in write.items() if fam in k for v in vs]
    if fk:
        out[fam] = {"launches": len(fk), "fetch_MB_per_launch_corrected": round(2 * sum(fk) / len(fk) / 1024, 2),
                    "write_MB_per_launch": round(sum(wk) / max(len(wk), 1) / 1024, 2)}
allf = [v for k, vs in fetch.items() if "gemm_" in k for v in vs]
allw = [v for k, vs in write.items() if "gemm_" in k for v in vs]
out["gemm_family"] = {"launches": len(allf), "traffic_MB_per_launch": round((2 * sum(allf) / len(allf) + sum(allw) / len(allw)) / 1024, 2)}
print(json.dumps(out, indent=1))
