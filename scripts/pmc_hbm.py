"""Per-kernel-family memory-side traffic and rate from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py:
bytes that crossed the L2 <-> fabric boundary (HBM + Infinity Cache) per launch and GB/s over the dispatch duration of the
same pass.  gfx950 correction as in pmc_traffic.py (FETCH_SIZE x2, both in KiB)."""
import csv, glob, json, sys, collections

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])        # value, duration ns, launches
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        a = acc[r["Kernel_Name"].split("(")[0]]
        a[0] += float(r["Counter_Value"]); a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[2] += 1
    return acc

FAM = [("gemm", lambda k: "gemm_lc" in k), ("attention fwd", lambda k: "attn_q_kernel<0" in k),
       ("attention bwd", lambda k: "attn_" in k and "attn_q_kernel<0" not in k),
       ("layernorm", lambda k: "ln_fwd" in k or "ln_bwd" in k), ("groupnorm", lambda k: "gn_" in k),
       ("geglu bwd", lambda k: "geglu_il" in k), ("kd loss", lambda k: "kd_loss" in k),
       ("concat / split / accumulate", lambda k: any(s in k for s in ("concat2", "split2", "accum_kernel", "sumpool")))]
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for name, pred in FAM:
    fb = sum(2 * v[0] * 1024 for k, v in fetch.items() if pred(k))
    wb = sum(v[0] * 1024 for k, v in write.items() if pred(k))
    ns = sum(v[1] for k, v in fetch.items() if pred(k))
    n = sum(v[2] for k, v in fetch.items() if pred(k))
    if n:
        out[name] = {"launches": n, "read_MB_per_launch": round(fb / n / 1e6, 2), "written_MB_per_launch": round(wb / n / 1e6, 2),
                     "GB_per_s": round((fb + wb) / ns, 1), "of_8_TB_per_s": round((fb + wb) / ns / 8000.0, 3)}
print(json.dumps(out, indent=1))
