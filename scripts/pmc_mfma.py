"""MFMA utilisation per kernel family from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts MFMA-pipe cycles summed over the SIMDs (16 per 16x16x32 bf16,
32 per 32x32x16); GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the dispatch's active cycles, so
utilisation = busy / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs) and effective clock = GUI_ACTIVE / 8 / duration."""
import csv, glob, json, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
busy, gui, n = collections.Counter(), collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    fam = ("gemm" if "gemm_" in k else "attn_fwd" if ("attn_q_kernel<0" in k or "xattn_fwd" in k) else "attn_bwd" if "attn_" in k else None)
    if fam is None: continue
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[fam] += float(r["Counter_Value"]); n[fam] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": gui[fam] += float(r["Counter_Value"])
out = {}
for fam in busy:
    cyc = gui[fam] / 8.0
    out[fam] = {"launches": n[fam], "mfma_busy_cycles": busy[fam], "kernel_cycles": cyc,
                "mfma_util": round(busy[fam] / (cyc * 256 * 4), 4) if cyc else None}
print(json.dumps(out, indent=1))
