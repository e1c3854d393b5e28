"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (single-stream run): how much of the
step is launch bubbles, and after which kernels."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows))
# keep the last 40 % of the trace (timed steps, not init)
ks = ks[int(len(ks) * 0.6):]
busy = sum(e - s for s, e, _ in ks)
span = ks[-1][1] - ks[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = 0
hist = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g = s1 - e0
    if g > 200000: continue          # step boundaries / host syncs
    if g < 0: g = 0
    tot_gap += g
    gaps[n0 + " -> " + n1][0] += g; gaps[n0 + " -> " + n1][1] += 1
    hist[min(g // 1000, 20)] += 1
print(f"kernels {len(ks)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  gaps(<200us) {tot_gap/1e6:.2f} ms  avg gap {tot_gap/len(ks)/1e3:.2f} us")
print("gap histogram (us: count):", sorted(hist.items()))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g/1e6:7.3f} ms  {n:5d} x {g/n/1e3:6.2f} us  {k}")
