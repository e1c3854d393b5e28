#!/bin/bash
# bench.py side configurations on ONE box (run through gpurun from the repo root): the default line, per-GPU batch 8
# (BASELINE configs[2] per rank), the SSD-1B student under the SDXL teacher (configs[3]), the 52-token student context,
# the whole training_step with the frozen front end, the ControlNet inference loop, the nine aspect-ratio buckets.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/side
mkdir -p $O
cd $R; ulimit -c 0
python3 bench.py --no-cpu-baseline --no-roofline > $O/bench_default.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-roofline --batch 8 > $O/bench_b8.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-roofline --student ssd1b > $O/bench_ssd1b.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-roofline --ctx 52 > $O/bench_ctx52.json 2>/dev/null
python3 scripts/bucket_bench.py > $O/bucket_bench.log 2>&1
[ -f scripts/full_step_bench.py ] && python3 scripts/full_step_bench.py > $O/full_step_bench.log 2>&1
[ -f scripts/infer_bench.py ] && python3 scripts/infer_bench.py > $O/infer_bench.log 2>&1
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d["value"], "images/s", d["ms_per_step"], "ms |", d.get("passes", ""), "|", d["config"]["workload"][-60:])
PY
grep -v amdgpu $O/bucket_bench.log | tail -2
tail -3 $O/full_step_bench.log 2>/dev/null
tail -3 $O/infer_bench.log 2>/dev/null
