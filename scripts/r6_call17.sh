#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c17; mkdir -p $O
python bench.py --breakdown > $O/bench_line.json 2> $O/bench_breakdown.txt
tail -c 600 $O/bench_line.json; grep -v amdgpu $O/bench_breakdown.txt
