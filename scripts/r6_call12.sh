#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c12; mkdir -p $O
bash scripts/side_configs.sh > $O/side_configs.txt 2>&1
MASTER_PORT=29533 python bench.py --gpus 1 --force-collective --batch 8 --no-cpu-baseline --steps 8 --warmup 2 > $O/bench_b8_collective.json 2> $O/bench_b8_collective.err
cat $O/side_configs.txt | grep -v amdgpu; tail -c 1500 $O/bench_b8_collective.json; tail -3 $O/bench_b8_collective.err
