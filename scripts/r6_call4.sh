#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c4; mkdir -p $O
python scripts/k640_bench.py > $O/k640.log 2>&1
python scripts/xattn_time.py > $O/xattn.log 2>&1
for hf in 0 1 0 1 0 1; do echo "== PEA_ATTN_BWD_HEAVY_FIRST=$hf"; PEA_ATTN_BWD_HEAVY_FIRST=$hf python scripts/step_time.py; done > $O/step_heavy_first.log 2>&1
cat $O/k640.log $O/xattn.log $O/step_heavy_first.log | grep -v "amdgpu.ids"
