#!/bin/bash
# round 6, call 2: attention backward -- dK/dV epilogue through LDS (E2), dQ role reads delta (E3), heavy-first order (E1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c2; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python scripts/lib_multi.py old,e2only,e3only scripts/attn_bwd_time.py > $O/attn_variants.log 2>&1
for hf in 0 1 0 1; do echo "== PEA_ATTN_BWD_HEAVY_FIRST=$hf"; PEA_ATTN_BWD_HEAVY_FIRST=$hf python scripts/attn_bwd_time.py; done > $O/attn_heavy_first.log 2>&1
cp pea_diffusion_amd/libpea_hip_old.so pea_diffusion_amd/libpea_hip_alt.so
python scripts/lib_ab.py 2 scripts/step_time.py > $O/step_ab_old.log 2>&1
for hf in 0 1 0 1; do echo "== PEA_ATTN_BWD_HEAVY_FIRST=$hf"; PEA_ATTN_BWD_HEAVY_FIRST=$hf python scripts/step_time.py; done > $O/step_heavy_first.log 2>&1
tail -3 $O/tests.log; cat $O/attn_variants.log $O/attn_heavy_first.log $O/step_ab_old.log $O/step_heavy_first.log | grep -v "^$"
