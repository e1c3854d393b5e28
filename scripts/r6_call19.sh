#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c19; mkdir -p $O
python -m pytest tests/test_model_gpu.py tests/test_buckets_gpu.py tests/test_dead_rows_gpu.py -x -q -m gpu -k "tiny or bucket or dead_rows_sdxl_1024_bench or repeatable" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
for v in 1 0 1 0; do echo "== PEA_GEMM_PF=$v"; PEA_GEMM_PF=$v python scripts/step_time.py; done > $O/step_pf.log 2>&1
tail -3 $O/tests.log; grep -v amdgpu $O/step_pf.log
