#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c20; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "geglu or gemm" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
for v in 0 1 0 1; do echo "== PEA_GEMM_EPI_PF=$v"; PEA_GEMM_EPI_PF=$v python scripts/geglu_bwd_bench.py; done > $O/geglu_bwd.log 2>&1
for i in 1 2 3; do for v in 0 1; do echo "== PEA_GEMM_EPI_PF=$v"; PEA_GEMM_EPI_PF=$v python scripts/step_time.py; done; done > $O/step_epf.log 2>&1
tail -3 $O/tests.log; grep -v amdgpu $O/geglu_bwd.log; grep -v amdgpu $O/step_epf.log
