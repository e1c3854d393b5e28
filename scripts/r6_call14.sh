#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c14; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
for v in 0 1 0 1; do echo "== S4=$v"; if [ $v = 1 ]; then PEA_GEMM_V25_S4=1 python scripts/gemm_fixed_cost.py; else python scripts/gemm_fixed_cost.py; fi; done > $O/fixed_cost.log 2>&1
for i in 1 2 3; do for v in 0 1; do echo "== PEA_GEMM_V25_S4=$v"; if [ $v = 1 ]; then PEA_GEMM_V25_S4=1 python scripts/step_time.py; else python scripts/step_time.py; fi; done; done > $O/step_s4.log 2>&1
tail -3 $O/tests.log; grep -v amdgpu $O/fixed_cost.log | grep -v "host floor"; grep -v amdgpu $O/step_s4.log
