#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c13; mkdir -p $O
for i in 1 2 3 4 5; do for v in 0 1280; do echo "== PEA_GEMM_KSW_MINK=$v"; PEA_GEMM_KSW_MINK=$v python scripts/step_time.py; done; done > $O/step_ksw.log 2>&1
grep -v amdgpu $O/step_ksw.log
