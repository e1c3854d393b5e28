#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c9; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
bash scripts/ab_vs_r05.sh 3 > $O/ab_vs_r05.log 2>&1
tail -6 $O/tests_gpu.log; cat $O/ab_vs_r05.log
