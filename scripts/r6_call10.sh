#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c10; mkdir -p $O
python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -8 $O/tests_gpu.log; tail -2 $O/smoke.log
