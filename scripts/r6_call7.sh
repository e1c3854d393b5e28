#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c7; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or ln_" > $O/tests_ops.log 2>&1; echo "rc=$?" >> $O/tests_ops.log
for v in 0 8192 0 8192; do echo "== PEA_LN_BWD_ONE_ROW_BELOW=$v"; PEA_LN_BWD_ONE_ROW_BELOW=$v python scripts/ln_bench.py; done > $O/ln.log 2>&1
for v in 0 8192 0 8192; do echo "== PEA_LN_BWD_ONE_ROW_BELOW=$v"; PEA_LN_BWD_ONE_ROW_BELOW=$v python scripts/step_time.py; done > $O/step_ln.log 2>&1
python bench.py --no-cpu-baseline --breakdown --steps 12 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -3 $O/tests_ops.log; grep -v amdgpu.ids $O/ln.log $O/step_ln.log; grep -v amdgpu.ids $O/bench.err | tail -12
