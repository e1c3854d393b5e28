#!/bin/bash
# round 6, call 1: the changed tests + smoke + one bench line with the energy fields
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c1
python -m pytest tests/test_dist_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "dist or resume or forward_tiny or checkpoint or repeatable or training_step_vs_oracle" > gpurun_out/r6c1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6c1/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6c1/smoke.log 2>&1
python bench.py --breakdown > gpurun_out/r6c1/bench.json 2> gpurun_out/r6c1/bench.err
tail -3 gpurun_out/r6c1/tests.log; cat gpurun_out/r6c1/smoke.log | tail -2; tail -12 gpurun_out/r6c1/bench.err
