#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c5; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > $O/tests_ops.log 2>&1; echo "rc=$?" >> $O/tests_ops.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full_model_step_vs_oracle_512 or merged_passes or full_size_properties or training_step_vs_oracle" > $O/tests_model.log 2>&1; echo "rc=$?" >> $O/tests_model.log
for v in 0 1 0 1; do echo "== PEA_XATTN_NO_DEFER=$v"; if [ $v = 1 ]; then PEA_XATTN_NO_DEFER=1 python scripts/step_time.py; else python scripts/step_time.py; fi; done > $O/step_defer.log 2>&1
tail -4 $O/tests_ops.log; tail -6 $O/tests_model.log; cat $O/step_defer.log | grep -v "amdgpu.ids"
