#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c6; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > $O/tests_ops.log 2>&1; echo "rc=$?" >> $O/tests_ops.log
python scripts/xattn_time.py > $O/xattn.log 2>&1
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full_model_step_vs_oracle_512 or merged_passes or training_step_vs_oracle" > $O/tests_model.log 2>&1; echo "rc=$?" >> $O/tests_model.log
for v in 3 2 3 2; do echo "== PEA_XATTN_BWD_VER=$v"; PEA_XATTN_BWD_VER=$v python scripts/step_time.py; done > $O/step_ver.log 2>&1
tail -4 $O/tests_ops.log; cat $O/xattn.log;  tail -4 $O/tests_model.log; cat $O/step_ver.log | grep -v "amdgpu.ids"
