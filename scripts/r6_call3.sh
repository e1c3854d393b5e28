#!/bin/bash
# round 6, call 3: cross-attention backward v2, attn_delta rows, GroupNorm split; parity first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c3; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn or groupnorm" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python scripts/xattn_time.py > $O/xattn.log 2>&1
python scripts/attn_bwd_time.py > $O/attn.log 2>&1
python scripts/gn_bench.py > $O/gn.log 2>&1
for v in 0 1 0 1; do echo "== PEA_XATTN_BWD_V1=$v"; if [ $v = 1 ]; then PEA_XATTN_BWD_V1=1 python scripts/step_time.py; else python scripts/step_time.py; fi; done > $O/step_xattn.log 2>&1
tail -5 $O/tests.log; cat $O/xattn.log $O/attn.log $O/gn.log $O/step_xattn.log | grep -v "amdgpu.ids"
