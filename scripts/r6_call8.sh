#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c8; mkdir -p $O
python scripts/energy_rank.py 2 > $O/energy_rank.log 2>&1
grep -v amdgpu.ids $O/energy_rank.log
