#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c21; mkdir -p $O
python scripts/family_energy.py > $O/family_energy.log 2>&1
grep -v amdgpu $O/family_energy.log
