#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c16; mkdir -p $O
python scripts/lib_multi.py unrollqb scripts/xattn_time.py > $O/xattn_unroll.log 2>&1
cp pea_diffusion_amd/libpea_hip_unrollqb.so pea_diffusion_amd/libpea_hip_alt.so
python scripts/lib_ab.py 3 scripts/step_time.py > $O/step_unroll.log 2>&1
grep -v amdgpu $O/xattn_unroll.log $O/step_unroll.log
