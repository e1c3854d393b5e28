for i in 1 2; do
  PEA_BWD_SIDE=0 timeout 240 python scripts/step_time.py >> gpurun_out/r05_ab_bside.log 2>&1
  timeout 240 python scripts/step_time.py >> gpurun_out/r05_ab_bside.log 2>&1
done
grep -v amdgpu gpurun_out/r05_ab_bside.log
python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_gputests_3.log 2>&1; tail -4 gpurun_out/r05_gputests_3.log
