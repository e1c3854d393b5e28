#!/bin/bash
# same-box A/B of the whole step: the round-5 tree (.ab_r05/: commit 8bb53a6, built from its own sources) against this tree,
# alternating processes; prints ms / MHz / W / J per step of each run.
# Staging .ab_r05/ (git-ignored; in the authoring container, before the gpurun call):
#   git worktree add /tmp/r05 8bb53a6 && make -C /tmp/r05/pea_diffusion_amd/csrc -j6
#   mkdir -p .ab_r05/tests/golden .ab_r05/profiles && cp -r /tmp/r05/pea_diffusion_amd /tmp/r05/include /tmp/r05/bench.py .ab_r05/
#   cp /tmp/r05/tests/golden/mlp_sdxl_6M.npz .ab_r05/tests/golden/ && cp /tmp/r05/profiles/r05_pmc_traffic.json .ab_r05/profiles/
#   rm -f .ab_r05/pea_diffusion_amd/csrc/*.o && git worktree remove /tmp/r05 --force
cd $GRAFT_REPO_ROOT
N=${1:-3}
B="--steps 16 --warmup 3 --no-cpu-baseline --no-roofline --no-dead-row-line"
for i in $(seq $N); do
  for t in r05 r06; do
    if [ $t = r05 ]; then d=.ab_r05; else d=.; fi
    python $d/bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; pw=g.get('power_w_mean') or g['power_w_median']; print('$t', d['ms_per_step'], 'ms', g['sclk_mhz_median'], 'MHz', pw, 'W', round(pw*d['ms_per_step_mean']*1e-3,2), 'J/step', d['value'], 'images/s')"
  done
done
