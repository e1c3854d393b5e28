#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root): bench line with CPU baseline and
# per-family breakdown, rocprofv3 kernel stats of the same command, two PMC passes (FETCH_SIZE / WRITE_SIZE), the MFMA
# utilisation pass, the per-launch profile and the GEMM / attention variant tables.  Everything lands in gpurun_out/ with the
# round tag ($1, default r03); copy what should be judged into profiles/.
T=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --breakdown --dump-prof $O/${T}_per_launch_profile.csv > $O/${T}_bench_line.json 2> $O/${T}_bench_breakdown.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-dead-row-line > $O/${T}_rocprofv3_bench_line.json 2> /dev/null
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/${T}_rocprofv3_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dead-row-line > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dead-row-line > /dev/null 2>&1
python3 $R/scripts/pmc_traffic.py /tmp/prof_fetch /tmp/prof_write > $O/${T}_pmc_traffic.json 2> $O/pmc_err.txt
python3 $R/scripts/pmc_hbm.py /tmp/prof_fetch /tmp/prof_write > $O/${T}_pmc_hbm.json 2>> $O/pmc_err.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_mfma -o r --output-format csv -- python3 $R/bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dead-row-line > /dev/null 2>&1
python3 $R/scripts/pmc_mfma.py /tmp/prof_mfma > $O/${T}_pmc_mfma.json 2>> $O/pmc_err.txt
python3 $R/scripts/gemm_bench.py 24,25,27,28,29 > $O/${T}_gemm_variants.log 2>&1
python3 $R/scripts/gemm_bench.py 24,25,27,28,29 merged > $O/${T}_gemm_variants_merged.log 2>&1
python3 $R/scripts/attn_bench.py > $O/${T}_attn_bench.log 2>&1
python3 $R/scripts/xattn_time.py > $O/${T}_xattn_time.log 2>&1
tail -1 $O/${T}_bench_line.json | cut -c1-300
cat $O/${T}_pmc_traffic.json $O/${T}_pmc_mfma.json
