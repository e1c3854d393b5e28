T=r04
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-dead-row-line > $O/${T}_rocprofv3_bench_line.json 2> /dev/null
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/${T}_rocprofv3_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dead-row-line > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dead-row-line > /dev/null 2>&1
python3 $R/scripts/pmc_traffic.py /tmp/prof_fetch /tmp/prof_write > $O/${T}_pmc_traffic.json 2> $O/pmc_err.txt
python3 $R/scripts/pmc_hbm.py /tmp/prof_fetch /tmp/prof_write > $O/${T}_pmc_hbm.json 2>> $O/pmc_err.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_mfma -o r --output-format csv -- python3 $R/bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dead-row-line > /dev/null 2>&1
python3 $R/scripts/pmc_mfma.py /tmp/prof_mfma > $O/${T}_pmc_mfma.json 2>> $O/pmc_err.txt
cat $O/${T}_pmc_traffic.json $O/${T}_pmc_mfma.json
