#!/bin/bash
# PMC decomposition of the cross-attention kernels (run through gpurun from the repo root); counters in their own passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -f $O/r06_xattn_pmc.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/xpmc$i -o r --output-format csv -- python3 $R/scripts/xattn_pmc_probe.py > /dev/null 2> $O/xattn_pmc_err$i.txt
  python3 $R/scripts/pmc_generic.py /tmp/xpmc$i >> $O/r06_xattn_pmc.txt 2>> $O/xattn_pmc_err$i.txt
done
cat $O/r06_xattn_pmc.txt
