#!/bin/bash
# PMC decomposition of the self-attention kernels' wave cycles (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+" | sort -u > $O/pmc_counter_names.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/apmc$i -o r --output-format csv -- python3 $R/scripts/attn_pmc_probe.py > /dev/null 2> $O/attn_pmc_err$i.txt
  python3 $R/scripts/pmc_generic.py /tmp/apmc$i >> $O/r03_attn_pmc.txt 2>> $O/attn_pmc_err$i.txt
done
cat $O/r03_attn_pmc.txt
