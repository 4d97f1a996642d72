#!/bin/bash
# usage: tools/lab_pmc.sh <tag> <lab args...>   -- three rocprofv3 --pmc passes over tools/band_lab, summaries into gpurun_out/<tag>_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
out=$R/gpurun_out/${tag}_pmc.txt; : > $out
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set -d /tmp/pmc_$i -o p -- $R/tools/band_lab "$@" > /tmp/pmc_$i.log 2>&1
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "=== pass $i: $set" >> $out
  if [ -n "$db" ]; then python3 $R/tools/pmc_summary.py $db >> $out 2>&1; else tail -5 /tmp/pmc_$i.log >> $out; fi
done
cat $out
