#!/bin/bash
# usage: tools/kernel_pmc.sh <tag> <command...>  -- two rocprofv3 --pmc passes (issue/wait/MFMA-busy counters; LDS counters) plus a kernel
# trace over the command, per-kernel averages into gpurun_out/<tag>_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
out=$R/gpurun_out/${tag}_pmc.txt; : > $out
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/kp_$i
  rocprofv3 --pmc $set -d /tmp/kp_$i -o p -- "$@" > /tmp/kp_$i.log 2>&1
  db=$(find /tmp/kp_$i -name "*.db" | head -1)
  echo "=== pass $i: $set" >> $out
  if [ -n "$db" ]; then python3 $R/tools/pmc_summary.py $db >> $out 2>&1; else tail -5 /tmp/kp_$i.log >> $out; fi
done
rm -rf /tmp/kp_t
rocprofv3 --kernel-trace -d /tmp/kp_t -o k -- "$@" > /tmp/kp_t.log 2>&1
echo "=== kernel trace" >> $out
python3 $R/tools/rocpd_stats.py $(find /tmp/kp_t -name "*.db" | head -1) >> $out 2>&1
