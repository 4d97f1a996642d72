#!/bin/bash
# usage: tools/h2u_pmc.sh  -- rocprofv3 --pmc passes (issue / LDS+VMEM / texture+L2) over one training step, the heads-backward kernels' rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/h2u_pmc.txt; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  for up in 1 0; do
    i=$((i+1))
    rm -rf /tmp/hp_$i
    DBX_HEAD2_UP=$up rocprofv3 --pmc $set -d /tmp/hp_$i -o p -- python $R/tools/gpu_layer_times.py f16 > /tmp/hp_$i.log 2>&1
    db=$(find /tmp/hp_$i -name "*.db" | head -1)
    echo "=== DBX_HEAD2_UP=$up: $set" >> $out
    if [ -n "$db" ]; then PMC_FILTER='head2_|upsample_bwd' python3 $R/tools/pmc_summary.py $db >> $out 2>&1; else tail -5 /tmp/hp_$i.log >> $out; fi
  done
done
cat $out
