#!/bin/bash
# usage: tools/collect_profiles.sh <tag> [dtype]  -- everything the bench line's roofline object cites, into gpurun_out/:
#   <tag>_kernel_stats.txt   rocprofv3 --kernel-trace per-kernel durations of the bench command
#   <tag>_mfma_busy.txt      cycle / MFMA-busy / wait counters per kernel (own --pmc pass)
#   <tag>_pmc_traffic.json   HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (own --pmc passes), stamped with the source hash
#   <tag>_timeline.txt       kernel sequence of one step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; dt=${2:-f16}
mkdir -p $(dirname $R/gpurun_out/$tag)
B="python $R/bench.py --dtype $dt --no-cpu-baseline --no-inference"
rm -rf /tmp/cp_*
rocprofv3 --kernel-trace -d /tmp/cp_kt -o k -- $B --steps 10 --warmup 5 > /tmp/cp_kt.log 2>&1
DB=$(find /tmp/cp_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB > $R/gpurun_out/${tag}_kernel_stats.txt 2>&1
python3 $R/tools/rocpd_timeline.py $DB > $R/gpurun_out/${tag}_timeline.txt 2>&1
grep '"metric"' /tmp/cp_kt.log | cut -c1-400 >> $R/gpurun_out/${tag}_kernel_stats.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/cp_pm -o p -- $B --steps 2 --warmup 1 > /tmp/cp_pm.log 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/cp_pm -name "*.db" | head -1) > $R/gpurun_out/${tag}_mfma_busy.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/cp_f -o f -- $B --steps 2 --warmup 1 > /tmp/cp_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/cp_w -o w -- $B --steps 2 --warmup 1 > /tmp/cp_w.log 2>&1
cd $R && python3 tools/pmc_traffic.py $(find /tmp/cp_f -name "*.db" | head -1) $(find /tmp/cp_w -name "*.db" | head -1) gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt 2>&1
head -12 gpurun_out/${tag}_kernel_stats.txt | cut -c1-60,108-175
