#!/bin/bash
# usage: tools/bench_prof.sh <tag>  -- rocprofv3 kernel trace + cycle/MFMA counters of a short bench run; summaries into gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1
rm -rf /tmp/kt_$tag /tmp/pm_$tag
rocprofv3 --kernel-trace -d /tmp/kt_$tag -o k -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-inference > /tmp/kt_$tag.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/kt_$tag -name "*.db" | head -1) > $R/gpurun_out/${tag}_kernel_stats.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pm_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference > /tmp/pm_$tag.log 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/pm_$tag -name "*.db" | head -1) > $R/gpurun_out/${tag}_mfma_busy.txt 2>&1
head -40 $R/gpurun_out/${tag}_kernel_stats.txt | cut -c1-70,108-190
