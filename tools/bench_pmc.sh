#!/bin/bash
# usage: tools/bench_pmc.sh <tag> [env assignments...]  -- cycle / MFMA-busy counters of a short bench run into gpurun_out/<tag>_mfma_busy.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
rm -rf /tmp/pm_$tag
env "$@" rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pm_$tag -o p -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-inference > /tmp/pm_$tag.log 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/pm_$tag -name "*.db" | head -1) > $R/gpurun_out/${tag}_mfma_busy.txt 2>&1
