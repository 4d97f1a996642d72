#!/bin/bash
# usage: tools/step_trace.sh <tag> [env...]  -- rocprofv3 kernel trace of a short bench run; per-kernel stats into gpurun_out/<tag>_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
mkdir -p $(dirname $R/gpurun_out/$tag)
rm -rf /tmp/st_kt
env "$@" rocprofv3 --kernel-trace -d /tmp/st_kt -o k -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-inference > /tmp/st_kt.log 2>&1
DB=$(find /tmp/st_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB > $R/gpurun_out/${tag}_kernel_stats.txt 2>&1
python3 $R/tools/rocpd_timeline.py $DB > $R/gpurun_out/${tag}_timeline.txt 2>&1
grep '"metric"' /tmp/st_kt.log | cut -c1-300 >> $R/gpurun_out/${tag}_kernel_stats.txt
