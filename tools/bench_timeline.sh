#!/bin/bash
# usage: tools/bench_timeline.sh <tag> [bench args]  -- kernel timeline of one training step into gpurun_out/<tag>_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace -d /tmp/tl_$tag -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-inference "$@" > /tmp/tl_$tag.log 2>&1
python3 $R/tools/rocpd_timeline.py $(find /tmp/tl_$tag -name "*.db" | head -1) > $R/gpurun_out/${tag}_timeline.txt 2>&1
