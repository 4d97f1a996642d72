#!/bin/bash
# usage: tools/ab_kernels.sh VAR=a VAR=b -- per-kernel averages (rocprofv3 --kernel-trace, tools/rocpd_stats.py) of the training step under two
# settings of one environment switch, on one box: which kernels paid for / gained from the switch.  Writes gpurun_out/abk_<setting>.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  rm -rf /tmp/abk_$kv
  env $kv rocprofv3 --kernel-trace -d /tmp/abk_$kv -o k -- python $R/bench.py --no-cpu-baseline --no-inference --steps 10 --warmup 5 > /tmp/abk.log 2>&1
  DB=$(find /tmp/abk_$kv -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $DB > $R/gpurun_out/abk_$kv.txt 2>&1
done
