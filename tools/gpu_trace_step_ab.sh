#!/bin/bash
# usage: tools/gpu_trace_step_ab.sh <kernel-name regex> -- average durations (rocprofv3 kernel trace over a few training steps,
# tools/gpu_quick_bench.py) of the matching kernels for the product library and every variant under densebox_amd/csrc/variants/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for lib in product $(ls $R/densebox_amd/csrc/variants/*.so 2>/dev/null); do
  tag=$(basename $lib .so | sed 's/libdensebox_hip_//')
  rm -rf /tmp/tsab; if [ "$lib" == "product" ]; then unset DBX_LIB; else export DBX_LIB=$lib; fi
  rocprofv3 --kernel-trace -d /tmp/tsab -o k -- python $R/bench.py --no-cpu-baseline --no-inference --steps 5 --warmup 2 > /tmp/tsab.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/tsab -name "*.db" | head -1) 2>&1 | grep -E "$1" | awk -v t="[$tag]" '{printf "%-12s %-70s calls %4s avg %9s us\n", t, substr($1,1,70), $2, $4}'
done
