#!/bin/bash
# usage: tools/gpu_trace_ab.sh <layers,comma> -- per-kernel average durations (rocprofv3 kernel trace) of tools/gpu_wgrad_bench.py for the
# product library and every variant under densebox_amd/csrc/variants/, one layer per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for layer in $(echo $1 | tr ',' ' '); do
  for lib in product $(ls $R/densebox_amd/csrc/variants/*.so 2>/dev/null); do
    tag=$(basename $lib .so | sed 's/libdensebox_hip_//')
    rm -rf /tmp/tab; if [ "$lib" == "product" ]; then unset DBX_LIB; else export DBX_LIB=$lib; fi
    rocprofv3 --kernel-trace -d /tmp/tab -o k -- python $R/tools/gpu_wgrad_bench.py f16 10 $layer > /tmp/tab.log 2>&1
    python3 $R/tools/rocpd_stats.py $(find /tmp/tab -name "*.db" | head -1) 2>&1 | grep "wgrad" | awk -v t="[$layer $tag]" '{printf "%-22s %-60s calls %4s avg %9s us\n", t, substr($1,1,60), $2, $4}'
  done
done
