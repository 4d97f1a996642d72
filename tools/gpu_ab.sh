#!/bin/bash
# usage: tools/gpu_ab.sh <rounds> <cmd...> -- interleaved same-box A/B: runs the command alternately with every library under
# densebox_amd/csrc/variants/ (DBX_LIB) and with the product build, <rounds> times; prints each run's output tagged with the variant
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in product $(ls $R/densebox_amd/csrc/variants/*.so 2>/dev/null); do
    tag=$(basename $lib .so | sed 's/libdensebox_hip_//')
    if [ "$lib" == "product" ]; then "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/[product] /"; else DBX_LIB=$lib "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/[$tag] /"; fi
  done
done
