#!/bin/bash
# usage: tools/lab_ablate.sh <shape indices, space separated> -- runs every lab binary under tools/labbin/ (ablation / variant builds of
# tools/band_lab.hip) on the given band_lab shapes, default dispatch (variant 0); two passes so that box warm-up does not favour a binary
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for pass in 1 2; do
for s in $1; do
  for b in $(ls $R/tools/labbin/); do
    echo "[$b shape $s pass $pass] $($R/tools/labbin/$b ${2:-0} 20 $s 2>&1 | grep -E 'variant' | sed 's/maxerr.*bad/bad/')"
  done
done
done
