#!/bin/bash
# usage: tools/boundary_probe.sh -- dependent-kernel boundaries by kernel shape, un-profiled (in-kernel stamps) and as rocprofv3 reports them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_boundaries.txt
echo "== un-profiled: in-kernel s_memrealtime stamps (first workgroup in, last workgroup out)" > $O
$R/tools/labbin/probe_boundary >> $O 2>&1
echo >> $O
echo "== the same binary under rocprofv3 --kernel-trace: its own in-kernel stamps (what the profiler's presence does to the boundary)" >> $O
rm -rf /tmp/bd_prof
rocprofv3 --kernel-trace -d /tmp/bd_prof -o k -- $R/tools/labbin/probe_boundary >> $O 2>&1
echo >> $O
echo "== ... and the gaps rocprofv3 itself reports for that run (end of a kernel -> start of the next, median per kernel under test)" >> $O
python3 $R/tools/rocpd_gaps.py $(find /tmp/bd_prof -name "*.db" | head -1) >> $O 2>&1
