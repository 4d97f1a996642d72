cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/r3r
rm -rf /tmp/it
rocprofv3 --kernel-trace -d /tmp/it -o k -- python $R/tools/gpu_infer_trace.py 512 512 f16 > /tmp/it.log 2>&1
DB=$(find /tmp/it -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB > $R/gpurun_out/r3r/infer512_timeline.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB > $R/gpurun_out/r3r/infer512_stats.txt 2>&1
tail -3 /tmp/it.log
