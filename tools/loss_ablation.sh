R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for v in product nomine nofinal; do
  if [ $v == product ]; then bash $R/tools/step_trace.sh r4f/$v; else bash $R/tools/step_trace.sh r4f/$v DBX_LIB=$R/densebox_amd/csrc/variants/libdensebox_hip_$v.so; fi
  echo $v $(grep "loss_kernel" $R/gpurun_out/r4f/${v}_kernel_stats.txt | cut -c100-150)
done
