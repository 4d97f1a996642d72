# usage: tools/ws1x1_ablation.sh "<dbg values>"  -- lab build of conv_igemm.hip (tools/build_variant.sh lab conv_igemm.hip -DDBX_LAB) with DBX_WS_DBG bits:
# 1 no periods (epilogue only), 2 no stores, 4 no band DMA, 8 no weight loads
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/r3v
for d in ${1:-0 1 2 4 8 6 10 12 14}; do echo "== DBX_WS_DBG=$d"; DBX_LIB=$R/densebox_amd/csrc/variants/libdensebox_hip_lab.so DBX_WS_DBG=$d python $R/tools/gpu_conv1x1_bench.py f16 2>&1 | grep -v amdgpu; done > $R/gpurun_out/r3v/abl.log 2>&1
