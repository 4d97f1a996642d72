cd $GRAFT_REPO_ROOT
export DBX_LIB=$GRAFT_REPO_ROOT/densebox_amd/csrc/variants/libdensebox_hip_ab.so
for t in 0 1; do echo "== TSYNC=$t"; DBX_P8_TSYNC=$t python tools/gpu_conv_plan_bench.py f16 64 2 2>&1 | grep p8; DBX_P8_TSYNC=$t python tools/gpu_conv1x1_bench.py f16 2>&1 | grep "p8"; done
echo "== whole step"
bash tools/ab_env.sh 3 30 DBX_P8_TSYNC=0 DBX_P8_TSYNC=1 DBX_P8_HEADS=0 2>&1 | tail -4
