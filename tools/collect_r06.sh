#!/bin/bash
# usage: tools/collect_r06.sh -- everything profiles/r06_* cites, on one box in one call: kernel stats / timeline / MFMA-busy / PMC traffic for f16 and
# bf16 (tools/collect_profiles.sh), the per-layer error budget of the 16-bit forward, the per-tensor gradient errors of the 16-bit step, the
# dynamic range of its gradient maps, the stage-2 heads backward pieces, the bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
python tools/gpu_layer_error_budget.py gpurun_out/r06_layer_error_budget_f16.json > gpurun_out/r06_layer_error_budget_f16.txt 2>&1
python tools/gpu_layer_error_budget.py gpurun_out/r06_layer_error_budget_bf16.json --dtype bf16 > gpurun_out/r06_layer_error_budget_bf16.txt 2>&1
python tools/gpu_lowprec_err.py gpurun_out/r06_lowprec_errors.json > gpurun_out/r06_lowprec_errors.txt 2>&1
( echo "# per-tensor error of the 16-bit training step (default path: lin_bwd + heads_gen) against the fp32 oracle: tests/test_hip_backward.py::test_training_step_16bit_default_path_vs_oracle_per_tensor, DBX_PRINT_GRAD_ERR=1"; DBX_PRINT_GRAD_ERR=1 python -m pytest tests/test_hip_backward.py -q -s -k 16bit_default 2>&1 | grep -E "rel_l2|worst rel|passed|failed" ) > gpurun_out/r06_grad_errors.txt 2>&1
python tools/gpu_grad_range.py gpurun_out/r06_grad_range.json > gpurun_out/r06_grad_range.txt 2>&1
python tools/gpu_head2_bench.py f16 > gpurun_out/r06_head2_pieces.txt 2>&1
tools/collect_profiles.sh r06_f16 f16 > /dev/null 2>&1
tools/collect_profiles.sh r06_bf16 bf16 > /dev/null 2>&1
cd $R
python tools/mfma_busy_summary.py gpurun_out/r06_f16_mfma_busy.txt gpurun_out/r06_bf16_mfma_busy.txt gpurun_out/r06_f16_kernel_stats.txt gpurun_out/r06_bf16_kernel_stats.txt > gpurun_out/r06_mfma_busy_summary.txt 2>&1
cp gpurun_out/r06_f16_pmc_traffic.json gpurun_out/r06_pmc_traffic.json
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json     # (the bench line quotes the traffic of these very sources)
( for v in 1 0 1 0; do echo "DBX_F16_GUARD=$v $(DBX_F16_GUARD=$v python bench.py --no-cpu-baseline --no-inference --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") ms/step"; done; for v in 1 0 1 0; do echo "DBX_POOL_WGRAD=$v $(DBX_POOL_WGRAD=$v python bench.py --no-cpu-baseline --no-inference --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") ms/step"; done ) > gpurun_out/r06_ab.txt 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_err.txt
tail -c 1500 gpurun_out/r06_bench_line.json
