#!/bin/bash
# usage: tools/collect_r05.sh -- everything profiles/r05_* cites, on one box in one call (tools/collect_profiles.sh for both types,
# the low-precision error summary, the band-kernel stage stamps, the bench line)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
python tools/gpu_lowprec_err.py gpurun_out/r05_lowprec_errors.json > gpurun_out/r05_lowprec_errors.txt 2>&1
tools/collect_profiles.sh r05_f16 f16 > /dev/null 2>&1
tools/collect_profiles.sh r05_bf16 bf16 > /dev/null 2>&1
cd $R
python tools/mfma_busy_summary.py gpurun_out/r05_f16_mfma_busy.txt gpurun_out/r05_bf16_mfma_busy.txt gpurun_out/r05_f16_kernel_stats.txt gpurun_out/r05_bf16_kernel_stats.txt > gpurun_out/r05_mfma_busy_summary.txt 2>&1
cp gpurun_out/r05_f16_pmc_traffic.json gpurun_out/r05_pmc_traffic.json
if false; then
  ( echo "# tools/band_lab.hip built with -DBAND_TS: shader-clock stamps around the stage barrier of conv3x3_band_kernel (bf16, batch 64)"; for s in 0 2 6; do tools/labbin/lab_ts 0 5 $s 2>&1 | grep -E "GFLOP|mean of"; done ) > gpurun_out/r05_band_stage_stamps.txt 2>&1
fi
cp gpurun_out/r05_pmc_traffic.json profiles/r05_pmc_traffic.json     # (the bench line quotes the traffic of these very sources)
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_err.txt
tail -c 1500 gpurun_out/r05_bench_line.json
