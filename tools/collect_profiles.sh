#!/bin/bash
# Copies the summaries of the last tools/final_profile.sh run (gpurun_out/final, gpurun_out/fft) into profiles/<round>_*.
#   bash tools/collect_profiles.sh r2
set -e
P=${1:?round prefix, e.g. r2}
F=gpurun_out/final
cp $F/bench.json profiles/${P}_final_bench_b32.json
cp $F/fp32_kernel_stats.csv profiles/${P}_final_bench_b32_fp32parity_kernel_stats.csv
cp $F/bf16_kernel_stats.csv profiles/${P}_final_bench_b32_bf16_kernel_stats.csv
grep "^{" $F/bench_under_rocprof_fp32.log > profiles/${P}_final_bench_b32_fp32parity_under_rocprof.json
grep "^{" $F/bench_under_rocprof_bf16.log > profiles/${P}_final_bench_b32_bf16_under_rocprof.json
cp $F/hbm_traffic_pmc_fp32.json profiles/${P}_hbm_traffic_pmc_fp32parity.json
cp $F/hbm_traffic_pmc_bf16.json profiles/${P}_hbm_traffic_pmc_bf16.json
cp $F/bench_fp32_with_traffic.json profiles/${P}_final_bench_b32_fp32parity_with_traffic.json
cp $F/bench_2rank_gloo.json profiles/${P}_final_bench_2rank_gloo_rehearsal.json
cp gpurun_out/fft/proj_bench.json profiles/${P}_projection_path.json
cp gpurun_out/fft/fft_kernel_stats.csv profiles/${P}_fft_path_kernel_stats.csv
cp gpurun_out/fft/fft_hbm_pmc.json profiles/${P}_fft_path_hbm_pmc.json
tail -1 $F/pytest_gpu.log
