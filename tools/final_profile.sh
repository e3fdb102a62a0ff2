#!/bin/bash
# (the profiled runs pass --eager: same kernels as the default hipGraph replay, but exactly warm-up + timed + 1 steps per process,
# which is what the per-step columns of the summaries divide by)
# Round-end measurement pass on ONE MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/final_profile.sh'
# Writes everything under gpurun_out/final/; the summaries that are kept go to profiles/ (see profiles/README.md).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python bench.py --steps 64 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-200 $O/bench.json
cd /tmp && export TMPDIR=/tmp
for DT in fp32 bf16; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$DT -o runc -- python3 $R/bench.py --dtype $DT --steps 3 --warmup 1 --eager --no_cpu_baseline --no_fft_bench --no_pmc > $O/bench_under_rocprof_$DT.log 2>&1 || { echo "rocprof $DT failed"; tail -5 $O/bench_under_rocprof_$DT.log; }
  find $O/prof_$DT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${DT}_kernel_stats.csv
  find $O/prof_$DT -name "*kernel_trace.csv" -delete
  grep "^{" $O/bench_under_rocprof_$DT.log | cut -c1-160
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${DT}_$C -o runc -- python3 $R/bench.py --dtype $DT --steps 2 --warmup 1 --eager --no_cpu_baseline --no_fft_bench --no_prof --no_pmc > $O/pmc_${DT}_$C.log 2>&1 || echo "pmc $DT $C failed"
  done
  (cd $R && python tools/pmc_summary.py $O/pmc_${DT}_FETCH_SIZE $O/pmc_${DT}_WRITE_SIZE 4 $DT $O/bench.json > $O/hbm_traffic_pmc_$DT.json) || echo "pmc summary $DT failed"
  rm -rf $O/pmc_${DT}_FETCH_SIZE $O/pmc_${DT}_WRITE_SIZE
done
cd $R
# headline line again with the PMC traffic of this commit filled in
timeout -k 10 300 python bench.py --dtype fp32 --steps 64 --warmup 5 --no_cpu_baseline --no_fft_bench --pmc_json $O/hbm_traffic_pmc_fp32.json > $O/bench_fp32_with_traffic.json 2>/dev/null; cut -c1-120 $O/bench_fp32_with_traffic.json
# two ranks sharing the one GPU over gloo through bench.py's own launcher (RCCL needs one device per rank)
PAA_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --seconds 2 --label_tokens 30 --no_cpu_baseline --no_fft_bench > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; cut -c1-200 $O/bench_2rank_gloo.json
bash tools/fft_profile.sh > $O/fft_profile.log 2>&1; tail -4 $O/fft_profile.log
echo FINAL_OK
