#!/bin/bash
# Round-end measurement pass on ONE MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/final_profile.sh'
# Writes everything under gpurun_out/final/; the summaries that are kept go to profiles/ (see profiles/README.md).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json
timeout -k 10 200 python bench.py --dtype fp32 --steps 10 --no_cpu_baseline > $O/bench_fp32.json 2>/dev/null; cut -c1-160 $O/bench_fp32.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/bench_under_rocprof.log 2>&1
grep "^{" $O/bench_under_rocprof.log | cut -c1-160
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o runc -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_prof > $O/pmc_fetch.log 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o runc -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_prof > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 3 > $O/hbm_traffic_pmc.json
rm -rf $O/pmc_fetch $O/pmc_write
find $O/prof -name "*kernel_trace.csv" -delete
# two ranks sharing the one GPU over gloo: rehearsal of the --gpus N launch path (RCCL needs one device per rank)
PAA_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --seconds 2 --label_tokens 30 --no_cpu_baseline > $O/bench_2rank_gloo.log 2>&1
grep "^{" $O/bench_2rank_gloo.log | cut -c1-200
echo FINAL_OK
