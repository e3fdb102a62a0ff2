set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no_cpu_baseline > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['bf16']['value'],d['roofline']['all_gemm_variants'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -o runc -- python3 $R/bench.py --dtype fp32 --steps 3 --warmup 1 --no_cpu_baseline --no_fft_bench > $O/prof_fp32.log 2>&1
find $O/prof_fp32 -name "*kernel_trace.csv" -delete
find $O/prof_fp32 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fp32_kernel_stats.csv
head -30 $O/fp32_kernel_stats.csv | cut -c1-150
