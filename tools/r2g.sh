set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "not full_size" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no_cpu_baseline --no_fft_bench > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['bf16']['value'],d['bf16']['ms_per_step'],d['bf16']['vs_headline_mode']['grad_sign_flip_rate']);print(d['roofline']['all_gemm_variants']);print(d['bf16']['roofline']['all_gemm_variants'])"
