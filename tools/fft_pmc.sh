#!/bin/bash
# SQ counters of the fused FFT projection kernels (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash tools/fft_pmc.sh'
# Counters in their own runs with --kernel-trace only; per-kernel averages -> gpurun_out/fft_pmc/summary.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fft_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o run -- python3 $R/tools/proj_bench.py --spectral --batch-only > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections, re
for i in (1, 2):
    fs = glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs: print("pass", i, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(paa::SpecArgs.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).strip() + " grid=" + r.get("Grid_Size", "?")
        if "spec" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k, v in acc.items():
        print(i, k[-60:], {a: "%.4g" % (b / cnt[k][a]) for a, b in v.items()})
PY
find $O -name "*.csv" -size +2M -delete
