#!/bin/bash
# PMC passes over the GEMM A/B bench (one shape family), run through gpurun from the repo root:
#   gpurun --timeout 600 -- 'bash tools/gemm_pmc.sh "conv1 fwd"'
# Counters in their own runs with --kernel-trace only (no --stats, no sys-trace), summaries under gpurun_out/pmc_gemm/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o run -- python3 $R/tools/gemm_ring_bench.py --one "$1" > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv, glob, collections, re
for i in (1,2,3):
    fs = glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs: print("pass", i, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in acc.items():
        print(i, k[-70:], {a: "%.4g" % b for a, b in v.items()})
PY
find $O -name "*.csv" -size +2M -delete
