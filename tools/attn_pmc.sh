#!/bin/bash
# SQ counters of the attention kernels inside one fp32-parity step (run through gpurun from the repo root): gpurun -- 'bash tools/attn_pmc.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/attn_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o run -- python3 $R/bench.py --dtype fp32 --steps 2 --warmup 1 --eager --no_cpu_baseline --no_fft_bench --no_prof --no_pmc > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections, re
for i in (1, 2):
    fs = glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs: print("pass", i, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).strip()
        if not ("attn" in k or "k_gemm_ring2<256" in k or "k_gemm_ring2<192, 256, 32, 1, 2, 4, true, true" in k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k, v in acc.items():
        print(i, k[-64:], {a: "%.4g" % (b / cnt[k][a]) for a, b in v.items()})
PY
find $O -name "*.csv" -size +2M -delete
