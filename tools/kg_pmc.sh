#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the conv1 products in plain K order and in gemm.h's k_group order (gpurun, repo root):
#   gpurun --timeout 900 -- 'bash tools/kg_pmc.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/kg_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PAA_KG_PROBE=1
for C in ${KG_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  for S in "conv1 fwd gelu" "conv1 fwd gelu kg" "conv1 dgrad even" "conv1 dgrad even kg"; do
    if [ -n "$KG_ONLY" ] && [ "$S" != "$KG_ONLY" ]; then continue; fi
    D=$O/${C}_$(echo $S | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o run -- python3 $R/tools/gemm_ring_bench.py --exact --one "$S" > $D.log 2>&1 || { echo "$C $S failed"; tail -5 $D.log; }
  done
done
python3 - <<PY
import csv, glob, collections, re
for d in sorted(glob.glob("$O/*/")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no output"); continue
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).strip()
        if "gemm" not in k: continue
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in acc:
        mult = 2 if "FETCH" in d else 1
        print(d.split("/")[-2], k[-60:], "launches", cnt[k], "MB/launch %.1f" % (acc[k] * 1024 * mult / cnt[k] / 1e6))
PY
find $O -name "*.csv" -size +2M -delete
