#!/bin/bash
# FFT / projection path measurement on one MI355X (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash tools/fft_profile.sh'
# timing JSON, rocprofv3 --kernel-trace --stats summary and FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/fft/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fft
mkdir -p $O
cd $R
timeout -k 10 200 python tools/proj_bench.py > $O/proj_bench.json 2> $O/proj_bench.err; tail -30 $O/proj_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/tools/proj_bench.py --spectral > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fft_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o run -- python3 $R/tools/proj_bench.py --spectral > $O/pmc_$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re, json
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % C, recursive=True)
    if not fs: continue
    tot = collections.defaultdict(float); cnt = collections.Counter(); grid = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != C: continue
        k = re.sub(r"\([^()]*\)$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).strip() + " grid=" + r.get("Grid_Size", "?")
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in tot:
        out.setdefault(k, {})[C] = tot[k] / cnt[k]
        out[k]["launches"] = cnt[k]
res = []
for k, v in out.items():
    if "spec" not in k: continue
    fb = v.get("FETCH_SIZE", 0) * 1024 * 2; wb = v.get("WRITE_SIZE", 0) * 1024      # KB units; gfx950 FETCH_SIZE x2 for wide streaming reads
    res.append({"kernel": k, "launches": v["launches"], "fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb)})
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KB x1024, FETCH_SIZE doubled (MI355X_MICROARCH.md HBM)", "kernels": res}, open("$O/fft_hbm_pmc.json", "w"), indent=1)
for r in res: print(r)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -12 $O/fft_kernel_stats.csv | cut -c1-170
