#!/bin/bash
# Same-box A/B of several builds of the library, alternating processes:  bash tools/scratch/ab_libs.sh name1=path1 name2=path2 ...
# (each build is copied over libpaa_hip.so for its runs; the shipped library is restored at the end)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=psychoacoustic-adverserial-attacks_amd/libpaa_hip.so
cp $L /tmp/lib_head.so
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}
    cp $path $L
    timeout -k 10 200 python bench.py --dtype fp32 --steps 40 --warmup 5 --no_cpu_baseline --no_fft_bench --no_pmc 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
cp /tmp/lib_head.so $L
