#!/bin/bash
# Same-box A/B of two builds of libpaa_hip.so (HEAD vs tools/scratch/libpaa_hip_r3start.so), alternating processes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=psychoacoustic-adverserial-attacks_amd/libpaa_hip.so
cp $L /tmp/lib_head.so
for round in 1 2; do
  for which in head old; do
    if [ $which = head ]; then cp /tmp/lib_head.so $L; else cp tools/scratch/libpaa_hip_r3start.so $L; fi
    timeout -k 10 200 python bench.py --dtype fp32 --steps 40 --warmup 5 --no_cpu_baseline --no_fft_bench --no_pmc 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
cp /tmp/lib_head.so $L
