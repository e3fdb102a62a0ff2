#!/bin/bash
# Builds diagnostic variants of the library with parts of the ring kernel removed (gemm_ring.hip, PAA_ABL bits) into
# psychoacoustic-adverserial-attacks_amd/build/abl/ — run HERE (no GPU needed), then on the box: python tools/gemm_ablate.py
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/psychoacoustic-adverserial-attacks_amd
mkdir -p $P/build/abl
python -c "import sys; sys.path.insert(0, '$P'); import build_ext; build_ext.build()" > /dev/null
for X in 0 1 3 7 15; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -DPAA_ABL=$X -c $P/csrc/gemm_ring.hip -o $P/build/abl/gemm_ring_$X.o &
done
wait
for X in 0 1 3 7 15; do
  OBJS=$(ls $P/build/*.o | grep -v gemm_ring.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/abl/libpaa_abl_$X.so $OBJS $P/build/abl/gemm_ring_$X.o
done
ls -la $P/build/abl/*.so
