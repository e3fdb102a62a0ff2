#!/bin/bash
# Builds diagnostic variants of the library with parts of the ring kernel removed (gemm_ring.hip, PAA_ABL bits) into
# psychoacoustic-adverserial-attacks_amd/build_exp/abl/ — run HERE (no GPU needed), then on the box:
#   PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python tools/gemm_ablate.py
# (the ablation bits, like every diagnostic, exist only in -DPAA_EXPERIMENTS builds: the objects of the diagnostic library are reused)
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/psychoacoustic-adverserial-attacks_amd
mkdir -p $P/build_exp/abl
PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python -c "import sys; sys.path.insert(0, '$P'); import build_ext; build_ext.build()" > /dev/null
for X in 0 1 3 7 15; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -DPAA_EXPERIMENTS -DPAA_ABL=$X -c $P/csrc/gemm_ring.hip -o $P/build_exp/abl/gemm_ring_$X.o &
done
wait
for X in 0 1 3 7 15; do
  OBJS=$(ls $P/build_exp/*.o | grep -v gemm_ring.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build_exp/abl/libpaa_abl_$X.so $OBJS $P/build_exp/abl/gemm_ring_$X.o
done
ls -la $P/build_exp/abl/*.so
