"""Runs tools/gemm_ring_bench.py against one of the diagnostic libraries tools/gemm_ablate.sh built (PAA_ABL bits: 1 no epilogue,
2 no operand DMA, 4 no barrier, 8 no fragment reads) — results are wrong by construction, only the timings mean something.
    ABL=7 PAA_SQ_PROBE=1 PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python tools/gemm_ablate.py --one probe
(the ablated kernels are the two-stage rings of gemm_ring.hip, which only the diagnostic library holds)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paa_amd import _lib

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "build_exp", "abl", f"libpaa_abl_{os.environ.get('ABL', '0')}.so")
_lib._check_current = lambda: None          # an ablation build is not the library of the sources beside it, on purpose
import gemm_ring_bench  # noqa: E402  (tools/ is this script's directory)

gemm_ring_bench.main()
