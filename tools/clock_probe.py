"""In-kernel clock held under the ring GEMM kernels (diagnostic build: PAA_EXTRA_HIPCC_FLAGS="-DPAA_EXPERIMENTS -DPAA_CLOCK_STAMP"):
shader ticks / 100 MHz real-time ticks around each workgroup's tile loop, after >= 2 s of back-to-back launches on random
data.  Prints GHz per (precision, shape).  The shipped library carries no stamps."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from paa_amd import _lib
import gemm_ring_bench as gb

L = _lib.lib()
fn = L.paa_debug_ring_clock_ghz
fn.restype = C.c_double
fn.argtypes = [C.c_int]
for prec, cfg in ((0, 8), (0, 5), (1, 7), (1, 4)):
    for (nm, M, N, K, lda, ep) in (("conv1 fwd gelu", 512000, 512, 1536, 1024, "gelu"), ("ffn2 resid", 16000, 768, 3072, None, "res")):
        d, outs, keep = gb.build(M, N, K, lda, ep, prec)
        L.paa_gemm_config(cfg)
        st = _lib.stream_ptr()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.5:
            for _ in range(20):
                L.paa_gemm(C.byref(d), st)
            torch.cuda.synchronize()
            n += 20
        ms = gb.timeit(d, 10)
        ghz = fn(256 if cfg in (4, 5) else 512)
        print(f"prec={prec} cfg={cfg} {nm:16s}: {2.0 * M * N * K / ms / 1e9:7.1f} TF  in-kernel clock {ghz:.3f} GHz  (clock-adjusted dense bf16 MFMA peak {2500 * ghz / 2.4:.0f} TF)", flush=True)
        del d, outs, keep
        torch.cuda.empty_cache()
L.paa_gemm_config(0)
