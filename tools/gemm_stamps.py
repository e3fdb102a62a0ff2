"""Where a K slab of the ring GEMM goes (diagnostic build with s_memtime stamps; run on the GPU box):

    PAA_EXTRA_HIPCC_FLAGS="-DPAA_EXPERIMENTS -DPAA_R2_STAMP" python tools/gemm_stamps.py

For the model's large split-mode shapes: workgroup 0's eight waves, per tile — cycles of the K loop, of which parked at the counted
vmcnt wait (operands not landed) and at the barrier (waiting for the slowest wave), and the epilogue.  MFMA floor of a slab:
48 MFMAs x 32 cycles x 2 waves per SIMD = 3072 cycles."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PAA_EXTRA_HIPCC_FLAGS", "-DPAA_EXPERIMENTS -DPAA_R2_STAMP")
import numpy as np
import torch

from paa_amd import build_ext

build_ext.build(verbose=False)
import gemm_ring_bench as G          # noqa: E402  (tools/ on sys.path through this file's directory)

L = G.L
L.paa_debug_r2_stamps.argtypes = [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * (8 * 16 * 6))()
for (nm, M, N, K, lda, ep) in G.SHAPES:
    if nm.endswith(" kg"):
        continue
    d, outs, keep = G.build(M, N, K, lda, ep, 1, 0)
    L.paa_gemm_config(0)
    for _ in range(3):
        ms = G.timeit(d, 3)
    assert L.paa_debug_r2_stamps(buf) == 0            # (also clears the table)
    G.timeit(d, 1)
    assert L.paa_debug_r2_stamps(buf) == 0
    s = np.array(buf, dtype=np.int64).reshape(8, 16, 6)
    tiles = int((s[0, :, 0] > 0).sum())
    use = s[:, 1:max(tiles, 2), :] if tiles > 2 else s[:, :max(tiles, 1), :]      # skip the first tile (cold start) where there are more
    nk = use[..., 0].mean()
    loop, vm, bar, epi = use[..., 1].mean(), use[..., 2].mean(), use[..., 3].mean(), use[..., 4].mean()
    fl = 2.0 * M * N * K
    print(f"{nm:22s} M={M:7d} N={N:5d} K={K:5d}  {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF | tiles seen {tiles:2d} slabs/tile {nk:5.1f} | per slab: {loop / nk:7.0f} cyc "
          f"(vmcnt wait {vm / nk:6.0f}, barrier {bar / nk:6.0f}, rest {(loop - vm - bar) / nk:6.0f}) | epilogue {epi:7.0f} cyc = {epi / (loop + epi) * 100:4.1f} % of the tile", flush=True)
    # per-wave spread of the two waits (tile 1)
    k = 1 if tiles > 1 else 0
    print("      per wave (tile %d): vmcnt " % k + " ".join(f"{v / max(s[w, k, 0], 1):5.0f}" for w, v in enumerate(s[:, k, 2])) +
          " | barrier " + " ".join(f"{v / max(s[w, k, 0], 1):5.0f}" for w, v in enumerate(s[:, k, 3])), flush=True)
    del d, outs, keep
    torch.cuda.empty_cache()
