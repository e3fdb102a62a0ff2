"""Helpers for the -m gpu tests: run a paa_gemm descriptor on device buffers, report errors."""
import ctypes as C

import numpy as np
import torch

from paa_amd import _lib


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def run_gemm(d: dict, bufs: dict, offs: dict = None):
    """d: descriptor fields (ints/floats) with operand keys A,B,C,(bias,aux,residual,C_pre) naming entries of
    ``bufs`` (torch cuda float32 tensors); offs: element offsets per operand name."""
    offs = offs or {}
    desc = _lib.PaaGemmDesc()
    desc.a_kcontig = 1
    desc.b_kcontig = 1
    desc.batch = 1
    desc.batch2 = 1
    desc.alpha = 1.0
    for k, v in d.items():
        if k in ("A", "B", "C", "bias", "aux", "residual", "C_pre", "A_lo", "B_lo", "Cb", "Cb_lo"):
            t = bufs[v]
            base = {'A_lo': 'A', 'B_lo': 'B', 'Cb': 'C', 'Cb_lo': 'C', 'C_pre': 'C'}.get(k, k)
            setattr(desc, k, t.data_ptr() + t.element_size() * offs.get(k, offs.get(base, 0)))
        else:
            setattr(desc, k, v)
    _lib.check(_lib.lib().paa_gemm(C.byref(desc), _lib.stream_ptr()))
    torch.cuda.synchronize()


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
