"""Numpy statement of the paa_gemm descriptor semantics (csrc/gemm.h), used by the tests both as the
reference for the HIP kernel (GPU) and to check, without a GPU, that the descriptors csrc/model.hip
builds for the convolutions compute what torch's conv1d / its backward compute (host logic)."""
from __future__ import annotations

import math

import numpy as np


def gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    from scipy.special import erf
    return 0.5 * (1.0 + erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


DEFAULTS = dict(lda=0, a_kcontig=1, a_kseg=0, a_kseg_stride=0, a_window=0, a_pad=0, a_rows_valid=0, ldb=0, b_kcontig=1,
                ldc=0, batch=1, batch2=1, a_s1=0, a_s2=0, b_s1=0, b_s2=0, c_s1=0, c_s2=0, alpha=1.0, bias=None, bias_s2=0,
                act=0, C_pre=None, aux=None, ld_aux=0, aux_s1=0, aux_s2=0, residual=None, ld_res=0, res_s1=0, res_s2=0,
                row_period=0, row_valid=0, accumulate=0, precision=0, aux_gate=0)


def emulate(d: dict, A: np.ndarray, B: np.ndarray, C: np.ndarray, a_off=0, b_off=0, c_off=0, bias=None, aux=None,
            aux_off=0, residual=None, res_off=0, C_pre=None):
    """Apply descriptor ``d`` to flat float arrays (offsets in elements).  Writes C (and C_pre) in place, float64 math."""
    g = dict(DEFAULTS)
    g.update(d)
    M, N, K = g["M"], g["N"], g["K"]
    m = np.arange(M)[:, None]
    k = np.arange(K)[None, :]
    n = np.arange(N)[None, :]
    for z in range(g["batch"]):
        z1, z2 = divmod(z, g["batch2"])
        ao = a_off + z1 * g["a_s1"] + z2 * g["a_s2"]
        bo = b_off + z1 * g["b_s1"] + z2 * g["b_s2"]
        co = c_off + z1 * g["c_s1"] + z2 * g["c_s2"]
        if g["a_kcontig"]:
            if g["a_window"]:
                js, kc = k // g["a_kseg"], k % g["a_kseg"]
                tr = m + js - g["a_pad"]
                ok = (tr >= 0) & (tr < g["a_rows_valid"])
                idx = ao + np.where(ok, tr, 0) * g["lda"] + kc
                Ad = np.where(ok, A[idx], 0.0)
            else:
                koff = (k // g["a_kseg"]) * g["a_kseg_stride"] + k % g["a_kseg"] if g["a_kseg"] else k
                Ad = A[ao + m * g["lda"] + koff]
        else:
            Ad = A[ao + k * g["lda"] + m]
        kk = np.arange(K)[:, None]
        Bd = B[bo + n * g["ldb"] + kk] if g["b_kcontig"] else B[bo + kk * g["ldb"] + n]
        v = (Ad.astype(np.float64) @ Bd.astype(np.float64)) * g["alpha"]
        if bias is not None:
            v = v + bias[z2 * g["bias_s2"] + np.arange(N)][None, :]
        cidx = co + m * g["ldc"] + n
        if g["act"] == 1:
            if C_pre is not None:
                C_pre[cidx] = gelu_grad(v) if g["aux_gate"] else v
            v = gelu(v)
        elif g["act"] == 2:
            ax = aux[aux_off + z1 * g["aux_s1"] + z2 * g["aux_s2"] + m * g["ld_aux"] + n].astype(np.float64)
            v = v * (ax if g["aux_gate"] else gelu_grad(ax))
        if residual is not None:
            v = v + residual[res_off + z1 * g["res_s1"] + z2 * g["res_s2"] + m * g["ld_res"] + n]
        if g["row_period"] > 0:
            dead = (np.arange(M) % g["row_period"]) >= g["row_valid"]
            v[dead] = 0.0
            if C_pre is not None and g["act"] == 1:
                C_pre[cidx[dead]] = 0.0
        if g["accumulate"]:
            v = v + C[cidx]
        C[cidx] = v
    return C


def padded_rows(a, length):
    """The padded row counts csrc/model.hip derives: P_last = T_last + extra, P_{i-1} = stride_i * P_i."""
    T = a.feat_lengths(length)
    for extra in range(1, 65):
        p = T[-1] + extra
        P = [0] * len(T)
        ok = True
        for i in range(len(T) - 1, -1, -1):
            P[i] = p
            ok &= p >= T[i]
            p *= a.conv_stride[i]
        if ok:
            return T, P
    raise ValueError("no layout")
