"""Constraint projections on the HIP kernels — call surface of the reference's
``src/core/projections.py`` for the time-domain norms; the frequency-domain norms are fused with
their STFT / iSTFT in ``training_utils.train.perturbation_constraint`` (one launch sequence, no
spectrum in HBM), which is how the reference's dispatcher (train.py:38-66) always uses them."""
from __future__ import annotations

import types

import torch

from .. import _lib, runtime


def _project(p, clean, norm_type, **over):
    ns = types.SimpleNamespace(norm_type=norm_type, n_fft=1024, hop_length=256, win_length=1024, sr=16000, **over)
    q = runtime.as_f32_cuda(p, "p").clone()
    rows, L = (q.shape[0], q.shape[-1]) if q.dim() == 2 else (1, q.shape[-1])
    c = None if clean is None else runtime.as_f32_cuda(clean, "clean")
    pr = runtime.get_proj(ns, q.device, rows, max(L, 1024))
    prm = runtime.params_of(ns)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().paa_project(pr.h, prm, _lib.ptr(q), rows, _lib.ptr(c), 0 if c is None else c.shape[0], L,
                                          _lib.stream_ptr()))
    return q


def project_snr(clean, perturbation, snr_db):
    """projections.py:11-35."""
    return _project(perturbation, clean, "snr", snr_db=snr_db)


def project_linf(p, min_val, max_val):
    """projections.py:37-39: ``torch.clamp(p, min_val, max_val)`` — any box, not only the symmetric one the
    dispatcher passes (train.py:87)."""
    q = runtime.as_f32_cuda(p, "p").clone()
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().paa_clamp(_lib.ptr(q), q.numel(), float(min_val), float(max_val), _lib.stream_ptr()))
    return q


def project_l2(p, epsilon):
    """projections.py:41-46."""
    return _project(p, None, "l2", l2_size=float(epsilon))


def project_tv(p, args, clean_audio):
    """projections.py:56-66."""
    return _project(p, clean_audio, "tv", tv_epsilon=float(args.tv_epsilon))
