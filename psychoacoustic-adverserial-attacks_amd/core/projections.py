"""Constraint projections on the HIP kernels — call surface of the reference's ``src/core/projections.py``:
``project_snr / project_linf / project_l2 / project_tv`` on waveforms and ``project_min_max_freqs /
compute_fm_weighted_norm_interp / project_fm_norm / project_phon_level`` on a complex (B, F, T) STFT tensor, with the
reference's argument order.  The dispatcher (``training_utils.train.perturbation_constraint``, train.py:38-66) does not
go through the spectrum-level functions: it uses the fused STFT -> per-bin op -> iSTFT launch, which never
materialises the spectrum."""
from __future__ import annotations

import types

import torch

from .. import _lib, runtime


def _project(p, clean, norm_type, **over):
    ns = types.SimpleNamespace(norm_type=norm_type, n_fft=1024, hop_length=256, win_length=1024, sr=16000, **over)
    src = runtime.as_f32_cuda(p, "p")
    q = torch.empty_like(src)
    rows, L = (q.shape[0], q.shape[-1]) if q.dim() == 2 else (1, q.shape[-1])
    c = None if clean is None else runtime.as_f32_cuda(clean, "clean")
    pr = runtime.get_proj(ns, q.device, rows, max(L, 1024))
    prm = runtime.params_of(ns)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().paa_project_to(pr.h, prm, _lib.ptr(src), _lib.ptr(q), rows, _lib.ptr(c),
                                             0 if c is None else c.shape[0], L, _lib.stream_ptr()))
    return q


def _spectrum(stft_p):
    """complex (B, F, T) tensor -> contiguous (B, T, F, 2) float32 storage (a no-op for what compute_stft returns)."""
    if not stft_p.is_cuda:
        raise RuntimeError(f"stft_p must live on the GPU (got {stft_p.device}); there is no CPU fallback")
    if stft_p.dtype != torch.complex64 or stft_p.dim() != 3:
        raise TypeError(f"stft_p must be a complex64 (B, F, T) tensor, got {stft_p.dtype} {tuple(stft_p.shape)}")
    return torch.view_as_real(stft_p.transpose(1, 2).contiguous())


def _spectrum_project(stft_p, args, norm_type, interp=None, spl_thresh=None, **over):
    S = _spectrum(stft_p)
    B, T, F, _ = S.shape
    ns = types.SimpleNamespace(**{**vars(args), "norm_type": norm_type, **over})
    pr = runtime.get_proj(ns, S.device, B, int(ns.hop_length) * T + int(ns.n_fft), interp)
    if F != pr.F:
        raise ValueError(f"stft_p has {F} bins, expected {pr.F}")
    if spl_thresh is not None:
        pr.set_spl_thresh(spl_thresh)
    out = torch.empty_like(S)
    with torch.cuda.device(S.device):
        _lib.check(_lib.lib().paa_spectrum_project(pr.h, runtime.params_of(ns), _lib.ptr(S), _lib.ptr(out), B, T, _lib.stream_ptr()))
    return torch.view_as_complex(out).transpose(1, 2)


def project_min_max_freqs(args, stft_p, min_freq, max_freq):
    """projections.py:68-80: keeps only the bins OUTSIDE [min_freq, max_freq] (the reference's mask)."""
    return _spectrum_project(stft_p, args, "min_max_freqs", min_freq_attack=float(min_freq), max_freq_attack=float(max_freq))


def compute_fm_weighted_norm_interp(stft_p, interp, args):
    """projections.py:83-113 -> 0-d tensor sqrt(sum |S|^2 w(SPL, f))."""
    S = _spectrum(stft_p)
    B, T, F, _ = S.shape
    ns = types.SimpleNamespace(**{**vars(args), "norm_type": "fletcher_munson"})
    pr = runtime.get_proj(ns, S.device, B, int(ns.hop_length) * T + int(ns.n_fft), interp)
    if F != pr.F:
        raise ValueError(f"stft_p has {F} bins, expected {pr.F}")
    out = torch.empty(1, dtype=torch.float32, device=S.device)
    with torch.cuda.device(S.device):
        _lib.check(_lib.lib().paa_fm_weighted_norm(pr.h, _lib.ptr(S), B, T, _lib.ptr(out), _lib.stream_ptr()))
    return out[0]


def project_fm_norm(stft_p, args, interp):
    """projections.py:116-133: S * fm_epsilon / max(norm, 1e-8) if norm > fm_epsilon else S (predicated on the device)."""
    return _spectrum_project(stft_p, args, "fletcher_munson", interp=interp)


def project_phon_level(stft_p, args, spl_thresh, plot_debug=False, tag=""):
    """projections.py:138-159 (``plot_debug`` / ``tag`` accepted for call compatibility; plots are out of scope)."""
    return _spectrum_project(stft_p, args, "max_phon", spl_thresh=spl_thresh)


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
