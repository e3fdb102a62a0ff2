"""ORACLE — test infrastructure only (never imported by the product path).

CPU (torch, float32) restatement of the reference's constraint projections
(/root/reference/src/core/projections.py), STFT/iSTFT (/root/reference/src/core/fourier_transforms.py)
and the dispatcher (/root/reference/src/training_utils/train.py:27-99).  Pinned by
tests/golden/projections.npz, produced by importing the reference itself (oracle/gen_goldens.py).

Third-party arithmetic restated: ``torch.stft`` / ``torch.istft`` (torch==2.5.0 pinned by the
reference's env.yaml:10; 2.10.0 in this image) — restated explicitly as reflect-pad, framing,
periodic Hann, rfft / irfft, overlap-add and envelope division so that the HIP kernels have a
step-by-step target.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

from . import iso226


def default_args(**kw):
    """The fields of the reference's argparse namespace that the hot path reads
    (src/training_utils/parser.py:10-66 defaults)."""
    a = dict(norm_type="max_phon", fm_epsilon=2.0, l2_size=0.05, linf_size=1e-4, snr_db=64.0,
             min_freq_attack=120.0, max_freq_attack=20000.0, tv_epsilon=0.001, max_phon_level=20.0,
             phon_reference_db=65.0, sr=16000, n_fft=1024, hop_length=256, win_length=1024,
             lr=1e-4, attack_mode="untargeted", optimizer_type="pgd", target="delete", target_reps=5)
    a.update(kw)
    return SimpleNamespace(**a)


# ----------------------------------------------------------------------------- STFT / iSTFT
def hann(n: int) -> torch.Tensor:
    """Periodic Hann, as torch.hann_window(n) (fourier_transforms.py:20)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * k / n)).to(torch.float32)


def compute_stft(p: torch.Tensor, args) -> torch.Tensor:
    """fourier_transforms.py:4-29: (B, L) f32 -> (B, n_fft/2+1, 1 + L//hop) complex64, center=True
    (reflect pad n_fft/2), periodic Hann, one-sided, unnormalised."""
    n, hop = args.n_fft, args.hop_length
    pp = torch.nn.functional.pad(p[:, None, :], (n // 2, n // 2), mode="reflect")[:, 0, :]
    frames = pp.unfold(-1, n, hop) * hann(n)                       # (B, T, n)
    return torch.fft.rfft(frames, dim=-1).transpose(1, 2)          # (B, F, T)


def compute_istft(stft_p: torch.Tensor, args) -> torch.Tensor:
    """fourier_transforms.py:31-41: inverse of the above, no ``length=`` -> hop * (T - 1) samples."""
    n, hop = args.n_fft, args.hop_length
    w = hann(n)
    B, _, T = stft_p.shape
    y = torch.fft.irfft(stft_p.transpose(1, 2), n=n, dim=-1) * w   # (B, T, n)
    total = n + hop * (T - 1)
    out = torch.zeros(B, total, dtype=torch.float32)
    env = torch.zeros(total, dtype=torch.float32)
    for t in range(T):
        out[:, t * hop:t * hop + n] += y[:, t]
        env[t * hop:t * hop + n] += w * w
    out = out[:, n // 2: total - n // 2]
    env = env[n // 2: total - n // 2]
    return out / env


def align_to(length: int, x: torch.Tensor) -> torch.Tensor:
    """train.py:27-35: right zero-pad or crop to ``length``."""
    if x.shape[-1] == length:
        return x
    if x.shape[-1] < length:
        return torch.nn.functional.pad(x, (0, length - x.shape[-1]))
    return x[..., :length]


# ----------------------------------------------------------------------------- time domain
def project_linf(p, lo, hi):
    """projections.py:37-39."""
    return torch.clamp(p, lo, hi)


def project_l2(p, eps):
    """projections.py:41-46."""
    norm = torch.norm(p, p=2)
    if norm > eps:
        return p * (eps / norm)
    return p


def project_snr(clean, p, snr_db):
    """projections.py:11-35 — note whole-batch signal power and the sqrt(numel) target."""
    sp = torch.mean(clean ** 2)
    npow = torch.mean(p ** 2)
    cur = 10 * torch.log10(sp / (npow + 1e-12))
    if cur >= snr_db:
        return p
    snr_linear = 10 ** (snr_db / 10)
    target = torch.sqrt(sp / snr_linear * clean.numel())
    cn = torch.norm(p.view(-1), p=2)
    if cn < 1e-8:
        return p
    return p * (target / cn)


def project_tv(p, tv_epsilon, clean):
    """projections.py:56-66."""
    base = torch.sum(torch.abs(clean[:, 1:] - clean[:, :-1]))
    eps = tv_epsilon * base
    tv = torch.sum(torch.abs(p[:, 1:] - p[:, :-1]))
    if tv > eps:
        return p * (eps / tv)
    return p


# ----------------------------------------------------------------------------- frequency domain
def project_min_max_freqs(args, stft_p):
    """projections.py:68-80 — keeps only bins OUTSIDE [min, max]."""
    freqs = torch.fft.rfftfreq(n=args.n_fft, d=1 / args.sr)
    mask = ((freqs < args.min_freq_attack) | (freqs > args.max_freq_attack)).float().view(1, -1, 1)
    return stft_p * mask


def fm_weighted_norm(stft_p, args):
    """projections.py:83-113 — scipy bilinear weights in float64, cast to float32, sqrt(sum(pow*w))."""
    B, F, T = stft_p.shape
    power = stft_p.abs() ** 2
    spl = 10 * torch.log10(power + 1e-10)
    freqs = torch.fft.rfftfreq(n=args.n_fft, d=1 / args.sr).view(1, F, 1).expand(B, F, T)
    q = torch.stack([spl, freqs], dim=-1).reshape(-1, 2).numpy()
    w = torch.tensor(iso226.interp_weights(q).reshape(B, F, T), dtype=torch.float32)
    return torch.sqrt((power * w).sum())


def project_fm_norm(stft_p, args):
    """projections.py:116-133."""
    norm = fm_weighted_norm(stft_p, args)
    if norm <= args.fm_epsilon:
        return stft_p
    return stft_p * (args.fm_epsilon / norm.clamp(min=1e-8))


def project_phon_level(stft_p, args, spl_thresh):
    """projections.py:138-159 — every bin is rebuilt from (clipped dB magnitude, phase)."""
    mag_db = 20 * torch.log10(stft_p.abs() + 1e-8)
    thr = spl_thresh - spl_thresh.max() + args.phon_reference_db
    mag = 10 ** (torch.where(mag_db > thr, thr, mag_db) / 20)
    return mag * torch.exp(1j * stft_p.angle())


def perturbation_constraint(p, clean, args, spl_thresh=None):
    """train.py:38-99 dispatcher; ``spl_thresh`` is the (1, F, 1) tensor of build.py:325-348."""
    nt = args.norm_type
    if nt in ("fletcher_munson", "min_max_freqs", "max_phon"):
        s = compute_stft(p, args)
        if nt == "min_max_freqs":
            s = project_min_max_freqs(args, s)
        elif nt == "fletcher_munson":
            s = project_fm_norm(s, args)
        else:
            s = project_phon_level(s, args, spl_thresh)
        y = compute_istft(s, args)
        return align_to(clean.shape[-1], y) if clean is not None else y
    if nt == "l2":
        return project_l2(p, args.l2_size)
    if nt == "linf":
        return project_linf(p, -args.linf_size, args.linf_size)
    if nt == "snr":
        if clean is None:
            raise ValueError("SNR projection requires clean_audio ro compare to")
        return project_snr(clean, p, args.snr_db)
    if nt == "tv":
        if clean is None:
            raise ValueError("TV projection can benefit from clean_audio for bounds")
        return project_tv(p, args.tv_epsilon, clean)
    raise ValueError(f"Unknown norm_type: {nt!r}")


def spl_thresh_tensor(args) -> torch.Tensor:
    """build.py:325-348 -> (1, F, 1) float32."""
    return torch.from_numpy(iso226.phon_threshold(args.max_phon_level, args.n_fft, args.sr)).view(1, -1, 1)
