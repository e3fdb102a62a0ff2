"""STFT / iSTFT on the HIP FFT kernels — same call surface as the reference's
``src/core/fourier_transforms.py`` (``compute_stft(p, args)``, ``compute_istft(stft_p, args)``)."""
from __future__ import annotations

import torch

from .. import _lib, runtime


def compute_stft(p: torch.Tensor, args) -> torch.Tensor:
    """(B, L) float32 -> (B, n_fft/2+1, 1 + L//hop) complex64, as torch.stft(center=True, hann, onesided)
    (fourier_transforms.py:20-29).  The kernel writes frame-major (B, T, F); the returned tensor is the
    transposed view of that storage."""
    x = runtime.as_f32_cuda(p, "p")
    if x.dim() == 1:
        x = x[None]
    B, L = x.shape
    pr = runtime.get_proj(args, x.device, B, L)
    T = 1 + L // pr.hop
    out = torch.empty(B, T, pr.F, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().paa_stft(pr.h, _lib.ptr(x), B, L, _lib.ptr(out), _lib.stream_ptr()))
    return torch.view_as_complex(out).transpose(1, 2)


def compute_istft(stft_p: torch.Tensor, args) -> torch.Tensor:
    """(B, F, T) complex64 -> (B, hop * (T - 1)) float32 (fourier_transforms.py:31-41; no ``length=``)."""
    if not stft_p.is_cuda:
        raise RuntimeError("stft_p must live on the GPU; there is no CPU fallback")
    S = torch.view_as_real(stft_p.transpose(1, 2).contiguous()).contiguous()      # (B, T, F, 2)
    B, T, F, _ = S.shape
    pr = runtime.get_proj(args, S.device, B, pr_len(args, T))
    if F != pr.F:
        raise ValueError(f"stft_p has {F} bins, expected {pr.F}")
    out = torch.empty(B, pr.hop * (T - 1), dtype=torch.float32, device=S.device)
    with torch.cuda.device(S.device):
        _lib.check(_lib.lib().paa_istft(pr.h, _lib.ptr(S), B, T, _lib.ptr(out), _lib.stream_ptr()))
    return out


def pr_len(args, T: int) -> int:
    return int(args.hop_length) * T + int(args.n_fft)
