"""Per-device projection context cache + argparse-namespace -> paa_params translation."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .core import iso

_PROJ = {}
_TABLE = None


def weight_table():
    global _TABLE
    if _TABLE is None:
        _TABLE = iso.build_weight_interpolator()
    return _TABLE


class Proj:
    """Owns one ``paa_proj`` handle (FFT twiddles, window, FM table, max_phon contour, workspace)."""

    def __init__(self, device, n_fft, hop, win, sr, max_batch, max_len, interp=None):
        self.device = torch.device(device)
        self.key = (n_fft, hop, win, sr)
        self.max_batch, self.max_len = max_batch, max_len
        self.F = n_fft // 2 + 1
        self.hop, self.n_fft = hop, n_fft
        tab = (interp if isinstance(interp, iso.WeightTable) else weight_table()).for_bins(n_fft, sr)
        self._fm = np.ascontiguousarray(tab, dtype=np.float64)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().paa_proj_create(C.byref(h), n_fft, hop, win, sr, self._fm.ctypes.data_as(C.c_void_p),
                                                  None, max_batch, max_len))
        self.h = h
        self._thr_key = None

    def set_spl_thresh(self, spl_thresh):
        """spl_thresh: the (1, F, 1) tensor of build.init_phon_threshold_tensor (uploaded when it changes)."""
        if spl_thresh is None:
            return
        # identity + version; the tensor is kept alive so its id / storage cannot be recycled by another contour
        key = (id(spl_thresh), spl_thresh._version)
        if key == self._thr_key:
            return
        host = np.ascontiguousarray(spl_thresh.detach().reshape(-1).to("cpu", torch.float32).numpy())
        if host.shape[0] != self.F:
            raise ValueError(f"spl_thresh has {host.shape[0]} bins, expected {self.F}")
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().paa_proj_set_spl_thresh(self.h, host.ctypes.data_as(C.c_void_p)))
        self._thr_key = key
        self._thr_ref = spl_thresh

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().paa_proj_destroy(self.h)
        except Exception:
            pass


def get_proj(args, device, rows: int, length: int, interp=None) -> Proj:
    """One projection context per (device, frame geometry, weight table).  A custom ``iso.WeightTable`` gets a context of
    its own (its identity is part of the key: it is never silently replaced by the default table); a context that has
    to grow for a larger batch / length keeps its table and its max_phon contour."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("paa_amd runs on the GPU only (device %r); there is no CPU fallback" % (device,))
    table = interp if isinstance(interp, iso.WeightTable) else None
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           int(args.n_fft), int(args.hop_length), int(args.win_length), int(args.sr), id(table) if table is not None else None)
    pr = _PROJ.get(key)
    if pr is None or pr.max_batch < rows or pr.max_len < length:
        mb = max(rows, pr.max_batch if pr else 1)
        ml = max(length, pr.max_len if pr else 1)
        new = Proj(device, key[1], key[2], key[3], key[4], mb, ml, table)
        new._table = table                      # keeps the table alive: its id is in the key
        if pr is not None and getattr(pr, "_thr_ref", None) is not None:
            new.set_spl_thresh(pr._thr_ref)
        pr = new
        _PROJ[key] = pr
    return pr


def params_of(args) -> _lib.PaaParams:
    """argparse namespace (training_utils/parser.py) -> paa_params."""
    nt = getattr(args, "norm_type", None)
    if nt not in _lib.NORM_IDS:
        raise ValueError(f"Unknown norm_type: {nt!r}")          # train.py:98
    g = lambda k, d: float(getattr(args, k, d))
    return _lib.PaaParams(_lib.NORM_IDS[nt], g("l2_size", 0.05), g("linf_size", 1e-4), g("snr_db", 64), g("tv_epsilon", 1e-3),
                          g("fm_epsilon", 2), g("min_freq_attack", 120), g("max_freq_attack", 20000),
                          g("phon_reference_db", 65), g("lr", 1e-4),
                          +1 if getattr(args, "attack_mode", "untargeted") == "untargeted" else -1)


def as_f32_cuda(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()
