"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the reference's ISO-226 equal-loudness tables
(/root/reference/src/core/iso.py) and of the per-bin phon threshold
(/root/reference/src/training_utils/build.py:325-348).

Third-party arithmetic restated here: scipy (unpinned in the reference's env.yaml; 1.15.3 in this
image) ``PchipInterpolator`` and ``RegularGridInterpolator(method="linear", bounds_error=False,
fill_value=1.0)``.  Pinned by tests/golden/iso.npz, which was produced by importing the
reference itself (oracle/gen_goldens.py).
"""
from __future__ import annotations

import numpy as np
from scipy.interpolate import PchipInterpolator

# iso.py:60-84 — the 29 one-third-octave bands and their alpha / L_u / T_f parameters.
FREQS = (20.0, 25.0, 31.5, 40.0, 50.0, 63.0, 80.0, 100.0, 125.0, 160.0, 200.0, 250.0, 315.0,
         400.0, 500.0, 630.0, 800.0, 1000.0, 1250.0, 1600.0, 2000.0, 2500.0, 3150.0, 4000.0,
         5000.0, 6300.0, 8000.0, 10000.0, 12500.0)
ALPHA = (0.532, 0.506, 0.480, 0.455, 0.432, 0.409, 0.387, 0.367, 0.349, 0.330, 0.315, 0.301,
         0.288, 0.276, 0.267, 0.259, 0.253, 0.250, 0.246, 0.244, 0.243, 0.243, 0.243, 0.242,
         0.242, 0.245, 0.254, 0.271, 0.301)
L_U = (-31.6, -27.2, -23.0, -19.1, -15.9, -13.0, -10.3, -8.1, -6.2, -4.5, -3.1, -2.0, -1.1,
       -0.4, 0.0, 0.3, 0.5, 0.0, -2.7, -4.1, -1.0, 1.7, 2.5, 1.2, -2.1, -7.1, -11.2, -10.7,
       -3.1)
T_F = (78.5, 68.7, 59.5, 51.1, 44.0, 37.5, 31.5, 26.5, 22.1, 17.9, 14.4, 11.4, 8.6, 6.2,
       4.4, 3.0, 2.2, 2.4, 3.5, 1.7, -1.3, -4.2, -6.0, -5.4, -1.5, 6.0, 12.6, 13.9, 12.3)


def iso226_spl(phon: float, freqs) -> np.ndarray:
    """SPL (dB) needed at ``freqs`` for loudness ``phon`` — iso.py:86-124 (PCHIP over the 29 bands
    plus a 20 kHz knot that repeats the 20 Hz value) and iso.py:161-171 (closed form)."""
    if phon < 0 or phon > 90:                                     # iso.py:97-98
        raise ValueError("Phon must be in range [0, 90]")
    freqs = np.asarray(freqs, dtype=np.float64)
    if np.any(freqs < 20.0) or np.any(freqs > 20000.0):           # iso.py:152-153
        raise ValueError("Frequency must be in [20, 20000] Hz")
    knots = np.array(FREQS + (20000.0,))
    alpha = PchipInterpolator(knots, np.array(ALPHA + (ALPHA[0],)))(freqs)
    lu = PchipInterpolator(knots, np.array(L_U + (L_U[0],)))(freqs)
    tf = PchipInterpolator(knots, np.array(T_F + (T_F[0],)))(freqs)
    a = 0.00447 * ((10.0 ** (0.025 * phon)) - 1.15)
    b = (0.4 * (10.0 ** (((tf + lu) / 10.0) - 9.0))) ** alpha
    return ((10.0 / alpha) * np.log10(a + b)) - lu + 94.0


def weight_grid():
    """(phons[10], freqs[30], W[10,30]) — iso.py:176-235: SPL grid → clip((1 - SPL/max)^2, 0, 1)."""
    phons = np.arange(0, 100, 10)
    freqs = np.array(FREQS + (20000.0,))
    spl = np.array([iso226_spl(float(p), freqs) for p in phons])
    w = np.clip((1.0 - spl / spl.max()) ** 2, 0.0, 1.0)
    return phons.astype(np.float64), freqs, w


def interp_weights(points: np.ndarray) -> np.ndarray:
    """Restatement of ``RegularGridInterpolator((phons, freqs), W, bounds_error=False,
    fill_value=1.0)`` (iso.py:261-266) evaluated at ``points[:, (phon, freq)]`` in float64."""
    phons, freqs, w = weight_grid()
    pts = np.asarray(points, dtype=np.float64)
    s, f = pts[:, 0], pts[:, 1]
    oob = (s < phons[0]) | (s > phons[-1]) | (f < freqs[0]) | (f > freqs[-1]) | np.isnan(s)
    s_safe = np.where(oob, phons[0], s)
    f_safe = np.where(oob, freqs[0], f)
    i = np.clip(np.searchsorted(phons, s_safe) - 1, 0, len(phons) - 2)
    j = np.clip(np.searchsorted(freqs, f_safe) - 1, 0, len(freqs) - 2)
    ys = (s_safe - phons[i]) / (phons[i + 1] - phons[i])
    yf = (f_safe - freqs[j]) / (freqs[j + 1] - freqs[j])
    out = (w[i, j] * (1 - ys) * (1 - yf) + w[i, j + 1] * (1 - ys) * yf
           + w[i + 1, j] * ys * (1 - yf) + w[i + 1, j + 1] * ys * yf)
    return np.where(oob, 1.0, out)


def phon_threshold(max_phon_level: float, n_fft: int = 1024, sr: int = 16000) -> np.ndarray:
    """(F,) float32 contour — build.py:331-347: ISO226(phon)(clip(rfftfreq, 20, 20000))."""
    freqs = np.fft.rfftfreq(n_fft, d=1.0 / sr)
    return iso226_spl(float(max_phon_level), np.clip(freqs, 20.0, 20000.0)).astype(np.float32)
