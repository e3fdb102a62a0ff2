"""Setup-time ISO-226 tables for the Fletcher-Munson and max_phon projections (host side, numpy).

Mirrors the interface of the reference's ``src/core/iso.py`` (``ISO226``,
``compute_iso226_weight_matrix``, ``perceptual_weight``, ``build_weight_interpolator``) but returns a
table object instead of a scipy interpolator: the per-element bilinear lookup the reference does on
the CPU (projections.py:104-109) runs inside the HIP frame kernel, which needs the weight grid
pre-interpolated along frequency to the rfft bins (``WeightTable.for_bins``).  PCHIP is implemented
here directly (Fritsch-Carlson, as scipy.interpolate.PchipInterpolator) so the runtime has no scipy
dependency; tests pin it against the goldens generated from the reference.
"""
from __future__ import annotations

import numpy as np

# iso.py:60-84 — ISO 226 parameters at the 29 one-third-octave bands.
FREQUENCIES = (20.0, 25.0, 31.5, 40.0, 50.0, 63.0, 80.0, 100.0, 125.0, 160.0, 200.0, 250.0, 315.0,
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


class _Pchip:
    """Monotone cubic Hermite interpolation with Fritsch-Carlson slopes (scipy's PCHIP)."""

    def __init__(self, x, y):
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        h = np.diff(self.x)
        m = np.diff(self.y) / h
        d = np.zeros_like(self.y)
        w1 = 2 * h[1:] + h[:-1]
        w2 = h[1:] + 2 * h[:-1]
        same = (np.sign(m[:-1]) * np.sign(m[1:])) > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            whm = (w1 / m[:-1] + w2 / m[1:]) / (w1 + w2)
        d[1:-1] = np.where(same, 1.0 / np.where(same, whm, 1.0), 0.0)
        d[0] = self._edge(h[0], h[1], m[0], m[1])
        d[-1] = self._edge(h[-1], h[-2], m[-1], m[-2])
        self.d = d

    @staticmethod
    def _edge(h0, h1, m0, m1):
        d = ((2 * h0 + h1) * m0 - h0 * m1) / (h0 + h1)
        if np.sign(d) != np.sign(m0):
            return 0.0
        if np.sign(m0) != np.sign(m1) and abs(d) > 3 * abs(m0):
            return 3.0 * m0
        return d

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        i = np.clip(np.searchsorted(self.x, q, side="right") - 1, 0, len(self.x) - 2)
        h = self.x[i + 1] - self.x[i]
        t = (q - self.x[i]) / h
        h00 = (1 + 2 * t) * (1 - t) ** 2
        h10 = t * (1 - t) ** 2
        h01 = t * t * (3 - 2 * t)
        h11 = t * t * (t - 1)
        return h00 * self.y[i] + h10 * h * self.d[i] + h01 * self.y[i + 1] + h11 * h * self.d[i + 1]


class ISO226:
    """Equal-loudness contour for one phon level: ``ISO226(phon)(freqs_hz) -> SPL dB`` (iso.py:34-173)."""

    reference = {"frequencies": FREQUENCIES, "alpha": ALPHA, "l_u": L_U, "t_f": T_F}

    def __init__(self, phon):
        if phon < 0 or phon > 90:
            raise ValueError("Phon must be in range [0, 90]")                       # iso.py:97-98
        self._phon = phon
        f = np.array(FREQUENCIES + (20000.0,))
        # iso.py:113-124: the 20 kHz knot repeats the 20 Hz value
        self._alpha = _Pchip(f, np.array(ALPHA + (ALPHA[0],)))
        self._lu = _Pchip(f, np.array(L_U + (L_U[0],)))
        self._tf = _Pchip(f, np.array(T_F + (T_F[0],)))

    def __call__(self, frequencies):
        fr = np.asarray(frequencies)
        if np.any(fr < 20.0) or np.any(fr > 20000.0):
            raise ValueError("Frequency must be in [20, 20000] Hz")                 # iso.py:152-153
        f64 = fr.astype(np.float64)
        alpha, lu, tf = self._alpha(f64), self._lu(f64), self._tf(f64)
        a = 0.00447 * ((10.0 ** (0.025 * self._phon)) - 1.15)
        b = (0.4 * (10.0 ** (((tf + lu) / 10.0) - 9.0))) ** alpha
        out = ((10.0 / alpha) * np.log10(a + b)) - lu + 94.0
        # iso.py:157 allocates the result with the input's dtype (float32 in build.py:331-340)
        return out.astype(fr.dtype if fr.dtype.kind == "f" else np.float64)


def compute_iso226_weight_matrix():
    """iso.py:176-199 -> (freqs[30], phons[10], spl[10, 30])."""
    phons = np.arange(0, 100, 10)
    freqs = np.array(FREQUENCIES + (20000.0,))
    spl = np.array([ISO226(float(p))(freqs) for p in phons])
    return freqs, phons, spl


def perceptual_weight(spl_matrix):
    """iso.py:202-235."""
    return np.clip((1 - (spl_matrix / spl_matrix.max())) ** 2, 0, 1)


class WeightTable:
    """What ``build_weight_interpolator`` returns here: the (phon x freq) grid plus its bilinear
    rule (bounds_error=False, fill_value=1.0 — iso.py:261-266)."""

    def __init__(self, phons, freqs, weights):
        self.phons = np.asarray(phons, dtype=np.float64)
        self.freqs = np.asarray(freqs, dtype=np.float64)
        self.weights = np.asarray(weights, dtype=np.float64)

    def for_bins(self, n_fft: int, sr: int) -> np.ndarray:
        """[10][n_fft/2+1] float64: grid lerped along frequency to each rfft bin; -1 marks bins outside
        [20 Hz, 20 kHz] (the interpolator's fill region).  The kernel then lerps along phon."""
        f = np.arange(n_fft // 2 + 1, dtype=np.float64) * (float(sr) / n_fft)
        oob = (f < self.freqs[0]) | (f > self.freqs[-1])
        fs = np.where(oob, self.freqs[0], f)
        j = np.clip(np.searchsorted(self.freqs, fs) - 1, 0, len(self.freqs) - 2)
        yf = (fs - self.freqs[j]) / (self.freqs[j + 1] - self.freqs[j])
        tab = self.weights[:, j] * (1 - yf) + self.weights[:, j + 1] * yf
        tab[:, oob] = -1.0
        return np.ascontiguousarray(tab)

    def __call__(self, points):
        """Host evaluation at [[phon, freq], ...] (table probes / diagnostics; the projection itself
        evaluates in the HIP kernel)."""
        pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
        s, f = pts[:, 0], pts[:, 1]
        oob = (s < self.phons[0]) | (s > self.phons[-1]) | (f < self.freqs[0]) | (f > self.freqs[-1])
        ss, ff = np.where(oob, self.phons[0], s), np.where(oob, self.freqs[0], f)
        i = np.clip(np.searchsorted(self.phons, ss) - 1, 0, len(self.phons) - 2)
        j = np.clip(np.searchsorted(self.freqs, ff) - 1, 0, len(self.freqs) - 2)
        ys = (ss - self.phons[i]) / (self.phons[i + 1] - self.phons[i])
        yf = (ff - self.freqs[j]) / (self.freqs[j + 1] - self.freqs[j])
        w = self.weights
        out = (w[i, j] * (1 - ys) * (1 - yf) + w[i, j + 1] * (1 - ys) * yf
               + w[i + 1, j] * ys * (1 - yf) + w[i + 1, j + 1] * ys * yf)
        return np.where(oob, 1.0, out)


def build_weight_interpolator() -> WeightTable:
    """iso.py:238-266."""
    freqs, phons, spl = compute_iso226_weight_matrix()
    return WeightTable(phons, freqs, perceptual_weight(spl))


def phon_threshold(max_phon_level: float, n_fft: int = 1024, sr: int = 16000) -> np.ndarray:
    """(F,) float32 — training_utils/build.py:325-348: ISO226(phon)(clip(rfftfreq, 20, 20000)), where
    the rfft frequencies are float32 (torch.fft.rfftfreq) and so is the result."""
    freqs = (np.arange(n_fft // 2 + 1, dtype=np.float32) * np.float32(1.0 / (n_fft * (1.0 / sr)))).astype(np.float32)
    return ISO226(float(max_phon_level))(np.clip(freqs, np.float32(20.0), np.float32(20000.0))).astype(np.float32)
