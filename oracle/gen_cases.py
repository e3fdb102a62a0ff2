"""ORACLE tooling — the case list shared by oracle/gen_goldens.py (which runs the reference) and
the tests (which never do)."""
from __future__ import annotations

from paa_amd import arch as A

from . import projections as OP

# (norm_type, extra CLI flags) — reference defaults plus the BASELINE.json epsilons.
NORM_CASES = [
    ("l2", []), ("linf", []), ("snr", ["--snr_db", "40"]), ("snr", []), ("tv", []),
    ("min_max_freqs", []), ("fletcher_munson", ["--fm_epsilon", "2.0"]),
    ("fletcher_munson", ["--fm_epsilon", "0.05"]), ("max_phon", []), ("max_phon", ["--max_phon_level", "25"]),
]
LENGTHS = [4096, 5000]
AMPS = [1e-4, 1e-2, 0.3]

PGD_CASES = [
    # name, arch, L, B, norm, extra flags
    ("tiny_group_snr", A.tiny("group", False), 8000, 3, "snr", ["--snr_db", "40"]),
    ("tiny_group_maxphon", A.tiny("group", False), 8000, 2, "max_phon", []),
    ("tiny_layer_stable_fm", A.tiny("layer", True), 8000, 2, "fletcher_munson", ["--fm_epsilon", "0.05"]),
    ("tiny_group_targeted_linf", A.tiny("group", False), 8000, 2, "linf",
     ["--attack_mode", "targeted", "--target", "ab", "--target_reps", "2"]),
    ("base_snr", A.BASE, 16000, 2, "snr", ["--snr_db", "40"]),
]
PGD_TEXTS = ["ab cd", "hello", "a b c", "xyz w"]

_FLOAT = {"snr_db", "fm_epsilon", "max_phon_level", "l2_size", "linf_size", "tv_epsilon", "lr",
          "min_freq_attack", "max_freq_attack", "phon_reference_db"}
_INT = {"target_reps", "n_fft", "hop_length", "win_length", "sr"}


def case_name(norm, extra, L, B, amp):
    tag = "_".join(x.strip("-") for x in extra) or "default"
    return f"{norm}|{tag}|L{L}|B{B}|a{amp:g}"


def cli_to_args(norm, extra=()):
    """['--snr_db', '40'] -> oracle args namespace (same defaults as the reference parser)."""
    kw = {"norm_type": norm}
    it = iter(extra)
    for flag in it:
        key = flag.lstrip("-")
        val = next(it)
        kw[key] = float(val) if key in _FLOAT else int(val) if key in _INT else val
    return OP.default_args(**kw)
