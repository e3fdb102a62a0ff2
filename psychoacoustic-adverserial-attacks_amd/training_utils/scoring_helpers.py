"""Best-score bookkeeping — reference ``src/training_utils/scoring_helpers.py`` (with its missing import fixed)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class Scores:
    ctc: float
    wer: float


def _is_better(curr: float, best: float, mode: str) -> bool:
    """Targeted: lower perturbed WER is better; untargeted: higher perturbed CTC loss is better (scoring_helpers.py:6-17)."""
    if mode == "targeted":
        return curr < best
    if mode == "untargeted":
        return curr > best
    raise ValueError(f"Unknown attack_mode: {mode!r}")


def _best_agg(values, mode: str) -> float:
    """Min for targeted, max for untargeted (scoring_helpers.py:19-23)."""
    if not values:
        return float("inf") if mode == "targeted" else float("-inf")
    return (min if mode == "targeted" else max)(values)
