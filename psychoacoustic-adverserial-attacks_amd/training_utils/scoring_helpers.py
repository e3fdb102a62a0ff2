"""Which of two scores is "better" depends on the attack: a targeted run wants the perturbed WER (against the
target string) DOWN, an untargeted run wants the perturbed CTC loss UP.  ``Objective`` folds that into one signed
comparison; the three names the reference's runner imports from its ``training_utils/scoring_helpers.py``
(``Scores``, ``_is_better``, ``_best_agg``) are kept as thin views of it so that callers written against the reference
keep working."""
from __future__ import annotations

import math
from typing import Iterable, NamedTuple


class Scores(NamedTuple):
    """(ctc, wer) pair returned by ``evaluation.evaluate``."""
    ctc: float
    wer: float


class Objective:
    """sense = +1: larger is better (untargeted), -1: smaller is better (targeted)."""
    _SENSE = {"untargeted": +1.0, "targeted": -1.0}

    def __init__(self, attack_mode: str):
        if attack_mode not in self._SENSE:
            raise ValueError(f"Unknown attack_mode: {attack_mode!r}")
        self.sense = self._SENSE[attack_mode]

    @property
    def worst(self) -> float:
        """The value every real score beats: -inf when maximising, +inf when minimising."""
        return -self.sense * math.inf

    def improves(self, candidate: float, incumbent: float) -> bool:
        """Strictly better; NaN never improves and ties do not count."""
        return self.sense * candidate > self.sense * incumbent

    def best(self, history: Iterable[float]) -> float:
        return max(history, key=lambda v: self.sense * v, default=self.worst)


def _is_better(curr: float, best: float, mode: str) -> bool:
    return Objective(mode).improves(curr, best)


def _best_agg(values, mode: str) -> float:
    return Objective(mode).best(values)
