"""Evaluation loop — reference ``src/training_utils/evaluation.py:5-31`` on the HIP forward path.
Note the reference adds ``p`` WITHOUT clamping here (evaluation.py:16), unlike the training step."""
from __future__ import annotations

import torch

from ..core import loss_helpers
from .scoring_helpers import Scores


def evaluate(args, eval_data_loader, p, model, processor, wer_metric, perturbed=False, epoch_number=-1) -> Scores:
    ctc_scores, wer_scores = [], []
    pp = None
    if perturbed and isinstance(p, torch.Tensor):
        pp = p.detach().to(model.device, torch.float32).reshape(1, -1).contiguous()
    for data, target_texts in eval_data_loader:
        data = data.to(args.device, torch.float32).contiguous()
        labels = loss_helpers.make_labels(target_texts, processor, args, len(data))
        r = model.forward(data, pp, labels, clamp=False)
        ctc_scores.append(float(r["loss"].item()))
        wer_scores.append(float(loss_helpers.compute_wer(r["logits"], target_texts, processor, wer_metric)))
    avg_ctc = sum(ctc_scores) / len(ctc_scores) if ctc_scores else float("inf")
    avg_wer = sum(wer_scores) / len(wer_scores) if wer_scores else float("inf")
    return Scores(ctc=avg_ctc, wer=avg_wer)
