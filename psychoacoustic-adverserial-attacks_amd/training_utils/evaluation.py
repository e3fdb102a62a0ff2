"""Evaluation loop — reference ``src/training_utils/evaluation.py:5-31`` on the HIP forward path.
Note the reference adds ``p`` WITHOUT clamping here (evaluation.py:16), unlike the training step."""
from __future__ import annotations

import torch

from ..core import loss_helpers
from .scoring_helpers import Scores


def evaluate(args, eval_data_loader, p, model, processor, wer_metric, perturbed=False, epoch_number=-1) -> Scores:
    ctc_scores, wer_scores, counts = [], [], []
    pp = None
    if perturbed and isinstance(p, torch.Tensor):
        pp = p.detach().to(model.device, torch.float32).reshape(1, -1).contiguous()
    for data, target_texts in eval_data_loader:
        if len(target_texts) == 0:           # an empty shard of a short global batch (build.shard_batches): zeros for the all-reduce
            ctc_scores.append(0.0); wer_scores.append(0.0); counts.append((0, 0))
            continue
        data = data.to(args.device, torch.float32).contiguous()
        labels = loss_helpers.make_labels(target_texts, processor, args, len(data))
        r = model.forward(data, pp, labels, clamp=False)
        ctc_scores.append(float(r["loss"].item()))
        pred_texts, ref_texts = loss_helpers.wer_texts(r["logits"], target_texts, processor)
        e, w = loss_helpers.wer_counts(pred_texts, ref_texts)
        counts.append((e, w))
        wer_scores.append(float(wer_metric.compute(predictions=pred_texts, references=ref_texts)) if wer_metric is not None
                          else e / max(w, 1))
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        # sharded evaluation: CTC loss is a sum over clips and WER is corpus-level within a batch, so the global batch's
        # scores are the sums over ranks (one small all-reduce per evaluation)
        from .train import global_wer_per_step
        t = torch.tensor(ctc_scores, dtype=torch.float64, device=model.device)
        torch.distributed.all_reduce(t)
        ctc_scores = t.cpu().tolist()
        wer_scores = global_wer_per_step(counts, model.device)
    avg_ctc = sum(ctc_scores) / len(ctc_scores) if ctc_scores else float("inf")
    avg_wer = sum(wer_scores) / len(wer_scores) if wer_scores else float("inf")
    return Scores(ctc=avg_ctc, wer=avg_wer)
