"""Same call surface as the reference's ``src/training_utils/train.py``:
``perturbation_constraint(p, clean_audio, args, interp, spl_thresh)`` and
``train_epoch(args, train_data_loader, p, model, epoch, processor, interp, wer_metric, spl_thresh, optimizer)``
— executed by libpaa_hip.so."""
from __future__ import annotations

import logging
import time
from dataclasses import dataclass
from typing import Iterable

import torch

from .. import _lib, runtime
from ..core import loss_helpers
from .pgd import FREQ_NORMS, PgdStepper

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class TrainEpochResult:          # train.py:15-19
    p: torch.Tensor
    avg_ctc: float
    avg_wer: float


def _avg(values: Iterable[float]) -> float:
    vals = list(values)
    return sum(vals) / max(len(vals), 1)


def global_wer_per_step(counts, device, group=None):
    """Data-parallel runs: ``counts`` = this rank's [(word errors, reference words)] per step.  The WER the reference would
    report for the GLOBAL batch of a step is sum_r errors / sum_r words (jiwer is corpus-level within a batch), so the
    per-step counts of all ranks are summed by ONE small all-reduce at the end of the epoch (every rank runs the same
    number of steps — the step's own all-reduce already requires that)."""
    import torch.distributed as dist
    t = torch.tensor(counts, dtype=torch.float64, device=device).reshape(-1, 2)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t = t.cpu()
    return [float(e) / max(float(w), 1.0) for e, w in t.tolist()]


def perturbation_constraint(p: torch.Tensor, clean_audio, args, interp, spl_thresh) -> torch.Tensor:
    """train.py:69-99.  Returns a new tensor; ``p`` is left untouched.  ``args.norm_type`` may be a
    '+'-joined list (extension, applied in the written order)."""
    norms = str(args.norm_type).split("+")
    for n in norms:
        if n not in _lib.NORM_IDS:
            raise ValueError(f"Unknown norm_type: {n!r}")                                   # train.py:98
        if n == "snr" and clean_audio is None:
            raise ValueError("SNR projection requires clean_audio ro compare to")           # train.py:91
        if n == "tv" and clean_audio is None:
            raise ValueError("TV projection can benefit from clean_audio for bounds")       # train.py:95
    src = runtime.as_f32_cuda(p.detach(), "p")
    q = torch.empty_like(src)
    rows, L = (q.shape[0], q.shape[1]) if q.dim() == 2 else (1, q.shape[0])
    clean = None if clean_audio is None else runtime.as_f32_cuda(clean_audio, "clean_audio")
    if clean is not None and clean.shape[-1] != L:
        raise ValueError(f"clean_audio length {clean.shape[-1]} != perturbation length {L}")
    pr = runtime.get_proj(args, q.device, rows, L, interp)
    out_len = L
    with torch.cuda.device(q.device):
        for i, n in enumerate(norms):
            a = type("A", (), dict(vars(args)))()
            a.norm_type = n
            if n == "max_phon":
                pr.set_spl_thresh(spl_thresh)
            nb = 0 if clean is None else clean.shape[0]
            if i == 0:       # the reference's functions return a new tensor: out of place from p (one fused launch for the FFT norms)
                _lib.check(_lib.lib().paa_project_to(pr.h, runtime.params_of(a), _lib.ptr(src), _lib.ptr(q), rows, _lib.ptr(clean),
                                                     nb, L, _lib.stream_ptr()))
            else:
                _lib.check(_lib.lib().paa_project(pr.h, runtime.params_of(a), _lib.ptr(q), rows, _lib.ptr(clean), nb, L,
                                                  _lib.stream_ptr()))
            if n in FREQ_NORMS and clean is None:
                out_len = pr.hop * (L // pr.hop)        # iSTFT length hop*(T-1); no _align_to without clean audio
    return q if out_len == L else q[..., :out_len]


def train_epoch(args, train_data_loader, p: torch.Tensor, model, epoch: int, processor, interp, wer_metric,
                spl_thresh, optimizer) -> TrainEpochResult:
    """train.py:103-182.  ``model`` is a ``paa_amd.model.PaaModel``."""
    ctc_scores, wer_scores, wer_counts, times = [], [], [], []
    logger.info("starting epoch: %d", epoch)
    if args.optimizer_type not in ("pgd", "adam"):
        raise NotImplementedError(f"Optimization type not implemented: {args.optimizer_type!r}")   # train.py:177
    if args.optimizer_type == "adam" and optimizer is None:
        raise ValueError("Adam optimizer selected but optimizer is None")                          # train.py:167
    L = p.shape[-1]
    stepper = getattr(model, "_stepper", None)
    if stepper is None or stepper.args is not args or stepper.L != L:
        stepper = PgdStepper(model, args, L, interp, spl_thresh)
        model._stepper = stepper
    if args.optimizer_type == "adam" and stepper.world > 1:
        raise NotImplementedError("the Adam branch runs on one GPU; the data-parallel step is the PGD branch (SURVEY 8e)")
    for clean_audio, target_texts in train_data_loader:
        t0 = time.perf_counter()
        clean_audio = clean_audio.to(args.device, torch.float32, non_blocking=True).contiguous()   # train.py:129
        labels = loss_helpers.make_labels(target_texts, processor, args, len(clean_audio))
        if args.optimizer_type == "pgd":
            if isinstance(p, torch.nn.Parameter) or p.requires_grad:
                p = p.detach()
            r = stepper.step(p, clean_audio, labels)
        else:
            if p.dtype != torch.float32 or not p.is_cuda:
                raise TypeError(f"the Adam branch needs a float32 perturbation on the GPU, got {p.dtype} on {p.device}")
            r = model.fwd_bwd(clean_audio, p.data, labels, stepper.direction)
            optimizer.zero_grad(set_to_none=True)
            p.grad = -r["grad"].view_as(p)          # gradient of (-direction * loss), train.py:170
            optimizer.step()
            with torch.no_grad():
                p.data = perturbation_constraint(p.data, clean_audio, args, interp, spl_thresh)   # train.py:172-175
        ctc_scores.append(float(r["loss"].item()))                                   # train.py:146
        pred_texts, ref_texts = loss_helpers.wer_texts(r["logits"], target_texts, processor)       # train.py:149-153
        e, w = loss_helpers.wer_counts(pred_texts, ref_texts)
        wer_counts.append((e, w))
        stepper.set_wer_counts(e, w)          # rides behind the next step's gradient (stats[3:5] = sums over ranks)
        wer_scores.append(float(wer_metric.compute(predictions=pred_texts, references=ref_texts)) if wer_metric is not None
                          else e / max(w, 1))
        times.append(time.perf_counter() - t0)
    if stepper.world > 1:                     # the loss in stats[0] is already the global batch's; make the WER global too
        wer_scores = global_wer_per_step(wer_counts, stepper.dev, stepper.group)
    return TrainEpochResult(p=p, avg_ctc=_avg(ctc_scores), avg_wer=_avg(wer_scores))
