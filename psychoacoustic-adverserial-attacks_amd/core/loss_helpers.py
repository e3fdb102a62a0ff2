"""Label generation, loss interface and WER bookkeeping around the HIP model — call surface of the
reference's ``src/core/loss_helpers.py``.  The tokenizer / greedy decoder / WER are small host-side
string routines (the reference delegates them to HF's ``Wav2Vec2Processor`` and ``jiwer``); when an
HF processor object is passed it is used as-is, otherwise the built-in 32-token character vocabulary
of the 960h checkpoints is used."""
from __future__ import annotations

import re

import torch

# SURVEY A.3 — vocabulary of facebook/wav2vec2-*-960h* (recalled; confirm against a local vocab.json).
VOCAB = ["<pad>", "<s>", "</s>", "<unk>", "|", "E", "T", "A", "O", "N", "I", "H", "S", "R", "D", "L", "U", "M",
         "W", "C", "F", "G", "Y", "P", "B", "V", "K", "'", "X", "J", "Q", "Z"]
_TOK2ID = {t: i for i, t in enumerate(VOCAB)}
PAD_ID, UNK_ID = 0, 3


def clean_transcripts(texts):
    """loss_helpers.py:7-9."""
    return [re.sub(r"\s+", " ", t.replace("<unk>", "").lower()).strip() for t in texts]


def tokenize(text: str):
    """Character CTC tokenizer with do_lower_case=False: ' ' -> '|', unknown characters -> <unk>."""
    return [_TOK2ID.get(ch, UNK_ID) for ch in text.replace(" ", "|")]


def make_labels(target_texts, processor, args, batch_size: int) -> torch.Tensor:
    """loss_helpers.py:13-20 -> (B, S_max) int64 on the CPU with -100 padding."""
    if args.attack_mode == "targeted":
        target_texts = [" ".join([args.target] * args.target_reps)] * batch_size
    texts = clean_transcripts(target_texts)
    if processor is not None:
        labels = processor(text=texts, return_tensors="pt", padding=True).input_ids
        labels[labels == processor.tokenizer.pad_token_id] = -100
        return labels
    ids = [tokenize(t) for t in texts]
    smax = max(1, max(len(i) for i in ids))
    labels = torch.zeros(len(ids), smax, dtype=torch.long)
    for r, i in enumerate(ids):
        if i:
            labels[r, :len(i)] = torch.tensor(i, dtype=torch.long)
    labels[labels == PAD_ID] = -100
    return labels


def get_loss_for_training(model, data, target_texts, processor, args):
    """loss_helpers.py:12-23: ``data`` is the already composed (perturbed) batch; returns (loss, logits)
    with HF's ctc_loss_reduction='sum'.  Forward only — the PGD step gets its gradient from
    ``training_utils.pgd.PgdStepper`` in the same launch sequence."""
    labels = make_labels(target_texts, processor, args, len(data))
    r = model.forward(data, None, labels)
    return r["loss"], r["logits"]


def get_loss(batch_waveforms, target_texts, processor, args, model):
    """loss_helpers.py:46-57 — the evaluation twin of ``get_loss_for_training`` with the reference's argument
    order ``(batch_waveforms, target_texts, processor, args, model)``; same computation."""
    return get_loss_for_training(model, batch_waveforms, target_texts, processor, args)


def get_logits(batch_waveforms, processor, args, model):
    """loss_helpers.py:34-43: the only place the reference applies the processor's feature extractor, i.e.
    ``Wav2Vec2FeatureExtractor(do_normalize=True)``: per-utterance zero mean / unit variance,
    ``(x - mean) / sqrt(var + 1e-7)`` with the population variance (HF feature_extraction_wav2vec2.py
    ``zero_mean_unit_var_norm``).  A processor object, when given, is used as is (host round trip, as the
    reference does); otherwise the same normalisation runs on the device."""
    if processor is not None:
        inputs = processor(batch_waveforms.cpu().tolist(), sampling_rate=args.sr, return_tensors="pt", padding=True)
        x = inputs.input_values.to(model.device, torch.float32)
    else:
        x = batch_waveforms.to(model.device, torch.float32)
        x = (x - x.mean(dim=-1, keepdim=True)) / torch.sqrt(x.var(dim=-1, keepdim=True, unbiased=False) + 1e-7)
    return model.forward(x.contiguous(), None, None)["logits"]


def argmax_ids(logits: torch.Tensor) -> torch.Tensor:
    """``torch.argmax(logits, dim=-1)`` (loss_helpers.py:26,61) through the C ABI: (..., V) f32 cuda -> (...) int16."""
    from .. import _lib, runtime
    x = runtime.as_f32_cuda(logits, "logits")
    ids = torch.empty(x.shape[:-1], dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().paa_argmax_ids(_lib.ptr(x), ids.numel(), x.shape[-1], _lib.ptr(ids), _lib.stream_ptr()))
    return ids


def greedy_decode_ids(pred_ids) -> list:
    """ids -> text as ``Wav2Vec2CTCTokenizer.batch_decode(skip_special_tokens=True)`` of transformers 5.15.0
    does it (pinned by tests/golden/labels.json): special ids dropped first, repeats collapsed after."""
    out = []
    for row in pred_ids:
        toks, prev = [], None
        for i in row:
            if i <= UNK_ID:
                continue
            if i != prev:
                toks.append(i)
            prev = i
        out.append("".join(" " if VOCAB[i] == "|" else VOCAB[i] for i in toks).strip())
    return out


def _edit_distance(a, b) -> int:
    d = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        prev, d[0] = d[0], i
        for j in range(1, len(b) + 1):
            cur = min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
            prev, d[j] = d[j], cur
    return d[len(b)]


def wer_counts(pred_texts, ref_texts):
    """(word errors, reference words) — corpus-level WER = errors / words, jiwer's definition."""
    errs = sum(_edit_distance(r.split(), p.split()) for p, r in zip(pred_texts, ref_texts))
    return errs, sum(len(r.split()) for r in ref_texts)


def wer_texts(logits, target_texts, processor):
    """The two string lists loss_helpers.py:26-30 hands to the WER metric: greedy CTC decode of ``logits`` and the cleaned
    references, both lower-cased."""
    pred_ids = argmax_ids(logits)
    if processor is not None:
        pred_texts = processor.batch_decode(pred_ids.long().cpu(), skip_special_tokens=True)
    else:
        pred_texts = greedy_decode_ids(pred_ids.tolist())
    return [p.strip().lower() for p in pred_texts], [t.lower() for t in clean_transcripts(target_texts)]


def compute_wer(logits, target_texts, processor, wer_metric):
    """loss_helpers.py:25-32."""
    pred_texts, ref_texts = wer_texts(logits, target_texts, processor)
    if wer_metric is not None:
        return wer_metric.compute(predictions=pred_texts, references=ref_texts)
    e, w = wer_counts(pred_texts, ref_texts)
    return e / max(w, 1)


def decode(logits, processor):
    """loss_helpers.py:60-62."""
    pred_ids = argmax_ids(logits)
    if processor is not None:
        return processor.batch_decode(pred_ids.long().cpu())
    return greedy_decode_ids(pred_ids.tolist())
