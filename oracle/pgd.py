"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of one PGD step of the reference's hot loop
(/root/reference/src/training_utils/train.py:103-164) with its label generation
(/root/reference/src/core/loss_helpers.py:7-23) and WER bookkeeping (loss_helpers.py:25-32).
Pinned by tests/golden/pgd_*.npz and tests/golden/labels.json (oracle/gen_goldens.py).

Third-party pieces restated: the HF ``Wav2Vec2CTCTokenizer`` character tokenizer / greedy CTC
decode (transformers 5.15.0 here, unpinned upstream) and ``jiwer`` word error rate (absent in
this image; restated as corpus-level word edit distance, which is jiwer's definition of WER).
"""
from __future__ import annotations

import re

import torch

from . import projections as P
from . import wav2vec2 as W

# The 32-token vocabulary of the 960h Wav2Vec2 CTC checkpoints (recalled, not on disk).
VOCAB = ["<pad>", "<s>", "</s>", "<unk>", "|", "E", "T", "A", "O", "N", "I", "H", "S", "R", "D", "L", "U", "M",
         "W", "C", "F", "G", "Y", "P", "B", "V", "K", "'", "X", "J", "Q", "Z"]
TOK2ID = {t: i for i, t in enumerate(VOCAB)}


def clean_transcripts(texts):
    """loss_helpers.py:7-9."""
    return [re.sub(r"\s+", " ", t.replace("<unk>", "").lower()).strip() for t in texts]


def tokenize(text: str) -> list:
    """Wav2Vec2CTCTokenizer with do_lower_case=False: spaces -> '|', one id per character,
    out-of-vocabulary characters (every lower-case letter) -> <unk>=3."""
    return [TOK2ID.get(ch, 3) for ch in text.replace(" ", "|")]


def make_labels(texts, args, batch_size) -> torch.Tensor:
    """loss_helpers.py:13-20: targeted override, clean, tokenize, pad with 0, then 0 -> -100."""
    if args.attack_mode == "targeted":
        texts = [" ".join([args.target] * args.target_reps)] * batch_size
    ids = [tokenize(t) for t in clean_transcripts(texts)]
    smax = max(len(i) for i in ids)
    lab = torch.zeros(len(ids), smax, dtype=torch.long)
    for r, i in enumerate(ids):
        lab[r, :len(i)] = torch.tensor(i, dtype=torch.long)
    lab[lab == 0] = -100
    return lab


def greedy_decode(logits: torch.Tensor) -> list:
    """loss_helpers.py:26-28: argmax -> ``processor.batch_decode(ids, skip_special_tokens=True)`` ->
    strip, lower.  As pinned by tests/golden/labels.json (transformers 5.15.0): the special ids
    (<pad>, <s>, </s>, <unk>) are dropped FIRST and repeats are collapsed afterwards, so a blank
    between two equal letters does not keep them apart ("hel<pad>lo" decodes to "helo")."""
    out = []
    for row in torch.argmax(logits, dim=-1).tolist():
        toks, prev = [], None
        for i in row:
            if i <= 3:
                continue
            if i != prev:
                toks.append(i)
            prev = i
        s = "".join(" " if VOCAB[i] == "|" else VOCAB[i] for i in toks)
        out.append(s.strip().lower())
    return out


def _edit_distance(a, b) -> int:
    d = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        prev, d[0] = d[0], i
        for j in range(1, len(b) + 1):
            cur = min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
            prev, d[j] = d[j], cur
    return d[len(b)]


def wer(pred_texts, ref_texts):
    """Corpus-level WER over the batch: sum(edit distance over words) / sum(reference words)."""
    errs = sum(_edit_distance(r.split(), p.split()) for p, r in zip(pred_texts, ref_texts))
    words = sum(len(r.split()) for r in ref_texts)
    return errs / max(words, 1), errs, words


def compute_wer(logits, target_texts):
    """loss_helpers.py:25-32."""
    refs = [t.lower() for t in clean_transcripts(target_texts)]
    return wer(greedy_decode(logits), refs)[0]


def pgd_step(sd, arch, args, clean, labels, p, spl_thresh=None):
    """train.py:124-164 for one batch -> dict(loss, logits, grad, p_new).

    ``labels`` is the (B, S) int64 tensor with -100 padding (loss_helpers.py:19-20)."""
    direction = +1 if args.attack_mode == "untargeted" else -1          # train.py:124
    p = p.detach().clone().requires_grad_(True)                          # train.py:133
    perturbed = (clean + p).clamp(-1.0, 1.0)                             # train.py:136
    loss, logits = W.forward(sd, arch, perturbed, labels)                # loss_helpers.py:21
    (direction * loss).backward()                                        # train.py:158
    grad = p.grad.detach().clone()
    with torch.no_grad():
        p_new = p.detach() + args.lr * grad.sign()                       # train.py:161
        p_new = P.perturbation_constraint(p_new, clean, args, spl_thresh)  # train.py:162
    return dict(loss=loss.detach(), logits=logits.detach(), grad=grad, p_new=p_new.detach())


def train_epoch(sd, arch, args, loader, p, optimizer=None, spl_thresh=None):
    """train.py:103-182 over a loader of (clean (B, L), texts): the PGD branch (:156-164) or the Adam branch (:165-175:
    zero_grad, (-direction * loss).backward(), optimizer.step(), p.data = constraint(p.data)).  Returns
    (p, mean of the per-batch sum-reduced CTC losses, mean of the per-batch WERs vs the ground truth)."""
    direction = +1 if args.attack_mode == "untargeted" else -1
    ctc, wers = [], []
    for clean, texts in loader:
        labels = make_labels(texts, args, len(clean))
        if args.optimizer_type == "pgd":
            r = pgd_step(sd, arch, args, clean, labels, p, spl_thresh)
            p = r["p_new"]
            loss, logits = r["loss"], r["logits"]
        else:
            optimizer.zero_grad(set_to_none=True)
            perturbed = (clean + p).clamp(-1.0, 1.0)
            loss, logits = W.forward(sd, arch, perturbed, labels)
            (-direction * loss).backward()
            optimizer.step()
            with torch.no_grad():
                p.data = P.perturbation_constraint(p.data, clean, args, spl_thresh)
            loss, logits = loss.detach(), logits.detach()
        ctc.append(float(loss))
        wers.append(compute_wer(logits, texts))
    return p, sum(ctc) / len(ctc), sum(wers) / len(wers)
