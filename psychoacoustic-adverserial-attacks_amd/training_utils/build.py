"""Setup-time pieces of the reference's ``src/training_utils/build.py`` that the hot path needs:
``init_phon_threshold_tensor``, ``init_perturbation``, ``load_model`` (local path / rule weights —
the reference fetches by name, build.py:229-230, which is impossible offline) and ``create_optimizer``."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import arch as A
from .. import synth
from ..core import iso
from ..model import PaaModel
from .train import perturbation_constraint


def init_phon_threshold_tensor(args):
    """build.py:325-348 -> (1, F, 1) float32 on args.device."""
    thr = iso.phon_threshold(float(args.max_phon_level), int(args.n_fft), int(args.sr))
    return torch.from_numpy(thr).to(args.device).view(1, -1, 1)


def init_perturbation(args, length, spl_thresh, interp, first_batch_data):
    """build.py:288-321.  Fresh init: N(0,1) from the counter-based generator (seed args.seed), then one
    projection.  Resume: a (1, L) float32 tensor saved with torch.save (loaded with weights_only=True)."""
    ckpt = getattr(args, "resume_from", None)
    if ckpt and os.path.isfile(ckpt):
        p = torch.load(ckpt, map_location="cpu", weights_only=True).detach().to(args.device, torch.float32)
    else:
        p = torch.from_numpy(synth.perturbation(length, seed=int(getattr(args, "seed", 5)))).to(args.device)
        p = perturbation_constraint(p=p, clean_audio=first_batch_data, args=args, interp=interp,
                                    spl_thresh=spl_thresh).detach()
    if p.shape[-1] != length:
        raise ValueError(f"Loaded perturbation length {p.shape[-1]} != expected {length}")     # build.py:315
    if args.optimizer_type == "adam":
        p = torch.nn.Parameter(p)
    elif args.optimizer_type != "pgd":
        raise NotImplementedError(f"Unsupported optimizer_type: {args.optimizer_type}")         # build.py:312
    return p


ARCHS = {"base": A.BASE, "large-lv60": A.LARGE_LV60, "tiny": A.tiny()}


def load_model(args, max_batch: int, length: int):
    """Local HF checkpoint directory (config.json + safetensors / bin) if --model_path, else rule weights."""
    path = getattr(args, "model_path", None)
    if path:
        from transformers import Wav2Vec2ForCTC, Wav2Vec2Processor
        hf = Wav2Vec2ForCTC.from_pretrained(path, local_files_only=True).eval()
        try:
            proc = Wav2Vec2Processor.from_pretrained(path, local_files_only=True)
        except Exception:
            proc = None
        return PaaModel.from_hf(hf, max_batch, length, args.dtype, args.device), proc
    a = ARCHS[getattr(args, "arch", "base")]
    return PaaModel(a, A.rule_weights(a), max_batch, length, args.dtype, args.device), None


def create_optimizer(args, p):
    """build.py:352-359."""
    optimizer = torch.optim.Adam([p], lr=args.lr)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=args.step_size, gamma=args.gamma)
    return optimizer, scheduler


def synthetic_loader(args, batch: int, length: int, steps: int, rank: int = 0, world: int = 1):
    """Fixed-length synthetic batches: ``steps`` batches of ``batch`` clips per rank,
    transcripts of lower-case words (=> <unk>/| labels)."""
    words = ["the", "quick", "brown", "fox", "jumps", "over", "a", "lazy", "dog", "and", "runs", "away"]
    out = []
    for s in range(steps):
        first = (s * world + rank) * batch
        x = torch.from_numpy(synth.clean_audio(batch, length, seed=int(args.seed), first_clip=first))
        texts = []
        for b in range(batch):
            u = synth.uniform(synth.key_of(f"txt{first + b}", int(args.seed)), 24)
            texts.append(" ".join(words[int(v * len(words))] for v in u))
        out.append((x, texts))
    return out


# ------------------------------------------------------------------------------------------------ runner setup
def attack_size_string(args) -> str:
    """build.py:235-247: the epsilon that names the run directory."""
    sizes = {"min_max_freqs": f"{args.min_freq_attack}", "fletcher_munson": f"{args.fm_epsilon}",
             "max_phon": f"{args.max_phon_level}", "l2": f"{args.l2_size}", "linf": f"{args.linf_size}",
             "snr": f"{args.snr_db}", "tv": f"{args.tv_epsilon}"}
    first = str(args.norm_type).split("+")[0]
    if first not in sizes:
        raise ValueError(f"Unsupported norm_type: {args.norm_type}")
    return sizes[first]


def create_logger(args, logs_root=None):
    """build.py:233-286: derive save_dir = <logs>/<mode>/<dataset>/<norm>_<size>_<mode>_<opt>, set up the "asr_attack"
    logger (rotating file + console) and discover a resumable checkpoint: an existing perturbation.pt in save_dir makes
    the run resume from it at results.json["epoch"] (unless --small_data), overriding --resume_from as the reference does."""
    import json
    import logging
    from logging.handlers import RotatingFileHandler
    args.attack_size_string = attack_size_string(args)
    root = logs_root or getattr(args, "logs_dir", None) or os.path.join(os.getcwd(), "logs")
    args.save_dir = os.path.join(root, args.attack_mode, args.dataset,
                                 f"{args.norm_type}_{args.attack_size_string}_{args.attack_mode}_{args.optimizer_type}")
    os.makedirs(args.save_dir, exist_ok=True)
    logger = logging.getLogger("asr_attack")
    logger.setLevel(logging.INFO)
    logger.handlers.clear()
    fmt = logging.Formatter("%(asctime)s | %(levelname)s | %(message)s")
    fh = RotatingFileHandler(os.path.join(args.save_dir, "train.log"), maxBytes=5 * 1024 * 1024, backupCount=3)
    fh.setFormatter(fmt)
    logger.addHandler(fh)
    if not getattr(args, "silent", False):
        ch = logging.StreamHandler()
        ch.setFormatter(fmt)
        logger.addHandler(ch)
    args.was_preempted = os.path.exists(os.path.join(args.save_dir, "perturbation.pt"))
    args.had_checkpoint = args.was_preempted
    chkpt_epoch = 0
    results_path = os.path.join(args.save_dir, "results.json")
    if os.path.exists(results_path):
        try:
            with open(results_path) as f:
                chkpt_epoch = int(json.load(f).get("epoch", 0))
        except Exception as e:          # noqa: BLE001
            logger.warning("Failed to read results.json: %s", e)
    if args.was_preempted and not getattr(args, "small_data", False):
        args.resume = True
        args.resume_from = os.path.join(args.save_dir, "perturbation.pt")
        logger.info("Resuming from checkpoint: %s (epoch=%d)", args.resume_from, chkpt_epoch)
    else:
        args.resume = False
    return logger, max(chkpt_epoch, 0)


def plan_dataset(lengths, seed: int, relative_audio_length: float, target_size: int = 30_000, n_splits: int = 4):
    """The selection the reference's ``create_data_loaders`` makes (build.py:104-208), as index arithmetic on the clip
    lengths alone: ``random.seed(seed)``; the clips of the ``n_splits`` dataset splits are concatenated split by split
    (clip i of the input belongs to split ``i % n_splits`` — the order a caller interleaving LibriSpeech's four splits
    produces) and shuffled; the first ``target_size`` are kept; ``min_len`` / ``audio_length`` are the 10 % and
    ``relative_audio_length`` quantiles (torch.quantile, float32, linear interpolation, truncated to int) of the FIRST
    300 of them; clips outside [min_len, audio_length] are DROPPED (build.py:189 — not cropped); a second shuffle of the
    survivors' positions gives the 80 / 10 / 10 split.  Returns dict(audio_length, min_len, train, eval, test) with
    clip indices into ``lengths``; train order is immaterial (the reference's train loader reshuffles every epoch)."""
    import random
    rnd = random.Random()
    rnd.seed(seed)
    order = [i for k in range(n_splits) for i in range(k, len(lengths), n_splits)]
    rnd.shuffle(order)
    order = order[:target_size]
    head = torch.tensor([float(lengths[i]) for i in order[:min(300, len(order))]], dtype=torch.float32)
    min_len = int(head.quantile(0.10).item())
    audio_length = int(head.quantile(float(relative_audio_length)).item())
    kept = [i for i in order if min_len <= lengths[i] <= audio_length][:target_size]
    idx = list(range(len(kept)))
    rnd.shuffle(idx)
    n_tr, n_ev = int(0.8 * len(idx)), int(0.1 * len(idx))
    pick = lambda sel: [kept[j] for j in sel]
    return dict(audio_length=audio_length, min_len=min_len, train=pick(idx[:n_tr]), eval=pick(idx[n_tr:n_tr + n_ev]),
                test=pick(idx[n_tr + n_ev:]))


def percentile_length(lengths, q: float) -> int:
    """build.py:41-61: the clip length every utterance is cropped / right-zero-padded to (q-quantile of the lengths)."""
    return int(np.quantile(np.asarray(lengths, dtype=np.int64), q))


def collate_fixed(waves, length: int) -> torch.Tensor:
    """build.py:186-191: crop or right-zero-pad each waveform to ``length`` -> (N, length) float32."""
    out = torch.zeros(len(waves), length, dtype=torch.float32)
    for i, w in enumerate(waves):
        w = torch.as_tensor(w, dtype=torch.float32).reshape(-1)
        n = min(length, w.numel())
        out[i, :n] = w[:n]
    return out


def _batches(x, texts, batch_size):
    return [(x[i:i + batch_size], texts[i:i + batch_size]) for i in range(0, len(texts), batch_size)]


def shard_batches(batches, rank: int, world: int, keep_empty: bool = False):
    """Data-parallel runs: every GLOBAL batch (x (n, L), texts) is cut into ``world`` contiguous shards whose sizes
    differ by at most one clip; rank r keeps shard r.  Ranks may therefore hold different numbers of clips in a step (the
    packed all-reduce carries the clip count, training_utils/pgd.py).  A batch with fewer clips than ranks is dropped from a
    TRAINING loader (every rank must run every step with at least one clip); an evaluation loader keeps it
    (``keep_empty``): the ranks past the last clip get an empty shard and contribute zeros to the evaluation's all-reduce."""
    if world <= 1:
        return list(batches)
    out = []
    for x, texts in batches:
        n = len(texts)
        if n < world and not keep_empty:
            continue
        lo = rank * n // world
        hi = (rank + 1) * n // world
        out.append((x[lo:hi], texts[lo:hi]))
    return out


def load_local_dataset(data_dir: str, sr: int):
    """A local directory of ``*.wav`` files with transcripts in ``*.trans.txt`` (LibriSpeech style: "<utt-id> TEXT")
    or ``transcripts.csv`` ("file,text").  No network, no torchaudio: PCM wav via the stdlib."""
    import csv
    import glob
    from . import save
    texts = {}
    for f in glob.glob(os.path.join(data_dir, "**", "*.trans.txt"), recursive=True):
        for line in open(f):
            k, _, t = line.strip().partition(" ")
            texts[k] = t
    csvp = os.path.join(data_dir, "transcripts.csv")
    if os.path.exists(csvp):
        for row in csv.reader(open(csvp)):
            if len(row) >= 2:
                texts[os.path.splitext(os.path.basename(row[0]))[0]] = row[1]
    waves, out_texts = [], []
    for f in sorted(glob.glob(os.path.join(data_dir, "**", "*.wav"), recursive=True)):
        key = os.path.splitext(os.path.basename(f))[0]
        if key not in texts:
            continue
        x, fsr = save.load_audio(f)
        if fsr != sr:
            raise ValueError(f"{f}: sample rate {fsr} != --sr {sr}")
        waves.append(x)
        out_texts.append(texts[key])
    if not waves:
        raise ValueError(f"no (wav, transcript) pairs found under {data_dir}")
    return waves, out_texts


def create_data_loaders(args, rank: int = 0, world: int = 1):
    """build.py:104-220 without the network: --data_dir (local wavs) or synthetic clips; fixed-length collate at the
    ``relative_audio_length`` quantile; 80/10/10 split; --small_data keeps ~1 % (at least 3 batches' worth).
    Returns (train, eval, test) lists of (batch (B, L) float32 CPU tensor, list[str]) and the clip length.
    ``world`` > 1: global batches of ``batch_size * world`` clips, of which this rank gets its shard (``shard_batches``)."""
    if world > 1:
        import copy
        g = copy.copy(args)
        g.batch_size = int(args.batch_size) * world
        tr, ev, te, length = create_data_loaders(g)
        return shard_batches(tr, rank, world), shard_batches(ev, rank, world, True), shard_batches(te, rank, world, True), length
    bs = int(args.batch_size)
    data_dir = getattr(args, "data_dir", None)
    if data_dir:
        # the reference's selection: length filter on the quantiles of the first 300 shuffled clips, seeded shuffles,
        # 80 / 10 / 10 split (build.py:183-208) — plan_dataset; collate = crop / right-zero-pad (build.py:41-61)
        waves, texts = load_local_dataset(data_dir, int(args.sr))
        plan = plan_dataset([len(w) for w in waves], int(args.seed), float(args.relative_audio_length), n_splits=1)
        length = plan["audio_length"]
        if getattr(args, "small_data", False):
            keep = max(3 * bs, len(plan["train"]) // 100)
            plan = {k: (v[:keep] if isinstance(v, list) else v) for k, v in plan.items()}
        mk = lambda ids: _batches(collate_fixed([waves[i] for i in ids], length), [texts[i] for i in ids], bs) if ids else []
        return mk(plan["train"]), mk(plan["eval"]), mk(plan["test"]), length
    else:
        length = int(round(float(getattr(args, "audio_seconds", 10.0)) * int(args.sr)))
        n = bs * int(getattr(args, "steps_per_epoch", 4)) * 10 // 8 + 2 * bs
        x = torch.from_numpy(synth.clean_audio(n, length, seed=int(args.seed)))
        words = ["the", "quick", "brown", "fox", "jumps", "over", "a", "lazy", "dog", "and", "runs", "away"]
        texts = []
        n_words = max(1, min(24, length // 320 // 12))       # ~1 character per 3 encoder frames: CTC-feasible at any length
        for i in range(n):
            u = synth.uniform(synth.key_of(f"txt{i}", int(args.seed)), n_words)
            texts.append(" ".join(words[int(v * len(words))] for v in u))
    n = len(texts)
    if getattr(args, "small_data", False):
        n = min(n, max(3 * bs, n // 100))
    n_tr, n_ev = int(0.8 * n), int(0.1 * n)
    tr = _batches(x[:n_tr], texts[:n_tr], bs)
    ev = _batches(x[n_tr:n_tr + n_ev], texts[n_tr:n_tr + n_ev], bs)
    te = _batches(x[n_tr + n_ev:n], texts[n_tr + n_ev:n], bs)
    return tr, ev, te, length
