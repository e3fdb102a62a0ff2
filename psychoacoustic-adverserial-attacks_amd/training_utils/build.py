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
