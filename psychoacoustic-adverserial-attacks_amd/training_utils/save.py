"""On-disk formats of the reference's ``src/training_utils/save.py`` that a run produces and a resume consumes:
``perturbation.pt`` (save.py:155-156; CPU float32 (1, L) tensor, loadable with ``weights_only=True``),
``results.json`` (save.py:226-256) and the int16 PCM wavs (save.py:11-21,159-160).  Plots are out of scope."""
from __future__ import annotations

import json
import os
import wave

import numpy as np
import torch


def save_pert(p, path):
    """save.py:155-156."""
    torch.save(p.detach().cpu(), path)


def load_pert(path, device="cpu"):
    """run_attack.py:187 — weights_only load of a saved perturbation."""
    return torch.load(path, map_location=device, weights_only=True)


def save_audio(filename, tensor, sample_rate=16000, amplify=1.0):
    """save.py:11-21: amplify, clamp to [-1, 1], int16 PCM mono wav (written with the stdlib ``wave`` module)."""
    t = torch.clamp(tensor.detach().cpu().float() * amplify, -1.0, 1.0)
    pcm = (t * 32767).to(torch.int16).reshape(-1).numpy()
    with wave.open(filename, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(pcm.astype("<i2").tobytes())


def load_audio(filename):
    """int16 / 32-bit PCM wav -> (float32 mono samples in [-1, 1], sample rate)."""
    with wave.open(filename, "rb") as w:
        n, ch, sw, sr = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"{filename}: unsupported sample width {sw}")
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    return x, sr


def save_by_epoch(args, p):
    """save.py:158-160 (audio artefacts only)."""
    save_audio(os.path.join(args.save_dir, "perturbation.wav"), p, sample_rate=args.sr)
    save_audio(os.path.join(args.save_dir, "perturbation_5x.wav"), p, sample_rate=args.sr, amplify=5)


def save_json_results(save_dir, norm_type, attack_size, **kwargs):
    """save.py:226-256 — same keys, rounding and ``perturbation_efficiency`` rule."""
    def safe_to_float(v):
        return {k: round(float(v[k]), 4) for k in v} if isinstance(v, dict) else (v if isinstance(v, str) else float(v))

    results = {"norm_type": norm_type, "attack_size": float(attack_size)}
    for key, val in kwargs.items():
        if val is not None:
            results[key] = safe_to_float(val)
    clean = kwargs.get("final_test_clean") or kwargs.get("test_loss_clean")
    pert = kwargs.get("final_test_perturbed") or kwargs.get("test_loss_perturbed")
    if clean is not None and pert is not None:
        if isinstance(clean, dict):
            results["perturbation_efficiency"] = {k: pert[k] / clean[k] for k in clean}
        else:
            results["perturbation_efficiency"] = pert / clean
    with open(os.path.join(save_dir, "results.json"), "w") as f:
        json.dump(results, f, indent=2)
    return results
