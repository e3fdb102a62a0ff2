"""ORACLE tooling — generates tests/golden/* by importing the REFERENCE itself.

Runs only in the build container (needs /root/reference); the goldens it writes are committed and
are what travels.  Recipe follows SURVEY.md Appendix B: ``torchaudio`` is absent in this image and
is needed by the reference only for wav writing / dataset download (training_utils/save.py:2,
build.py), so an empty module object is registered for it AFTER importing transformers; no
reference source is copied, patched or re-implemented here.

    python -m oracle.gen_goldens            # from the repo root

Inputs are produced by the counter-based generators in paa_amd.synth, so the fixtures hold
expected OUTPUTS (plus the small tables) only.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import transformers  # noqa: E402  (must precede the torchaudio placeholder)
from transformers import (Wav2Vec2Config, Wav2Vec2CTCTokenizer, Wav2Vec2FeatureExtractor,  # noqa: E402
                          Wav2Vec2ForCTC, Wav2Vec2Processor)

_ta = types.ModuleType("torchaudio")
_ta.datasets = types.ModuleType("torchaudio.datasets")
_ta.datasets.LIBRISPEECH = None
sys.modules["torchaudio"] = _ta
sys.modules["torchaudio.datasets"] = _ta.datasets

from core import fourier_transforms, iso, loss_helpers, projections  # noqa: E402  (reference)
from training_utils import build, parser, train  # noqa: E402  (reference)

from paa_amd import arch as A  # noqa: E402
from paa_amd import synth  # noqa: E402
from oracle import pgd as opgd  # noqa: E402
from oracle.gen_cases import AMPS, LENGTHS, NORM_CASES, PGD_CASES, PGD_TEXTS, case_name  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")



def ref_args(norm, extra=()):
    a = parser.create_arg_parser().parse_args(["--optimizer_type", "pgd", "--norm_type", norm, *extra])
    a.device = "cpu"
    return a


def gen_projections():
    out = {}
    interp = iso.build_weight_interpolator()
    for norm, extra in NORM_CASES:
        args = ref_args(norm, extra)
        spl = build.init_phon_threshold_tensor(args)
        for L in LENGTHS + [16000]:
            for B in ([1, 3] if norm in ("snr", "tv") else [1]):
                for amp in AMPS:
                    clean = torch.from_numpy(synth.clean_audio(B, L))
                    p = torch.from_numpy(synth.perturbation(L) * np.float32(amp))
                    q = train.perturbation_constraint(p, clean, args, interp, spl)
                    q = q.detach().numpy().astype(np.float32)
                    name = case_name(norm, extra, L, B, amp)
                    if L == 16000:      # long case: strided samples + norms keep the fixture small
                        out[name + "|samples"] = q[0, ::7].copy()
                        out[name + "|norms"] = np.array([np.abs(q).sum(dtype=np.float64),
                                                         np.sqrt((q.astype(np.float64) ** 2).sum()),
                                                         np.abs(q).max()], dtype=np.float64)
                    else:
                        out[name] = q
    # STFT / iSTFT on their own (fourier_transforms.py), incl. a (B=2) batch.
    args = ref_args("max_phon")
    for L in (4096, 5000):
        p = torch.from_numpy(np.concatenate([synth.perturbation(L) * np.float32(1e-2),
                                             synth.clean_audio(1, L)], 0))
        S = fourier_transforms.compute_stft(p, args)
        out[f"stft|L{L}"] = torch.view_as_real(S).numpy()
        out[f"istft|L{L}"] = fourier_transforms.compute_istft(S, args).numpy()
    # Scalars the projections go through, for diagnosing a mismatch.
    args = ref_args("fletcher_munson")
    for amp in AMPS:
        S = fourier_transforms.compute_stft(torch.from_numpy(synth.perturbation(4096) * np.float32(amp)), args)
        out[f"fm_norm|L4096|a{amp:g}"] = np.array(
            [float(projections.compute_fm_weighted_norm_interp(S, interp, args))], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "projections.npz"), **out)
    print("projections.npz:", len(out), "arrays")


def gen_iso():
    freqs, phons, spl = iso.compute_iso226_weight_matrix()
    w = iso.perceptual_weight(spl)
    interp = iso.build_weight_interpolator()
    probes = np.array([[-5, 1000], [0, 1000], [35, 1000], [90, 8000], [91, 1000], [40, 15.625], [40, 20],
                       [40, 7992], [12.5, 31.25], [89.99, 19999], [0, 20], [90, 20000], [45, 20000.1],
                       [3.3, 437.5], [77.7, 7984.375]], dtype=np.float64)
    out = dict(freqs=freqs, phons=phons.astype(np.float64), spl=spl, weights=w, probes=probes,
               probe_vals=interp(probes))
    for phon in (20, 25, 0, 90):
        a = ref_args("max_phon", ["--max_phon_level", str(phon)])
        out[f"spl_thresh_{phon}"] = build.init_phon_threshold_tensor(a).numpy()
    np.savez_compressed(os.path.join(GOLD, "iso.npz"), **out)
    print("iso.npz written")


def make_processor(tmp):
    vocab = {t: i for i, t in enumerate(opgd.VOCAB)}
    vp = os.path.join(tmp, "vocab.json")
    with open(vp, "w") as f:
        json.dump(vocab, f)
    tok = Wav2Vec2CTCTokenizer(vp, unk_token="<unk>", pad_token="<pad>", word_delimiter_token="|")
    return Wav2Vec2Processor(feature_extractor=Wav2Vec2FeatureExtractor(), tokenizer=tok)


TEXTS = ["hello world it's", "THE QUICK  brown fox", "a <unk> b", "delete delete delete delete delete",
         "  leading and trailing  ", "x"]


def gen_labels(proc):
    out = {}
    for mode in ("untargeted", "targeted"):
        args = ref_args("snr", ["--attack_mode", mode])
        texts = TEXTS if mode == "untargeted" else [" ".join([args.target] * args.target_reps)] * 3
        cleaned = loss_helpers.clean_transcripts(texts)
        lab = proc(text=cleaned, return_tensors="pt", padding=True).input_ids
        lab[lab == proc.tokenizer.pad_token_id] = -100
        out[mode] = dict(texts=texts, cleaned=cleaned, labels=lab.tolist())
    # greedy decode goldens: ids -> text through the HF tokenizer (loss_helpers.py:26-28)
    ids = [[0, 0, 11, 11, 0, 5, 15, 0, 15, 8, 4, 4, 18, 8, 13, 15, 14, 0, 0],
           [3, 3, 4, 27, 12, 0, 12, 4, 0, 0, 1, 2, 7, 7, 0, 7]]
    out["decode"] = dict(ids=ids, texts=[t.strip().lower() for t in proc.batch_decode(ids, skip_special_tokens=True)])
    with open(os.path.join(GOLD, "labels.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("labels.json written")


class _Wer:
    """Stand-in for hf-evaluate's ``wer`` metric object (absent offline, run_attack.py:27)."""

    def compute(self, predictions, references):
        return opgd.wer(predictions, references)[0]


def hf_model(a: A.Wav2Vec2Arch, sd_np):
    cfg = Wav2Vec2Config(vocab_size=a.vocab_size, hidden_size=a.hidden_size, num_hidden_layers=a.num_hidden_layers,
                         num_attention_heads=a.num_attention_heads, intermediate_size=a.intermediate_size,
                         conv_dim=list(a.conv_dim), conv_kernel=list(a.conv_kernel), conv_stride=list(a.conv_stride),
                         conv_bias=a.conv_bias, feat_extract_norm=a.feat_extract_norm,
                         num_conv_pos_embeddings=a.num_conv_pos_embeddings,
                         num_conv_pos_embedding_groups=a.num_conv_pos_embedding_groups,
                         do_stable_layer_norm=a.do_stable_layer_norm, pad_token_id=a.pad_token_id)
    m = Wav2Vec2ForCTC(cfg).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("masked_spec_embed" in k for k in missing), missing
    return m


def gen_pgd(proc):
    interp = iso.build_weight_interpolator()
    for name, a, L, B, norm, extra in PGD_CASES:
        args = ref_args(norm, extra)
        args.lr = 1e-4
        spl = build.init_phon_threshold_tensor(args)
        sd = A.rule_weights(a)
        model = hf_model(a, sd)
        clean = torch.from_numpy(synth.clean_audio(B, L))
        p0 = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2))
        texts = PGD_TEXTS[:B]
        # (a) gradient wrt p through the reference's loss helper (train.py:133-158)
        p = p0.clone().requires_grad_(True)
        perturbed = (clean + p).clamp_(-1.0, 1.0)
        loss, logits = loss_helpers.get_loss_for_training(model, perturbed, texts, proc, args)
        direction = +1 if args.attack_mode == "untargeted" else -1
        (direction * loss).backward()
        grad = p.grad.detach().numpy().copy()
        # (b) the reference's own epoch over a one-batch loader (train.py:103-182)
        res = train.train_epoch(args=args, train_data_loader=[(clean, texts)], p=p0.clone(), model=model, epoch=0,
                                processor=proc, interp=interp, wer_metric=_Wer(), spl_thresh=spl, optimizer=None)
        out = dict(loss=np.array([float(loss)]), avg_ctc=np.array([res.avg_ctc]), avg_wer=np.array([res.avg_wer]))
        if a is A.BASE:
            out["grad_samples"] = grad[0, ::13].copy()
            out["p_new_samples"] = res.p.detach().numpy()[0, ::13].copy()
            out["logits_samples"] = logits.detach().numpy()[:, ::7, :].copy()
        else:
            out["grad"] = grad
            out["p_new"] = res.p.detach().numpy()
            out["logits"] = logits.detach().numpy()
        np.savez_compressed(os.path.join(GOLD, f"pgd_{name}.npz"), **out)
        print(f"pgd_{name}.npz: loss={float(loss):.6f} ctc={res.avg_ctc:.6f} wer={res.avg_wer:.4f}")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    gen_iso()
    gen_projections()
    with tempfile.TemporaryDirectory() as tmp:
        proc = make_processor(tmp)
        gen_labels(proc)
        gen_pgd(proc)


if __name__ == "__main__":
    main()
