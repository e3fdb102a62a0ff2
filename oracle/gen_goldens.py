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


def gen_spectrum():
    """The spectrum-level entry points of core/projections.py (:68-159) on a complex (B, F, T) tensor — what a
    maintainer calling them directly (train.py:52-59) gets.  Input: compute_stft of the synth perturbation."""
    out = {}
    interp = iso.build_weight_interpolator()
    for L, amp in ((4096, 0.3), (5000, 1e-2)):
        args = ref_args("fletcher_munson", ["--fm_epsilon", "0.05"])
        spl = build.init_phon_threshold_tensor(args)
        p = torch.from_numpy(np.concatenate([synth.perturbation(L) * np.float32(amp), synth.clean_audio(1, L) * np.float32(4.0)], 0))
        S = fourier_transforms.compute_stft(p, args)
        tag = f"L{L}|a{amp:g}"
        out[f"minmax|{tag}"] = torch.view_as_real(projections.project_min_max_freqs(args, S, 120.0, 20000.0).contiguous()).numpy()
        out[f"minmax_500_3000|{tag}"] = torch.view_as_real(projections.project_min_max_freqs(args, S, 500.0, 3000.0).contiguous()).numpy()
        out[f"fm_norm|{tag}"] = np.array([float(projections.compute_fm_weighted_norm_interp(S, interp, args))], dtype=np.float64)
        out[f"fm|{tag}"] = torch.view_as_real(projections.project_fm_norm(S, args, interp).contiguous()).numpy()
        big = ref_args("fletcher_munson", ["--fm_epsilon", "1e9"])
        out[f"fm_inactive|{tag}"] = torch.view_as_real(projections.project_fm_norm(S, big, interp).contiguous()).numpy()
        out[f"phon|{tag}"] = torch.view_as_real(projections.project_phon_level(S, args, spl).contiguous()).numpy()
    np.savez_compressed(os.path.join(GOLD, "spectrum.npz"), **out)
    print("spectrum.npz:", len(out), "arrays")


TRAJ_TEXTS = ["ab cd", "hello", "a b c", "xyz w", "the fox", "lazy dog"]


def gen_trajectory(proc):
    """train.train_epoch over a 3-batch loader for 2 epochs — the PGD branch (train.py:156-164) and the Adam + StepLR
    branch (train.py:165-175, build.py:352-359, scheduler stepped per epoch as run_attack.py:170-171) — on the tiny
    group-norm model: p after every epoch, per-epoch mean CTC / WER."""
    interp = iso.build_weight_interpolator()
    a = A.tiny("group", False)
    L, B, NB = 8000, 2, 3
    model = hf_model(a, A.rule_weights(a))
    loader = [(torch.from_numpy(synth.clean_audio(B, L, first_clip=i * B)), TRAJ_TEXTS[i * B:(i + 1) * B]) for i in range(NB)]
    for opt in ("pgd", "adam"):
        args = ref_args("snr", ["--snr_db", "40"])
        args.optimizer_type = opt
        args.lr = 1e-4 if opt == "pgd" else 2e-4
        args.step_size, args.gamma = 1, 0.5
        spl = build.init_phon_threshold_tensor(args)
        p = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2))
        optimizer = scheduler = None
        if opt == "adam":
            p = torch.nn.Parameter(p.clone())
            optimizer, scheduler = build.create_optimizer(args, p)
        out = {}
        for ep in range(2):
            res = train.train_epoch(args=args, train_data_loader=loader, p=p, model=model, epoch=ep, processor=proc,
                                    interp=interp, wer_metric=_Wer(), spl_thresh=spl, optimizer=optimizer)
            p = res.p
            if scheduler is not None:
                scheduler.step()
            out[f"p_epoch{ep}"] = p.detach().numpy().copy()
            out[f"ctc_epoch{ep}"] = np.array([res.avg_ctc])
            out[f"wer_epoch{ep}"] = np.array([res.avg_wer])
        np.savez_compressed(os.path.join(GOLD, f"traj_{opt}.npz"), **out)
        print(f"traj_{opt}.npz: ctc {float(out['ctc_epoch0'][0]):.5f} -> {float(out['ctc_epoch1'][0]):.5f}")


def gen_formats():
    """results.json exactly as the reference's save.save_json_results writes it (save.py:226-256)."""
    from training_utils import save as ref_save          # reference module (matplotlib present, torchaudio placeholder)
    with tempfile.TemporaryDirectory() as tmp:
        ref_save.save_json_results(save_dir=tmp, norm_type="snr", attack_size="40.0", epoch=3, finished_training=True,
                                   eval_score_clean={"ctc": 1589.123456, "wer": 0.41234567},
                                   eval_score_perturbed={"ctc": 2250.98765, "wer": 0.987654},
                                   train_score={"ctc": 2100.5, "wer": 0.9}, skipped=None,
                                   final_test_clean={"ctc": 1600.0, "wer": 0.4}, final_test_perturbed={"ctc": 2400.0, "wer": 0.9})
        with open(os.path.join(tmp, "results.json")) as f:
            text = f.read()
    with open(os.path.join(GOLD, "results_ref.json"), "w") as f:
        f.write(text)
    print("results_ref.json written")


def data_plan_lengths(n=400, seed=11):
    """Synthetic clip lengths (samples) for the dataset-plan fixture: 0.6 .. 3.1 s, deterministic."""
    u = synth.uniform(synth.key_of("cliplen", seed), n)
    return [int(9600 + 40000 * float(v)) for v in u]


def gen_data_plan():
    """build.create_data_loaders (build.py:104-220) driven with an in-memory stand-in for the LibriSpeech download
    (the dataset itself is unavailable offline; the clips are synthetic zeros of known lengths whose transcripts carry
    their ids): which clips survive the [10 %, relative_audio_length] length filter, the clip length, and the 80/10/10
    split after the seeded shuffles."""
    lengths = data_plan_lengths()

    class FakeLibri:                          # yields (waveform (1, T), sample_rate, transcript, speaker, chapter, utterance)
        def __init__(self, root, url, folder_in_archive, download):
            k = ["test-clean", "test-other", "dev-clean", "dev-other"].index(url)
            self.items = [(torch.zeros(1, lengths[i]), 16000, f"clip {i}", 0, 0, i) for i in range(k, len(lengths), 4)]

        def __iter__(self):
            return iter(self.items)

        def __len__(self):
            return len(self.items)

    old = build.LIBRISPEECH
    build.LIBRISPEECH = FakeLibri
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            args = ref_args("snr")
            args.dataset, args.batch_size, args.seed = "LibreeSpeech", 7, 5
            tr, ev, te, audio_len = build.create_data_loaders(args)
            ids = lambda loader: [int(t.split()[1]) for _, texts in loader for t in texts]
            plan = {"n_clips": len(lengths), "seed": 5, "relative_audio_length": args.relative_audio_length, "audio_length": int(audio_len),
                    "train_ids_sorted": sorted(ids(tr)), "eval_ids": ids(ev), "test_ids": ids(te),
                    "batch_shape_eval": list(next(iter(ev))[0].shape)}
    finally:
        os.chdir(cwd)
        build.LIBRISPEECH = old
    with open(os.path.join(GOLD, "data_plan.json"), "w") as f:
        json.dump(plan, f)
    print("data_plan.json: audio_length", plan["audio_length"], "kept", len(plan["train_ids_sorted"]) + len(plan["eval_ids"]) + len(plan["test_ids"]))


def gen_large(proc):
    """One PGD step on the large-lv60 topology (LayerNorm feature extractor with conv bias, pre-LN encoder, hidden 1024,
    24 layers, 16 heads) through HF's Wav2Vec2ForCTC with rule weights: L = 16000, B = 1, targeted max_phon (BASELINE
    config 4's workload).  2 s clips: the 34-token target "delete" x 5 (all <unk> with | separators) has 25
    repeated neighbours and needs 59 frames, more than the 49 a 1 s clip gives (its CTC loss would be inf)."""
    a = A.LARGE_LV60
    args = ref_args("max_phon", ["--max_phon_level", "20", "--attack_mode", "targeted", "--target", "delete"])
    args.lr = 1e-4
    interp = iso.build_weight_interpolator()
    spl = build.init_phon_threshold_tensor(args)
    model = hf_model(a, A.rule_weights(a))
    L, B = 32000, 1
    clean = torch.from_numpy(synth.clean_audio(B, L))
    p0 = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2))
    texts = PGD_TEXTS[:B]
    p = p0.clone().requires_grad_(True)
    loss, logits = loss_helpers.get_loss_for_training(model, (clean + p).clamp_(-1.0, 1.0), texts, proc, args)
    (-loss).backward()                                                       # targeted: direction = -1 (train.py:124,158)
    grad = p.grad.detach().numpy().copy()
    res = train.train_epoch(args=args, train_data_loader=[(clean, texts)], p=p0.clone(), model=model, epoch=0, processor=proc,
                            interp=interp, wer_metric=_Wer(), spl_thresh=spl, optimizer=None)
    np.savez_compressed(os.path.join(GOLD, "pgd_large_lv60_maxphon.npz"), loss=np.array([float(loss)]), avg_ctc=np.array([res.avg_ctc]),
                        avg_wer=np.array([res.avg_wer]), grad_samples=grad[0, ::13].copy(),
                        p_new_samples=res.p.detach().numpy()[0, ::13].copy(), logits_samples=logits.detach().numpy()[:, ::7, :].copy())
    print(f"pgd_large_lv60_maxphon.npz: loss={float(loss):.6f}")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    only = set(sys.argv[1:])                 # e.g. `python -m oracle.gen_goldens spectrum traj` regenerates just those
    want = lambda k: not only or k in only
    if want("iso"):
        gen_iso()
    if want("projections"):
        gen_projections()
    if want("spectrum"):
        gen_spectrum()
    if want("formats"):
        gen_formats()
    if want("data"):
        gen_data_plan()
    with tempfile.TemporaryDirectory() as tmp:
        proc = make_processor(tmp)
        if want("labels"):
            gen_labels(proc)
        if want("pgd"):
            gen_pgd(proc)
        if want("traj"):
            gen_trajectory(proc)
        if want("large"):
            gen_large(proc)


if __name__ == "__main__":
    main()
