"""No-GPU checks of the rows SURVEY §8f marks 'next': on-disk formats, resume rule, fixed-length collate, split."""
import json
import os
import types

import numpy as np
import pytest
import torch

from paa_amd.training_utils import build, parser, save, scoring_helpers


def test_perturbation_pt_roundtrip(tmp_path):
    p = torch.randn(1, 1234)
    f = tmp_path / "perturbation.pt"
    save.save_pert(p.requires_grad_(True), str(f))
    q = torch.load(str(f), map_location="cpu", weights_only=True)      # run_attack.py:187
    assert q.dtype == torch.float32 and tuple(q.shape) == (1, 1234) and torch.equal(q, p.detach())
    assert torch.equal(save.load_pert(str(f)), p.detach())


def test_results_json_schema(tmp_path):
    r = save.save_json_results(str(tmp_path), "snr", "40.0", epoch=3, finished_training=True, best_epoch=2,
                               eval_score_clean={"ctc": 1.23456, "wer": 0.5}, final_test_clean={"ctc": 2.0, "wer": 0.4},
                               final_test_perturbed={"ctc": 3.0, "wer": 0.8}, train_score=None)
    d = json.load(open(tmp_path / "results.json"))
    assert d == r
    assert d["norm_type"] == "snr" and d["attack_size"] == 40.0 and d["epoch"] == 3.0 and d["finished_training"] == 1.0
    assert d["eval_score_clean"] == {"ctc": 1.2346, "wer": 0.5} and "train_score" not in d
    assert d["perturbation_efficiency"] == {"ctc": 1.5, "wer": 2.0}
    save.save_json_results(str(tmp_path), "snr", "40.0", epoch=-1, finished_training=False, error="boom")
    assert json.load(open(tmp_path / "results.json"))["error"] == "boom"


def test_wav_int16_roundtrip(tmp_path):
    x = torch.tensor([[0.0, 0.5, -0.5, 1.5, -2.0, 1e-4]])
    f = str(tmp_path / "perturbation.wav")
    save.save_audio(f, x, sample_rate=16000, amplify=1.0)
    y, sr = save.load_audio(f)
    assert sr == 16000
    ref = (torch.clamp(x, -1, 1) * 32767).to(torch.int16).numpy().reshape(-1)
    np.testing.assert_array_equal((y * 32768).round().astype(np.int16), ref)
    save.save_audio(f, x, amplify=5)
    y5, _ = save.load_audio(f)
    assert abs(y5[5] - 5e-4) < 1 / 32768


def test_collate_and_percentile():
    waves = [np.ones(n, np.float32) * (i + 1) for i, n in enumerate([10, 20, 30, 40, 50])]
    L = build.percentile_length([len(w) for w in waves], 0.80)
    assert L == int(np.quantile(np.array([10, 20, 30, 40, 50]), 0.8))
    x = build.collate_fixed(waves, L)
    assert tuple(x.shape) == (5, L)
    assert float(x[0, 9]) == 1 and float(x[0, 10]) == 0            # right zero-pad
    assert float(x[4, L - 1]) == 5                                  # crop


def _args(**kw):
    a = parser.create_arg_parser().parse_args(["--optimizer_type", "pgd", "--norm_type", "snr", "--snr_db", "40", "--silent"])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_logger_naming_and_resume(tmp_path):
    a = _args(logs_dir=str(tmp_path))
    _, ep = build.create_logger(a)
    assert a.save_dir.endswith(os.path.join("untargeted", "LibreeSpeech", "snr_40.0_untargeted_pgd")) and ep == 0
    assert a.resume is False and a.attack_size_string == "40.0"
    save.save_pert(torch.zeros(1, 8), os.path.join(a.save_dir, "perturbation.pt"))
    save.save_json_results(a.save_dir, "snr", "40.0", epoch=7, finished_training=False)
    b = _args(logs_dir=str(tmp_path), resume_from="/somewhere/else.pt")
    _, ep = build.create_logger(b)
    assert ep == 7 and b.resume is True and b.resume_from == os.path.join(b.save_dir, "perturbation.pt")
    c = _args(logs_dir=str(tmp_path), small_data=True)              # --small_data disables resume (build.py:279)
    _, ep = build.create_logger(c)
    assert c.resume is False


def test_local_dataset_and_split(tmp_path):
    rng = np.random.default_rng(0)
    lines = []
    for i in range(20):
        n = 3000 + 200 * i
        save.save_audio(str(tmp_path / f"utt{i:02d}.wav"), torch.from_numpy(rng.normal(size=n).astype(np.float32) * 0.05))
        lines.append(f"utt{i:02d} HELLO WORLD {i}")
    (tmp_path / "x.trans.txt").write_text("\n".join(lines))
    a = _args(data_dir=str(tmp_path), batch_size=4)
    tr, ev, te, L = build.create_data_loaders(a)
    # the reference DROPS clips outside the [10 %, 80 %] length quantiles (build.py:183-189): 3400..6000 samples survive
    assert L == int(np.quantile(np.array([3000 + 200 * i for i in range(20)]), 0.8)) == 6040
    kept = [t for part in (tr, ev, te) for _, texts in part for t in texts]
    assert sorted(int(t.split()[-1]) for t in kept) == list(range(2, 16))
    assert sum(len(t) for _, t in tr) == 11 and sum(len(t) for _, t in ev) == 1 and sum(len(t) for _, t in te) == 2
    assert tr[0][0].shape == (4, L)
    for x, texts in tr + ev + te:                                     # right zero-pad to L, never cropped here
        for row, t in zip(x, texts):
            n = 3000 + 200 * int(t.split()[-1])
            assert float(row[n:].abs().max()) == 0 and float(row[:n].abs().max()) > 0
    s = _args(batch_size=2, audio_seconds=0.25, steps_per_epoch=3)
    tr, ev, te, L = build.create_data_loaders(s)
    assert L == 4000 and len(tr) >= 3 and len(ev) >= 1 and len(te) >= 1


def test_scoring_helpers():
    assert scoring_helpers._is_better(2.0, 1.0, "untargeted") and not scoring_helpers._is_better(2.0, 1.0, "targeted")
    assert scoring_helpers._best_agg([1, 3, 2], "untargeted") == 3 and scoring_helpers._best_agg([1, 3, 2], "targeted") == 1
    assert scoring_helpers._best_agg([], "targeted") == float("inf")
    with pytest.raises(ValueError):
        scoring_helpers._is_better(1, 2, "sideways")
    up, down = scoring_helpers.Objective("untargeted"), scoring_helpers.Objective("targeted")
    assert up.worst == float("-inf") and down.worst == float("inf")
    assert not up.improves(float("nan"), 0.0) and not down.improves(float("nan"), 0.0)
    assert not up.improves(1.0, 1.0) and up.improves(float("inf"), up.worst) and not up.improves(up.worst, up.worst)
    assert scoring_helpers._best_agg([], "untargeted") == float("-inf")


def test_results_json_equals_reference_writer(tmp_path, gold):
    """tests/golden/results_ref.json was written by the REFERENCE's save.save_json_results (oracle/gen_goldens.py
    gen_formats); the same call here must give the same file, byte for byte."""
    save.save_json_results(save_dir=str(tmp_path), norm_type="snr", attack_size="40.0", epoch=3, finished_training=True,
                           eval_score_clean={"ctc": 1589.123456, "wer": 0.41234567},
                           eval_score_perturbed={"ctc": 2250.98765, "wer": 0.987654},
                           train_score={"ctc": 2100.5, "wer": 0.9}, skipped=None,
                           final_test_clean={"ctc": 1600.0, "wer": 0.4}, final_test_perturbed={"ctc": 2400.0, "wer": 0.9})
    from conftest import GOLD
    assert open(tmp_path / "results.json").read() == open(os.path.join(GOLD, "results_ref.json")).read()


def test_dataset_plan_equals_reference(gold):
    """tests/golden/data_plan.json: which clips the reference's create_data_loaders keeps (length filter on the 10 % /
    relative_audio_length quantiles of the first 300 shuffled clips, build.py:183-189), the clip length, and its
    80 / 10 / 10 split after the seeded shuffles — produced by running the reference on 400 synthetic clips."""
    from conftest import GOLD
    from paa_amd import synth
    ref = json.load(open(os.path.join(GOLD, "data_plan.json")))
    u = synth.uniform(synth.key_of("cliplen", 11), ref["n_clips"])
    lengths = [int(9600 + 40000 * float(v)) for v in u]
    plan = build.plan_dataset(lengths, ref["seed"], ref["relative_audio_length"])
    assert plan["audio_length"] == ref["audio_length"] == ref["batch_shape_eval"][1]
    assert sorted(plan["train"]) == ref["train_ids_sorted"]
    assert plan["eval"] == ref["eval_ids"] and plan["test"] == ref["test_ids"]
    assert all(plan["min_len"] <= lengths[i] <= plan["audio_length"] for k in ("train", "eval", "test") for i in plan[k])
