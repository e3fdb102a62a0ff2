"""-m gpu: the drop-in runner end to end on synthetic data (tiny architecture), the Adam branch, resume, and the
evaluation path's no-clamp semantics (evaluation.py:16) against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import rel_err
from oracle import pgd as opgd, projections as OP, wav2vec2 as OW
from paa_amd import arch as A, run_attack, synth
from paa_amd.model import PaaModel
from paa_amd.training_utils import parser

pytestmark = pytest.mark.gpu


def _run(tmp, extra):
    args = parser.create_arg_parser().parse_args(["--arch", "tiny", "--audio_seconds", "0.5", "--batch_size", "4",
                                                  "--steps_per_epoch", "2", "--num_epochs", "2", "--logs_dir", str(tmp),
                                                  "--dtype", "fp32", "--silent", *extra])
    rc = run_attack.main(args)
    return rc, args


def test_runner_pgd_and_resume(tmp_path):
    rc, args = _run(tmp_path, ["--optimizer_type", "pgd", "--norm_type", "snr", "--snr_db", "40"])
    assert rc == 0
    d = json.load(open(os.path.join(args.save_dir, "results.json")))
    assert d["finished_training"] == 1.0 and d["norm_type"] == "snr" and "final_test_perturbed" in d and "perturbation_efficiency" in d
    p = torch.load(os.path.join(args.save_dir, "perturbation.pt"), weights_only=True)
    assert tuple(p.shape) == (1, 8000) and torch.isfinite(p).all()
    assert os.path.exists(os.path.join(args.save_dir, "perturbation.wav")) and os.path.exists(os.path.join(args.save_dir, "perturbation_5x.wav"))
    # second launch finds the checkpoint and resumes from it (build.py:265-286)
    rc2, args2 = _run(tmp_path, ["--optimizer_type", "pgd", "--norm_type", "snr", "--snr_db", "40", "--num_epochs", "3"])
    assert rc2 == 0 and args2.resume is True


def test_runner_adam_targeted(tmp_path):
    rc, args = _run(tmp_path, ["--optimizer_type", "adam", "--norm_type", "max_phon", "--attack_mode", "targeted",
                               "--target", "ab", "--target_reps", "2", "--num_epochs", "1"])
    assert rc == 0
    assert json.load(open(os.path.join(args.save_dir, "results.json")))["finished_training"] == 1.0


def test_evaluation_forward_no_clamp():
    a = A.tiny()
    B, L = 2, 8000
    sdn = A.rule_weights(a)
    args = OP.default_args()
    clean = torch.from_numpy(synth.clean_audio(B, L) * 15)              # push samples beyond [-1, 1] so a clamp would show
    p = torch.from_numpy(synth.perturbation(L) * np.float32(0.3))
    labels = opgd.make_labels(["ab cd", "hello"], args, B)
    ref_loss, ref_logits = OW.forward(OW.to_torch(sdn), a, clean + p, labels)
    m = PaaModel(a, sdn, B, L, "fp32")
    r = m.forward(clean.cuda(), p.cuda(), labels, clamp=False)
    assert rel_err(r["logits"].cpu().numpy(), ref_logits.detach().numpy()) < 1e-3
    assert abs(float(r["loss"]) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    rc = m.forward(clean.cuda(), p.cuda(), labels, clamp=True)
    assert abs(float(rc["loss"]) - float(ref_loss)) > 1e-3 * abs(float(ref_loss))     # the clamped path differs
    r0 = m.forward(clean.cuda(), None, labels)
    ref0, _ = OW.forward(OW.to_torch(sdn), a, clean, labels)
    assert abs(float(r0["loss"]) - float(ref0)) < 1e-4 * abs(float(ref0))
