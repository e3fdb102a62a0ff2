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


@pytest.mark.parametrize("norm,stable", [("group", False), ("layer", True)])
def test_load_model_from_local_checkpoint(tmp_path, norm, stable):
    """Row (g): the reference loads its model by name (src/training_utils/build.py:225-231, from_pretrained); here --model_path
    names a LOCAL HuggingFace checkpoint directory.  A tiny Wav2Vec2ForCTC of each topology is written with save_pretrained
    (config.json + safetensors, plus the tokenizer / feature-extractor files of a Wav2Vec2Processor) and loaded back through
    build.load_model: architecture mapping, weight packing (parametrised weight-norm keys) and the processor must all come from the
    files — logits equal to the model built from the same weights directly (bit for bit) and to HF's own forward (fp32-parity
    tolerance); labels through the loaded processor equal to the built-in tokenizer's."""
    pytest.importorskip("transformers")
    import json as js
    from transformers import Wav2Vec2CTCTokenizer, Wav2Vec2FeatureExtractor, Wav2Vec2Processor
    from hf_util import hf_model
    from paa_amd.core import loss_helpers
    from paa_amd.training_utils import build
    a = A.tiny(norm, stable)
    sdn = A.rule_weights(a)
    hf = hf_model(a, sdn)
    ck = tmp_path / "ckpt"
    hf.save_pretrained(str(ck))
    vp = tmp_path / "vocab.json"
    vp.write_text(js.dumps({t: i for i, t in enumerate(loss_helpers.VOCAB)}))
    tok = Wav2Vec2CTCTokenizer(str(vp), unk_token="<unk>", pad_token="<pad>", word_delimiter_token="|")
    Wav2Vec2Processor(feature_extractor=Wav2Vec2FeatureExtractor(), tokenizer=tok).save_pretrained(str(ck))
    B, L = 2, 8000
    args = parser.create_arg_parser().parse_args(["--model_path", str(ck), "--dtype", "fp32", "--device", "cuda", "--norm_type", "snr",
                                                  "--optimizer_type", "pgd"])
    m, proc = build.load_model(args, max_batch=B, length=L)
    assert m.arch == a and proc is not None
    for texts in (["hello world it's", "a <unk> b"], ["ab cd", "hello"]):        # the second pair is CTC-feasible in 24 frames
        lab_proc = loss_helpers.make_labels(texts, proc, args, B)
        lab_own = loss_helpers.make_labels(texts, None, args, B)
        assert torch.equal(lab_proc, lab_own)
    clean = torch.from_numpy(synth.clean_audio(B, L)).cuda()
    p = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2)).cuda()
    got = m.fwd_bwd(clean, p, lab_proc, +1)
    m2 = PaaModel(a, sdn, B, L, "fp32")
    direct = m2.fwd_bwd(clean, p, lab_own, +1)
    torch.cuda.synchronize()
    assert torch.isfinite(got["grad"]).all() and float(got["grad"].abs().max()) > 0
    assert torch.equal(got["logits"], direct["logits"]) and torch.equal(got["grad"], direct["grad"])
    with torch.no_grad():
        ref = hf(input_values=(clean.cpu() + p.cpu()).clamp(-1, 1), labels=lab_own)
    assert rel_err(got["logits"].cpu().numpy(), ref.logits.numpy()) < 1e-3
    assert abs(float(got["loss"]) - float(ref.loss)) < 2e-4 * abs(float(ref.loss))
    # and the runner end to end on that checkpoint
    rc, rargs = _run(tmp_path / "logs", ["--optimizer_type", "pgd", "--norm_type", "linf", "--model_path", str(ck), "--num_epochs", "1"])
    assert rc == 0 and json.load(open(os.path.join(rargs.save_dir, "results.json")))["finished_training"] == 1.0


def test_bench_line_contract():
    """`python bench.py` (tiny shape so that it takes seconds): ONE JSON line on stdout with the driver's contract — metric / value /
    ms_per_step / n_gpus / scaling / dtype / data / config.workload — plus the `roofline` object measured live by HIP events around the
    GEMM launches and the `cpu_baseline` object (the oracle on a bounded sample, labelled as an extrapolated port); the step is replayed
    from captured hipGraphs except for the sampled step."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--arch", "tiny", "--steps", "4", "--warmup", "1", "--batch", "2", "--seconds", "1",
                        "--label_tokens", "10", "--cpu_batch", "1", "--cpu_steps", "1", "--no_baseline_faithful", "--no_fft_bench"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                   # stdout carries the one JSON line and nothing else
    d = json.loads(lines[0])
    assert d["metric"] == "pgd_steps_per_sec" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3
    assert "workload" in d["config"] and d["config"]["hip_graph"] is True and d["config"]["hip_graph_replayed_steps"] == 3
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    # traffic: HBM bytes per launch of the dominant kernel from the run's own rocprofv3 --pmc child passes, or null WITH the reason on stderr
    if roof["traffic"] is None:
        assert "roofline.traffic stays null" in r.stderr, r.stderr[-2000:]
    else:
        assert roof["traffic"] > 0 and roof["traffic_source"].startswith("measured by this run") and roof["traffic_over_algorithmic"] > 0.5
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["sampled_steps"] == 1
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"].startswith("port") and "extrapolated" in cb["kind"] and cb["sample"]
    assert "bf16" in d and d["bf16"]["value"] > 0 and "grad_sign_flip_rate" in d["bf16"]["vs_headline_mode"]
