"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/paa_hip.h declares, the ctypes
struct layouts match the C structs, and host-side argument validation raises the reference's exception types."""
import os
import re
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from paa_amd import _lib
    return _lib


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "paa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(paa_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    # every symbol the binding declares is in the header too
    assert set(lib.exported_symbols()) <= declared, set(lib.exported_symbols()) - declared


def test_struct_layouts_match(lib):
    import ctypes as C
    sizes = (C.c_int32 * 4)()
    lib.lib().paa_abi_sizes(sizes)
    assert list(sizes) == [C.sizeof(lib.PaaParams), C.sizeof(lib.PaaArch), C.sizeof(lib.PaaTensor), C.sizeof(lib.PaaGemmDesc)]
    assert lib.lib().paa_version() >= 100


def test_host_validation_without_gpu(lib):
    import torch
    from paa_amd import runtime
    from paa_amd.training_utils import parser, train
    args = parser.create_arg_parser().parse_args([])
    assert args.norm_type == "max_phon" and args.optimizer_type == "adam" and args.snr_db == 64 and args.lr == 1e-4
    assert args.n_fft == 1024 and args.hop_length == 256 and args.seed == 5 and args.batch_size == 64
    p = torch.zeros(1, 4096)
    bad = types.SimpleNamespace(**vars(args)); bad.norm_type = "l1"
    with pytest.raises(ValueError, match="Unknown norm_type"):
        train.perturbation_constraint(p, None, bad, None, None)
    snr = types.SimpleNamespace(**vars(args)); snr.norm_type = "snr"
    with pytest.raises(ValueError, match="SNR projection requires clean_audio"):
        train.perturbation_constraint(p, None, snr, None, None)
    l2 = types.SimpleNamespace(**vars(args)); l2.norm_type = "l2"
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the product path never computes on the CPU
        train.perturbation_constraint(p, None, l2, None, None)
    prm = runtime.params_of(snr)
    assert prm.norm_type == 2 and prm.direction == 1 and abs(prm.snr_db - 64) < 1e-6
    with pytest.raises(SystemExit):
        parser.create_arg_parser().parse_args(["--norm_type", "nope"])
    assert parser.create_arg_parser().parse_args(["--norm_type", "min_max_freqs+tv"]).norm_type == "min_max_freqs+tv"


def test_labels_match_reference_goldens():
    import json
    from paa_amd.core import loss_helpers
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "labels.json")))
    for mode in ("untargeted", "targeted"):
        args = types.SimpleNamespace(attack_mode=mode, target="delete", target_reps=5)
        texts = g[mode]["texts"]
        assert loss_helpers.clean_transcripts(texts) == g[mode]["cleaned"]
        assert loss_helpers.make_labels(texts, None, args, len(texts)).tolist() == g[mode]["labels"]
    assert [t.lower() for t in loss_helpers.greedy_decode_ids(g["decode"]["ids"])] == g["decode"]["texts"]


def test_iso_tables_match_reference_goldens(gold):
    import numpy as np
    from paa_amd.core import iso
    g = gold("iso.npz")
    tab = iso.build_weight_interpolator()
    np.testing.assert_allclose(tab.weights, g["weights"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(tab(g["probes"]), g["probe_vals"], rtol=1e-12, atol=1e-14)
    for phon in (20, 25, 0, 90):
        np.testing.assert_array_equal(iso.phon_threshold(phon), g[f"spl_thresh_{phon}"].reshape(-1))
    with pytest.raises(ValueError):
        iso.ISO226(91)
    with pytest.raises(ValueError):
        iso.ISO226(20)(np.array([10.0]))
