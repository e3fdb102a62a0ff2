"""In-process A/B of the fp32-parity PGD step under the switches only the diagnostic library (-DPAA_EXPERIMENTS) reads, on ONE
device: one model per variant at the headline shape, timed in alternating rounds (boxes differ by +-4 %, rounds on one box by 0.1 %).

    PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python psychoacoustic-adverserial-attacks_amd/build_ext.py     # here, no GPU needed
    gpurun -- 'PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python tools/model_ab.py default no_ail no_c0dma'

Variants: default = the shipped behaviour; no_ail = planar activation planes instead of the interleaved ones (gemm.h A_il / Cb_il;
read at model creation); no_c0dma = the register-staged conv0 GroupNorm backward instead of the LDS-DMA one (read per launch);
no_rln = f32 LayerNorm outputs written and read back as residuals instead of LN(x) evaluated in the epilogue (gemm.h res_ln_stats).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from paa_amd import _lib, arch as A, synth
from paa_amd.core import loss_helpers
from paa_amd.model import PaaModel
from paa_amd.training_utils import parser
from paa_amd.training_utils.pgd import PgdStepper


VARIANTS = {"default": {}, "no_ail": {"PAA_NO_AIL": "1"}, "no_c0dma": {"PAA_NO_C0DMA": "1"}, "no_rln": {"PAA_NO_RLN": "1"}}
SWITCHES = ("PAA_NO_AIL", "PAA_NO_C0DMA", "PAA_NO_RLN")


def set_env(name):
    for k in SWITCHES:
        os.environ[k] = VARIANTS[name].get(k, "0")


def main(names, steps=24, rounds=3):
    assert _lib.lib().paa_version() == 301, "build the diagnostic library first (see the module docstring)"
    a, B, L = A.BASE, 32, 160000
    clean = torch.from_numpy(synth.clean_audio(B, L, seed=5)).cuda()
    texts = [("the quick brown fox jumps over a lazy dog and runs " * 4)[:150] for _ in range(B)]
    args = parser.create_arg_parser().parse_args(["--norm_type", "snr", "--snr_db", "40", "--lr", "1e-4", "--optimizer_type", "pgd",
                                                  "--device", "cuda", "--dtype", "fp32"])
    labels = loss_helpers.make_labels(texts, None, args, B).to(device="cuda", dtype=torch.int32)
    p0 = (torch.from_numpy(synth.perturbation(L, seed=5)) * np.float32(2e-3)).cuda()
    st, ps = {}, {}
    for name in names:
        set_env(name)
        st[name] = PgdStepper(PaaModel(a, A.rule_weights(a), B, L, "fp32"), args, L)
        ps[name] = p0.clone()
    ms = {k: [] for k in st}
    for rnd in range(rounds + 1):
        for name in st:
            set_env(name)
            for _ in range(2):
                st[name].step(ps[name], clean, labels, want_logits=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st[name].step(ps[name], clean, labels, want_logits=False)
            torch.cuda.synchronize()
            if rnd:
                ms[name].append(round((time.perf_counter() - t0) * 1e3 / steps, 3))
    ref = ps[names[0]]
    print(json.dumps({"ms_per_step": ms, "p_equal_to_first_variant": {n: bool(torch.equal(ps[n], ref)) for n in names},
                      "max_abs_diff_to_first_variant": {n: float((ps[n] - ref).abs().max()) for n in names}}), flush=True)


if __name__ == "__main__":
    main([n for n in sys.argv[1:] if n in VARIANTS] or ["default", "no_ail", "no_c0dma"])
