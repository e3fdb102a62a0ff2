"""In-process A/B of the fp32-parity PGD step with the activation planes interleaved per 32-element group (gemm.h A_il / Cb_il, the
default) against planar planes, on ONE device: two models of the headline shape live side by side and are timed in alternating
rounds.  Needs the diagnostic library (PAA_NO_AIL is read only by -DPAA_EXPERIMENTS builds):

    PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python psychoacoustic-adverserial-attacks_amd/build_ext.py     # here, no GPU needed
    gpurun -- 'PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python tools/ail_ab.py'
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


def main(steps=24, rounds=3):
    assert _lib.lib().paa_version() == 301, "build the diagnostic library first (see the module docstring)"
    a, B, L = A.BASE, 32, 160000
    clean = torch.from_numpy(synth.clean_audio(B, L, seed=5)).cuda()
    texts = [("the quick brown fox jumps over a lazy dog and runs " * 4)[:150] for _ in range(B)]
    args = parser.create_arg_parser().parse_args(["--norm_type", "snr", "--snr_db", "40", "--lr", "1e-4", "--optimizer_type", "pgd",
                                                  "--device", "cuda", "--dtype", "fp32"])
    labels = loss_helpers.make_labels(texts, None, args, B).to(device="cuda", dtype=torch.int32)
    p0 = (torch.from_numpy(synth.perturbation(L, seed=5)) * np.float32(2e-3)).cuda()
    st, ps = {}, {}
    for name, env in (("interleaved", "0"), ("planar", "1")):
        os.environ["PAA_NO_AIL"] = env
        st[name] = PgdStepper(PaaModel(a, A.rule_weights(a), B, L, "fp32"), args, L)
        ps[name] = p0.clone()
    ms = {k: [] for k in st}
    for rnd in range(rounds + 1):
        for name in st:
            for _ in range(2):
                st[name].step(ps[name], clean, labels, want_logits=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st[name].step(ps[name], clean, labels, want_logits=False)
            torch.cuda.synchronize()
            if rnd:
                ms[name].append(round((time.perf_counter() - t0) * 1e3 / steps, 3))
    same = bool(torch.equal(ps["interleaved"], ps["planar"]))
    print(json.dumps({"ms_per_step": ms, "p_equal_after_all_steps": same,
                      "max_abs_diff": float((ps["interleaved"] - ps["planar"]).abs().max())}), flush=True)


if __name__ == "__main__":
    main()
