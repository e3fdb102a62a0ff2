"""In-process A/B of the fp32-parity PGD step with gelu'(v) kept by the forward pass (default) against the raw pre-activation
(PAA_NO_GATE32=1 at model creation: the backward epilogues evaluate gelu' themselves) — run on the GPU box with the diagnostic
library, the only one that reads the switch:  PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python tools/gate_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from paa_amd import arch as A, synth
from paa_amd.core import loss_helpers
from paa_amd.model import PaaModel
from paa_amd.training_utils import parser
from paa_amd.training_utils.pgd import PgdStepper


def main(steps=15, rounds=3):
    a, B, L = A.BASE, 32, 160000
    clean = torch.from_numpy(synth.clean_audio(B, L, seed=5)).cuda()
    texts = [("the quick brown fox jumps over a lazy dog and runs " * 4)[:150] for _ in range(B)]
    args = parser.create_arg_parser().parse_args(["--norm_type", "snr", "--snr_db", "40", "--lr", "1e-4", "--optimizer_type", "pgd",
                                                  "--device", "cuda", "--dtype", "fp32"])
    labels = loss_helpers.make_labels(texts, None, args, B).to(device="cuda", dtype=torch.int32)
    p0 = (torch.from_numpy(synth.perturbation(L, seed=5)) * np.float32(2e-3)).cuda()
    st, grads = {}, {}
    for no in ("0", "1"):
        os.environ["PAA_NO_GATE32"] = no
        st[no] = PgdStepper(PaaModel(a, A.rule_weights(a), B, L, "fp32"), args, L)
        p = p0.clone()
        st[no].step(p, clean, labels, want_logits=False)
        torch.cuda.synchronize()
        grads[no] = st[no].grad.clone()
    d = (grads["0"] - grads["1"]).abs().max().item() / grads["1"].abs().max().item()
    flips = (torch.sign(grads["0"]) != torch.sign(grads["1"])).float().mean().item()
    print(f"gradient: max |diff| / max |g| = {d:.3e}, sign flips {flips:.3e}")
    best = {}
    for rnd in range(rounds + 1):
        for no in ("0", "1"):
            p = p0.clone()
            for _ in range(2):
                st[no].step(p, clean, labels, want_logits=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st[no].step(p, clean, labels, want_logits=False)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            if rnd:
                best.setdefault(no, []).append(round(ms, 3))
    print({("gelu' kept by the forward pass" if k == "0" else "raw pre-activation kept"): v for k, v in best.items()})


if __name__ == "__main__":
    main()
