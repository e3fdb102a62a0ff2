"""In-process A/B of the fp32-parity PGD step with the one-pass conv0 GroupNorm backward (default) against the two-pass
path (the ABI's test hook paa_test_option(0, 1), read per call) — run on the GPU box."""
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


def main(steps=15, rounds=3, dtype="fp32"):
    a, B, L = A.BASE, 32, 160000
    clean = torch.from_numpy(synth.clean_audio(B, L, seed=5)).cuda()
    texts = [("the quick brown fox jumps over a lazy dog and runs " * 4)[:150] for _ in range(B)]
    args = parser.create_arg_parser().parse_args(["--norm_type", "snr", "--snr_db", "40", "--lr", "1e-4", "--optimizer_type", "pgd",
                                                  "--device", "cuda", "--dtype", dtype])
    labels = loss_helpers.make_labels(texts, None, args, B).to(device="cuda", dtype=torch.int32)
    m = PaaModel(a, A.rule_weights(a), B, L, dtype)
    st = PgdStepper(m, args, L)
    p = (torch.from_numpy(synth.perturbation(L, seed=5)) * np.float32(2e-3)).cuda()
    best = {}
    for rnd in range(rounds + 1):
        for two in ("0", "1"):
            _lib.check(_lib.lib().paa_test_option(0, int(two)))
            for _ in range(2):
                st.step(p, clean, labels, want_logits=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step(p, clean, labels, want_logits=False)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            if rnd:
                best[two] = min(best.get(two, 1e9), ms)
            print(f"round {rnd} two_pass={two}: {ms:.3f} ms", flush=True)
    _lib.lib().paa_test_option(0, 0)
    print({("two_pass" if k == "1" else "one_pass"): round(v, 3) for k, v in best.items()})


if __name__ == "__main__":
    main(dtype=sys.argv[1] if len(sys.argv) > 1 else "fp32")
