"""In-process A/B of the PGD step launched eagerly against the same step replayed from a captured hipGraph
(PgdStepper.capture), alternating rounds on ONE device at the headline shape (base architecture, 32 x 10 s, snr 40).
BASELINE.md section 3 / SURVEY 8d name hipGraph replay as the timing protocol; this is the measurement that decides whether
bench.py replays by default.  Prints one JSON line per arithmetic mode."""
import json
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


def main(steps=32, rounds=3, modes=("fp32", "bf16")):
    a, B, L = A.BASE, 32, 160000
    clean = torch.from_numpy(synth.clean_audio(B, L, seed=5)).cuda()
    texts = [("the quick brown fox jumps over a lazy dog and runs " * 4)[:150] for _ in range(B)]
    for dtype in modes:
        args = parser.create_arg_parser().parse_args(["--norm_type", "snr", "--snr_db", "40", "--lr", "1e-4", "--optimizer_type", "pgd",
                                                      "--device", "cuda", "--dtype", dtype])
        labels = loss_helpers.make_labels(texts, None, args, B).to(device="cuda", dtype=torch.int32)
        m = PaaModel(a, A.rule_weights(a), B, L, dtype)
        st = PgdStepper(m, args, L)
        p0 = (torch.from_numpy(synth.perturbation(L, seed=5)) * np.float32(2e-3)).cuda()
        p = p0.clone()
        logits = torch.empty(B, m.frames, a.vocab_size, device="cuda")
        graph, _ = st.capture(p, clean, labels, logits_out=logits)
        # same trajectory from the same start: eager and replayed steps must agree bit for bit
        p.copy_(p0)
        for _ in range(3):
            st.step(p, clean, labels, logits_out=logits)
        pe = p.clone()
        p.copy_(p0)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(pe, p))
        ms = {"eager": [], "graph": []}
        for rnd in range(rounds + 1):
            for kind in ("eager", "graph"):
                run = (lambda: st.step(p, clean, labels, logits_out=logits)) if kind == "eager" else graph.replay
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    run()
                torch.cuda.synchronize()
                if rnd:
                    ms[kind].append(round((time.perf_counter() - t0) * 1e3 / steps, 3))
        print(json.dumps({"dtype": dtype, "steps_per_round": steps, "ms_per_step": ms, "replay_equals_eager_bitwise": same}), flush=True)
        del graph, st, m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
