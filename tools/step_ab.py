"""In-process A/B of the whole PGD step under two paa_gemm kernel selections (run on the GPU box): ring kernels in their
automatic selection (config 0) against the register-staged kernels only (config 1), interleaved rounds on one device —
the only comparison that survives the +-4 % spread between gpurun boxes.  The switches exist only in the diagnostic library:

    PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python psychoacoustic-adverserial-attacks_amd/build_ext.py      # here
    gpurun -- 'PAA_EXTRA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python tools/step_ab.py'
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from paa_amd import _lib, arch as A, synth
from paa_amd.core import loss_helpers
from paa_amd.model import PaaModel
from paa_amd.training_utils import build, parser
from paa_amd.training_utils.pgd import PgdStepper


# auto = automatic selection, reg = register-staged kernels only; nosq = without the 256-column rings; nor2 = two-stage 256-column
# rings instead of the separate operand rings (gemm_ring2.hip); nobil = planar weight planes instead of the interleaved copy; nokg =
# plain K order in the convs
VARIANTS = ("auto", "auto nobil", "auto nor2", "reg")


def main(steps=15, rounds=3):
    lib = _lib.lib()
    a, B, L = A.BASE, 32, 160000
    clean = torch.from_numpy(synth.clean_audio(B, L, seed=5)).cuda()
    texts = [("the quick brown fox jumps over a lazy dog and runs " * 4)[:150] for _ in range(B)]
    for dtype in ("fp32", "bf16"):
        args = parser.create_arg_parser().parse_args(["--norm_type", "snr", "--snr_db", "40", "--lr", "1e-4", "--optimizer_type", "pgd",
                                                      "--device", "cuda", "--dtype", dtype])
        labels = loss_helpers.make_labels(texts, None, args, B).to(device="cuda", dtype=torch.int32)
        m = PaaModel(a, A.rule_weights(a), B, L, dtype)
        st = PgdStepper(m, args, L)
        p = (torch.from_numpy(synth.perturbation(L, seed=5)) * np.float32(2e-3)).cuda()
        best = {}
        for rnd in range(rounds + 1):
            for cfg in VARIANTS:
                os.environ["PAA_K_GROUP"] = "0" if "nokg" in cfg else "1"
                os.environ["PAA_NO_SQ"] = "1" if "nosq" in cfg else "0"
                os.environ["PAA_NO_R2"] = "1" if "nor2" in cfg else "0"
                os.environ["PAA_NO_BIL"] = "1" if "nobil" in cfg else "0"
                lib.paa_gemm_config(1 if "reg" in cfg else 0)
                for _ in range(2):
                    st.step(p, clean, labels, want_logits=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    st.step(p, clean, labels, want_logits=False)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / steps
                if rnd:
                    best.setdefault(cfg, []).append(ms)
        lib.paa_gemm_config(0)
        print(dtype, {c: [round(x, 3) for x in v] for c, v in best.items()}, flush=True)
        del st, m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
