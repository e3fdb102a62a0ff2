"""Latency / bandwidth of the projection path: every norm at the reference shape (1, L) — where it is
launch-latency-bound — and in the batched mode (32, L), reported as algorithmic GB/s (8*rows*L bytes: read p + write p,
plus 4*B*L for the clean-batch statistic of snr / tv) against the 8 TB/s HBM peak.  Run on the GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from paa_amd import _lib, runtime, synth
from paa_amd.training_utils import build, parser


def main(L=160000, B=32, iters=50):
    lib = _lib.lib()
    out = []
    clean = torch.from_numpy(synth.clean_audio(B, L)).cuda()
    for norm in ("l2", "linf", "snr", "tv", "min_max_freqs", "fletcher_munson", "max_phon"):
        args = parser.create_arg_parser().parse_args(["--norm_type", norm, "--snr_db", "40", "--device", "cuda"])
        for rows in (1, B):
            p = (torch.randn(rows, L, device="cuda") * 1e-2).contiguous()
            pr = runtime.get_proj(args, p.device, rows, L)
            pr.set_spl_thresh(build.init_phon_threshold_tensor(args))
            prm = runtime.params_of(args)
            st = _lib.stream_ptr()

            def call():
                _lib.check(lib.paa_project(pr.h, prm, _lib.ptr(p), rows, _lib.ptr(clean), B, L, st))
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            nbytes = 8 * rows * L + (4 * B * L if norm in ("snr", "tv") else 0)
            out.append(dict(norm=norm, rows=rows, us=round(us, 2), algorithmic_MB=round(nbytes / 1e6, 2),
                            GBps=round(nbytes / us / 1e3, 1), frac_of_8TBps=round(nbytes / us / 1e3 / 8000, 4)))
            print(out[-1], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
