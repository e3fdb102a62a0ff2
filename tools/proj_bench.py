"""Latency / bandwidth of the projection path: every norm at the reference shape (1, L) — where it is
launch-latency-bound — and in the batched mode (32, L), reported as algorithmic GB/s (8*rows*L bytes: read p + write p,
plus 4*B*L for the clean-batch statistic of snr / tv) against the 8 TB/s HBM peak.  Run on the GPU box:

    python tools/proj_bench.py                 # every norm, JSON list on the last line
    rocprofv3 --kernel-trace --stats ... -- python3 tools/proj_bench.py --spectral   # the three FFT norms only

`roofline_block()` is what bench.py embeds as `roofline_fft`.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paa_amd import _lib, runtime, synth
from paa_amd.training_utils import build, parser

SPECTRAL = ("min_max_freqs", "fletcher_munson", "max_phon")
ALL = ("l2", "linf", "snr", "tv") + SPECTRAL
HBM_PEAK_GBPS = 8000.0


def measure(norms=ALL, rows_list=(1, 32), L=160000, B=32, iters=50, dev="cuda", verbose=True, in_place=False):
    """in_place=False times paa_project_to (reads p, writes a new tensor — what the reference's functions do, and the one
    fused launch of the FFT norms); in_place=True times paa_project (what the PGD step calls on its resident p)."""
    lib = _lib.lib()
    out = []
    clean = torch.from_numpy(synth.clean_audio(B, L)).to(dev)
    for norm in norms:
        args = parser.create_arg_parser().parse_args(["--norm_type", norm, "--snr_db", "40", "--device", str(dev)])
        for rows in rows_list:
            p = (torch.randn(rows, L, device=dev) * 1e-2).contiguous()
            q = torch.empty_like(p)
            pr = runtime.get_proj(args, p.device, rows, L)
            pr.set_spl_thresh(build.init_phon_threshold_tensor(args))
            prm = runtime.params_of(args)
            st = _lib.stream_ptr()

            def call():
                if in_place:
                    _lib.check(lib.paa_project(pr.h, prm, _lib.ptr(p), rows, _lib.ptr(clean), B, L, st))
                else:
                    _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(p), _lib.ptr(q), rows, _lib.ptr(clean), B, L, st))
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
            out.append(dict(norm=norm, rows=rows, in_place=in_place, us=round(us, 2), algorithmic_MB=round(nbytes / 1e6, 2),
                            GBps=round(nbytes / us / 1e3, 1), frac_of_8TBps=round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4)))
            if verbose:
                print(out[-1], file=sys.stderr, flush=True)
    return out


def roofline_block(dev="cuda", L=160000, B=32):
    """FFT / projection path roofline for the bench line: the fused STFT -> per-bin projection -> iSTFT kernel of each
    spectral norm in the batched (32, L) mode (HBM-bound by its algorithmic bytes 8*rows*L) and its latency at the
    reference shape (1, L).  Timed with HIP events on the launch stream, 50 calls each."""
    rows = measure(SPECTRAL, (1, B), L, B, 50, dev, verbose=False)
    inpl = measure(SPECTRAL, (1,), L, B, 50, dev, verbose=False, in_place=True)
    blk = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "kernel": "k_spec_run<OP, 12> at (32, L), k_spec_fused<OP, ., 8> at (1, L) (STFT -> per-bin op -> iSTFT + overlap-add, one launch; + scale launch for fletcher_munson)",
           "algorithmic_bytes": 8 * B * L, "shape": [B, L], "call": "paa_project_to (out of place)", "norms": {}}
    for r in rows:
        d = blk["norms"].setdefault(r["norm"], {})
        if r["rows"] == B:
            d.update(achieved=r["GBps"], frac=r["frac_of_8TBps"], us=r["us"])
        else:
            d["latency_us_1xL"] = r["us"]
    for r in inpl:
        blk["norms"][r["norm"]]["latency_us_1xL_in_place"] = r["us"]
    worst = min(blk["norms"].values(), key=lambda d: d["achieved"])
    blk["achieved"], blk["frac"] = worst["achieved"], worst["frac"]
    blk["traffic"] = None
    return blk


if __name__ == "__main__":
    if "--batch-only" in sys.argv:          # counter passes (tools/fft_pmc.sh): the (32, L) launches only
        res = measure(SPECTRAL, rows_list=(32,), iters=10)
        print(json.dumps(res))
        sys.exit(0)
    res = measure(SPECTRAL if "--spectral" in sys.argv else ALL)
    res += measure(SPECTRAL if "--spectral" in sys.argv else ALL, in_place=True)
    print(json.dumps(res))
