"""The PGD inner step as one launch sequence on the device (SURVEY §8a S1-S6 + P0):

    forward + CTC + backward-to-waveform  ->  [all-reduce SUM over ranks]  ->  p += lr*sign(grad)  ->  projection

Data-parallel form: every rank holds the full universal perturbation and a shard of the
utterances; because HF's CTC reduction is 'sum', the global gradient is the sum of the shard
gradients, so ONE all-reduce (RCCL over xGMI via torch.distributed's "nccl" backend) of the packed
vector [grad(L) | loss, sum clean^2, TV(clean), 0...] per step suffices; every rank then applies the
identical sign step and projection, so replicas stay bit-identical without a broadcast.
"""
from __future__ import annotations

import torch

from .. import _lib, runtime

FREQ_NORMS = ("fletcher_munson", "min_max_freqs", "max_phon")
N_STATS = 8


class PgdStepper:
    def __init__(self, model, args, length: int, interp=None, spl_thresh=None, group=None):
        self.model, self.args, self.L = model, args, int(length)
        self.dev = model.device
        self.norms = str(args.norm_type).split("+")
        for n in self.norms:
            if n not in _lib.NORM_IDS:
                raise ValueError(f"Unknown norm_type: {n!r}")                  # train.py:98
        self.direction = +1 if args.attack_mode == "untargeted" else -1          # train.py:124
        self.packed = torch.zeros(self.L + N_STATS, dtype=torch.float32, device=self.dev)
        self.grad = self.packed[: self.L].view(1, self.L)
        self.stats = self.packed[self.L:]
        self.proj = runtime.get_proj(args, self.dev, 1, self.L, interp)
        if spl_thresh is not None:
            self.proj.set_spl_thresh(spl_thresh)
        self.group = group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(group)
        self._prm = []
        for n in self.norms:
            a = type("A", (), dict(vars(args)))()
            a.norm_type = n
            self._prm.append(runtime.params_of(a))

    def step(self, p: torch.Tensor, clean: torch.Tensor, labels: torch.Tensor, want_logits=True, logits_out=None):
        """In place on ``p`` (1, L).  Returns dict(loss: 0-d device tensor, summed over ALL ranks, logits)."""
        L = self.L
        B = clean.shape[0]
        lib = _lib.lib()
        out = {"grad": self.grad, "stats": self.stats}
        if logits_out is not None:
            out["logits"] = logits_out
        r = self.model.fwd_bwd(clean, p, labels, self.direction, want_grad=True, want_logits=want_logits, out=out)
        need_clean_stats = self.world > 1 and any(n in ("snr", "tv") for n in self.norms)
        st = _lib.stream_ptr()
        with torch.cuda.device(self.dev):
            if need_clean_stats:
                _lib.check(lib.paa_batch_stats(self.proj.h, _lib.ptr(clean), B, L, _lib.ptr(self.stats[1:3]), st))
            if self.world > 1:
                torch.distributed.all_reduce(self.packed, op=torch.distributed.ReduceOp.SUM, group=self.group)
            st = _lib.stream_ptr()
            _lib.check(lib.paa_sign_step(_lib.ptr(p), _lib.ptr(self.grad), float(self.args.lr), L, st))   # train.py:160-161
            for n, prm in zip(self.norms, self._prm):                                                   # train.py:162
                if need_clean_stats and n in ("snr", "tv"):
                    _lib.check(lib.paa_project_ext(self.proj.h, prm, _lib.ptr(p), 1, _lib.ptr(self.stats[1:3]),
                                                   float(B * self.world * L), L, st))
                else:
                    _lib.check(lib.paa_project(self.proj.h, prm, _lib.ptr(p), 1, _lib.ptr(clean), B, L, st))
        r["loss"] = self.stats[0]
        return r

    def capture(self, p, clean, labels, logits_out=None):
        """Capture one step on fixed buffers into a hipGraph (the launch sequence allocates nothing and never
        synchronises, so it is capturable as is).  Returns (graph, result dict); ``graph.replay()`` re-runs the step
        in place on ``p`` with whatever ``clean`` / ``labels`` currently hold.  Single-rank only."""
        if self.world > 1:
            raise RuntimeError("graph capture of the data-parallel step is not enabled")
        lab = labels.to(device=self.dev, dtype=torch.int32).contiguous()
        if logits_out is None:
            logits_out = torch.empty(clean.shape[0], self.model.frames, self.model.arch.vocab_size, device=self.dev)
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):                       # warm-up on the side stream, as torch's capture rules require
            self.step(p, clean, lab, logits_out=logits_out)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            r = self.step(p, clean, lab, logits_out=logits_out)
        return g, r
