"""The PGD inner step as one launch sequence on the device (SURVEY §8a S1-S6 + P0):

    forward + CTC + backward-to-waveform  ->  [all-reduce SUM over ranks]  ->  p += lr*sign(grad)  ->  projection

Data-parallel form: every rank holds the full universal perturbation and a shard of the
utterances; because HF's CTC reduction is 'sum', the global gradient is the sum of the shard
gradients, so ONE all-reduce (RCCL over xGMI via torch.distributed's "nccl" backend) per step of the packed
f32 vector

    [ grad (L) | loss, sum clean^2, TV(clean), wer_errors, wer_ref_words, clips, 0, 0 ]

suffices; every rank then applies the identical sign step and projection, so replicas stay bit-identical
without a broadcast.  ``sum clean^2`` / ``TV(clean)`` are there because project_snr / project_tv use
whole-GLOBAL-batch statistics (projections.py:11-35, 56-66); the SNR target norm also uses the global
``clean.numel()`` = ``L * sum of the ranks' batch sizes``: every rank writes its own clip count into slot 5 each step
(``paa_batch_stats``), the all-reduce sums it (a small integer, exact in f32) and ``paa_project_ext`` reads the sum on the
device — so ranks may hold different numbers of clips, in any step, without a collective of their own.  The WER counters
are the host-side bookkeeping of the PREVIOUS step (train.py:149-153 runs one step behind the GPU), reduced with the same
buffer.  The layout of the 8 slots is defined HERE (ST_*) and documented in include/paa_hip.h (paa_model_fwd_bwd, d_stats).
"""
from __future__ import annotations

import torch

from .. import _lib, runtime

FREQ_NORMS = ("fletcher_munson", "min_max_freqs", "max_phon")
N_STATS = 8
ST_LOSS, ST_SQ, ST_TV, ST_WER_ERR, ST_WER_REF, ST_CLIPS = 0, 1, 2, 3, 4, 5


class PgdStepper:
    def __init__(self, model, args, length: int, interp=None, spl_thresh=None, group=None, force_collective=False):
        """``force_collective``: run the packed all-reduce (and the global-statistics projection) even with a single rank —
        the way the one-GPU test box executes the RCCL branch (tests/test_gpu_rccl.py)."""
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
        self.collective = self.world > 1 or bool(force_collective)
        if self.collective and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError("force_collective needs an initialised torch.distributed process group")
        self.need_clean_stats = self.collective and any(n in ("snr", "tv") for n in self.norms)
        self._prm = []
        for n in self.norms:
            a = type("A", (), dict(vars(args)))()
            a.norm_type = n
            self._prm.append(runtime.params_of(a))
        self._wer_ring = [torch.zeros(2, dtype=torch.float32).pin_memory() for _ in range(4)] if self.collective else None
        self._wer_i = 0
        self._wer_next = (0.0, 0.0)

    # ---- bookkeeping carried by the packed vector -------------------------------------------------------------
    def set_wer_counts(self, errors: float, ref_words: float):
        """Host-side WER counters of the PREVIOUS step (train.py:149-153): the next ``step`` writes them behind the
        gradient, so its all-reduce sums them over ranks; read the global sums from ``stats[3:5]`` afterwards."""
        self._wer_next = (float(errors), float(ref_words))

    def _push_wer(self):
        """This rank's WER counters of the previous step go behind the gradient (pinned ring: the copy is asynchronous,
        so a slot must stay untouched until the GPU has consumed it)."""
        if not self.collective:
            return
        h = self._wer_ring[self._wer_i & 3]
        self._wer_i += 1
        h[0], h[1] = self._wer_next
        self._wer_next = (0.0, 0.0)
        self.stats[ST_WER_ERR:ST_WER_REF + 1].copy_(h, non_blocking=True)

    # ---- the two halves of a step (everything before / after the collective) ----------------------------------
    def _pre(self, p, clean, labels, want_logits=True, logits_out=None):
        B = clean.shape[0]
        out = {"grad": self.grad, "stats": self.stats}
        if logits_out is not None:
            out["logits"] = logits_out
        r = self.model.fwd_bwd(clean, p, labels, self.direction, want_grad=True, want_logits=want_logits, out=out)
        if self.need_clean_stats:
            with torch.cuda.device(self.dev):
                _lib.check(_lib.lib().paa_batch_stats(self.proj.h, _lib.ptr(clean), B, self.L, _lib.ptr(self.stats[ST_SQ:ST_TV + 1]),
                                                      _lib.ptr(self.stats[ST_CLIPS:ST_CLIPS + 1]), _lib.stream_ptr()))
        return r

    def _post(self, p, clean):
        lib, L, B = _lib.lib(), self.L, clean.shape[0]
        with torch.cuda.device(self.dev):
            st = _lib.stream_ptr()
            _lib.check(lib.paa_sign_step(_lib.ptr(p), _lib.ptr(self.grad), float(self.args.lr), L, st))   # train.py:160-161
            for n, prm in zip(self.norms, self._prm):                                                   # train.py:162
                if self.need_clean_stats and n in ("snr", "tv"):
                    _lib.check(lib.paa_project_ext(self.proj.h, prm, _lib.ptr(p), 1, _lib.ptr(self.stats[ST_SQ:ST_TV + 1]),
                                                   _lib.ptr(self.stats[ST_CLIPS:ST_CLIPS + 1]), 0.0, L, st))
                else:
                    _lib.check(lib.paa_project(self.proj.h, prm, _lib.ptr(p), 1, _lib.ptr(clean), B, L, st))

    def step(self, p: torch.Tensor, clean: torch.Tensor, labels: torch.Tensor, want_logits=True, logits_out=None):
        """In place on ``p`` (1, L).  Returns dict(loss: 0-d device tensor, summed over ALL ranks, logits)."""
        p = runtime.as_f32_cuda(p, "p")
        clean = runtime.as_f32_cuda(clean, "clean_audio")
        if p.numel() != self.L or clean.shape[-1] != self.L:
            raise ValueError(f"Loaded perturbation length {p.numel()} / clip length {clean.shape[-1]} != expected {self.L}")
        self._push_wer()
        r = self._pre(p, clean, labels, want_logits, logits_out)
        if self.collective:
            torch.distributed.all_reduce(self.packed, op=torch.distributed.ReduceOp.SUM, group=self.group)
        self._post(p, clean)
        r["loss"] = self.stats[ST_LOSS]
        return r

    def capture(self, p, clean, labels, logits_out=None):
        """Capture one step on fixed buffers into hipGraphs (the launch sequence allocates nothing and never
        synchronises, so it is capturable as is).  Returns (graph, result dict); ``graph.replay()`` re-runs the step
        in place on ``p`` with whatever ``clean`` / ``labels`` currently hold.  With several ranks the halves before
        and after the collective are two graphs and the all-reduce runs between their replays."""
        lab = labels.to(device=self.dev, dtype=torch.int32).contiguous()
        if logits_out is None:
            logits_out = torch.empty(clean.shape[0], self.model.frames, self.model.arch.vocab_size, device=self.dev)
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):                       # warm-up on the side stream, as torch's capture rules require
            self.step(p, clean, lab, logits_out=logits_out)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        if not self.collective:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                r = self.step(p, clean, lab, logits_out=logits_out)
            return g, r
        # No collective may be in flight while a capture is open (the process group's watchdog thread polls its events), and the
        # captures only guard THIS thread's launches: the RCCL call between them runs eagerly.
        torch.cuda.synchronize(self.dev)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            r = self._pre(p, clean, lab, True, logits_out)
        torch.distributed.all_reduce(self.packed, op=torch.distributed.ReduceOp.SUM, group=self.group)
        torch.cuda.synchronize(self.dev)
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            self._post(p, clean)
        r["loss"] = self.stats[ST_LOSS]
        return _SplitGraph(self, g1, g2), r


class _SplitGraph:
    """replay() = pre-collective graph, all-reduce of the packed vector, post-collective graph."""

    def __init__(self, stepper, g1, g2):
        self.stepper, self.g1, self.g2 = stepper, g1, g2

    def replay(self):
        self.stepper._push_wer()
        self.g1.replay()
        torch.distributed.all_reduce(self.stepper.packed, op=torch.distributed.ReduceOp.SUM, group=self.stepper.group)
        self.g2.replay()
