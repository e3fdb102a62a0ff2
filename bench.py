#!/usr/bin/env python3
"""PGD steps/s on synthetic fixed-length 16 kHz clips (BASELINE.json metric, configs[1]).

    python bench.py --gpus N --steps K --warmup W        # N > 1: this process starts its N ranks itself (fresh child
                                                          # processes, before any GPU call) and relays rank 0's JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            # also fine: RANK / WORLD_SIZE / MASTER_* come from the launcher

Workload (config.workload): untargeted PGD, --norm_type snr --snr_db 40, Wav2Vec2-base architecture with
rule-generated weights (no checkpoint exists offline), 32 x 10 s clips per GPU, lr 1e-4.  One "step" is one
full pass of the hot path over one batch: compose+clamp -> Wav2Vec2 forward -> CTC -> backward to the
waveform -> [all-reduce over ranks] -> sign step -> projection, plus the per-step loss / greedy-decode
read-back the reference does (train.py:146-153), pipelined one step behind so it stays off the critical path.
Weak scaling: the per-GPU batch is fixed, `value` counts 32-clip steps over ALL ranks per second.

Two arithmetic modes are timed in the same invocation, W warm-up + exactly K timed steps each:
  * fp32-parity (split-bf16, three MFMA passes) — the HEADLINE (`value`, `dtype`, `roofline`): the reference computes in
    float32 and this is the mode that reproduces its sign(grad) on the goldens (tests/test_gpu_model.py);
  * bf16 — reported beside it in the `bf16` block with its own roofline and its MEASURED disagreement with the
    fp32-parity gradient on this very batch (sign-flip rate, cosine).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VARIANTS = {v: (f"k_gemm_bf<{'256' if v & 32 else '128'},{'64' if v & 8 else '128'},{'split' if v & 4 else 'bf16'}>" if v & 16 else
                f"k_gemm<{'64' if v & 8 else '128'},{'split' if v & 4 else 'bf16'},A{'k' if v & 2 else 'm'},B{'k' if v & 1 else 'n'}>")
            for v in range(64)}
VARIANTS.update({51: "k_gemm_bf<256,128,bf16>", 55: "k_gemm_bf<256,128,split>", 59: "k_gemm_bf<192,128,bf16>",
                 40: "k_gemm_win<bf16>", 44: "k_gemm_win<split>", 60: "k_gemm_ring<192,128,bf16>", 61: "k_gemm_ring<192,128,split>", 56: "k_gemm_ring<256,256,bf16>",
                 57: "k_gemm_ring<256,256,split>", 52: "k_gemm_ring<192,256,bf16>", 53: "k_gemm_ring<192,256,split>",
                 48: "k_gemm_ring<256,128,bf16>", 49: "k_gemm_ring<256,128,split>"})
DTYPE_NAME = {"fp32": "bf16x3 (split-bf16 hi+lo, three MFMA passes, f32 accumulate: fp32-parity)", "bf16": "bf16"}
MFMA_PEAK = 2500.0       # dense bf16 TFLOP/s (MI355X_MICROARCH.md); a split-mode product issues three such MFMAs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)     # >= 2 s of timed region in the headline mode (38 ms per step)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--arch", default="base", choices=["base", "large-lv60", "tiny"])
    ap.add_argument("--modes", default="fp32,bf16", help="comma list of fp32 (= fp32-parity, split-bf16) and bf16; the FIRST is the headline")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"], help="time only this mode (same as --modes X)")
    ap.add_argument("--norm_type", default="snr")
    ap.add_argument("--snr_db", type=float, default=40.0)
    ap.add_argument("--label_tokens", type=int, default=150)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_batch", type=int, default=2, help="clips in the bounded CPU-baseline sample")
    ap.add_argument("--cpu_steps", type=int, default=2, help="timed oracle steps after one warm-up step")
    ap.add_argument("--no_baseline_faithful", action="store_true", help="skip the weight-gradients-kept CPU variant")
    ap.add_argument("--no_prof", action="store_true", help="skip the per-launch GEMM event timing")
    ap.add_argument("--prof_steps", type=int, default=1,
                    help="number of timed steps (the last ones) whose GEMM launches carry HIP events; 0 = every timed step")
    ap.add_argument("--eager", action="store_true",
                    help="launch every step eagerly; default: the step is replayed from captured hipGraphs (BASELINE.md section 3 protocol; "
                         "same kernels, bit-identical results, same time — profiles/r3_graph_vs_eager_ab.txt), except the last "
                         "--prof_steps timed steps, which are launched eagerly so that HIP events can bracket every GEMM")
    ap.add_argument("--pmc_json", default=None, help="tools/pmc_summary.py output of THIS commit: fills roofline.traffic instead of the live counter passes")
    ap.add_argument("--no_pmc", action="store_true",
                    help="skip the live HBM-traffic measurement (two short rocprofv3 --pmc child runs of the headline mode BEFORE this process "
                         "touches the GPU; N = 1 only); roofline.traffic is then null unless --pmc_json is given")
    ap.add_argument("--no_fft_bench", action="store_true", help="skip the FFT / projection path timing (roofline_fft block)")
    return ap.parse_args()


def live_pmc(ar, mode):
    """HBM bytes per launch of every GEMM variant of the headline mode, measured NOW: two child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no other trace domain; KB units x 1024,
    FETCH_SIZE doubled: MI355X_MICROARCH.md, HBM / rocprofv3 — the arithmetic of tools/pmc_summary.py), same shape flags, 1 warm-up +
    2 timed eager steps.  Started before this process makes any GPU call.  -> ({variant: bytes per launch}, note) or (None, why)."""
    import glob
    import importlib.util
    import shutil
    import tempfile
    if any(k.startswith(("ROCPROFILER_", "ROCP_TOOL", "ROCPROF_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(here, "tools", "pmc_summary.py"))
        ps = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ps)
    except Exception as e:                               # noqa: BLE001
        return None, f"tools/pmc_summary.py: {e}"
    tmp = tempfile.mkdtemp(prefix="paa_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                            "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = "/tmp"
    child = [sys.executable, os.path.abspath(__file__), "--dtype", mode, "--steps", "2", "--warmup", "1", "--eager", "--no_cpu_baseline",
             "--no_fft_bench", "--no_prof", "--no_pmc", "--batch", str(ar.batch), "--seconds", str(ar.seconds), "--arch", ar.arch,
             "--norm_type", ar.norm_type, "--snr_db", str(ar.snr_db), "--label_tokens", str(ar.label_tokens)]
    tot, cnt = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            # own session: a pass that overruns is killed as a whole (profiler AND the profiled child), so nothing of it is left on
            # the GPU when the timed region of this process starts
            pr = subprocess.Popen([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "runc", "--"] + child,
                                  cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                _, err = pr.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                return None, f"rocprofv3 --pmc {counter} pass exceeded 180 s and was killed"
            if pr.returncode != 0 or not glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {pr.returncode}): {err[-300:]}"
            t, c = ps.load(d, counter)
            if not any(ps.variant_of(nm) for nm in t):
                return None, f"the --pmc {counter} pass recorded no GEMM launch (child stderr tail: {err[-300:]!r})"
            scale = 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
            for name, v in t.items():
                lab = ps.variant_of(name)
                if lab:
                    tot[lab] = tot.get(lab, 0.0) + v * scale
                    if counter == "FETCH_SIZE":
                        cnt[lab] = cnt.get(lab, 0) + c[name]
    except Exception as e:                               # noqa: BLE001
        return None, f"live counter passes: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {lab: int(tot[lab] / cnt[lab]) for lab in tot if cnt.get(lab)}
    return out, ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of the same command "
                 "(--steps 2 --warmup 1 --eager), KB x 1024, FETCH_SIZE x 2 (gfx950)")


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this one, which has
    made no GPU call (no exec of a GPU-initialised process), relay rank 0's stdout, fail loudly with the failing rank's
    stderr.  Rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [None] * n
    import threading

    def drain(i):
        outs[i] = procs[i].communicate()
    th = [threading.Thread(target=drain, args=(i,)) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    bad = [i for i, pr in enumerate(procs) if pr.returncode != 0]
    if bad:
        for i in bad:
            sys.stderr.write(f"---- rank {i} exited with {procs[i].returncode}; stderr tail ----\n{outs[i][1][-4000:]}\n")
        return 1
    sys.stderr.write(outs[0][1][-2000:])
    sys.stdout.write(outs[0][0])
    sys.stdout.flush()
    return 0


def make_transcript(i: int, n_chars: int, seed: int) -> str:
    from paa_amd import synth
    words = ["the", "quick", "brown", "fox", "jumps", "over", "a", "lazy", "dog", "and", "runs", "away", "home"]
    u = synth.uniform(synth.key_of(f"txt{i}", seed), n_chars)
    s, k = "", 0
    while len(s) < n_chars:
        s += words[int(u[k % n_chars] * len(words))] + " "
        k += 1
    return s[:n_chars].rstrip().ljust(n_chars, "a")


def cpu_baseline(a, args_ns, L, batch, label_tokens, seed, steps, weight_grads):
    """The oracle's PGD step (torch CPU, float32, all host cores) on a bounded sample of the same workload: one warm-up
    step, then `steps` timed ones.  weight_grads: keep the model parameters' requires_grad as the reference does
    (train.py:118 only calls .eval(), so train.py:158 also back-propagates into every weight — the "faithful" variant)."""
    import numpy as np
    import torch
    from oracle import pgd as opgd, wav2vec2 as OW
    from paa_amd import arch as A, synth
    sd = OW.to_torch(A.rule_weights(a))
    if weight_grads:
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    clean = torch.from_numpy(synth.clean_audio(batch, L, seed=seed))
    p = torch.from_numpy(synth.perturbation(L, seed=seed) * np.float32(1e-3))
    texts = [make_transcript(i, label_tokens, seed) for i in range(batch)]
    labels = opgd.make_labels(texts, args_ns, batch)
    p = opgd.pgd_step(sd, a, args_ns, clean, labels, p)["p_new"]          # warm-up (thread pools, allocator, oneDNN primitives)
    times = []
    for _ in range(steps):
        if weight_grads:
            for v in sd.values():
                if v.is_floating_point():
                    v.grad = None
        t0 = time.perf_counter()
        p = opgd.pgd_step(sd, a, args_ns, clean, labels, p)["p_new"]
        times.append(time.perf_counter() - t0)
    return times


def main():
    ar = parse()
    if ar.dtype:
        ar.modes = ar.dtype
    modes = [m for m in ar.modes.split(",") if m]
    for m in modes:
        if m not in ("fp32", "bf16"):
            raise SystemExit(f"unknown mode {m!r}")
    if "WORLD_SIZE" not in os.environ and ar.gpus > 1:
        sys.exit(spawn_ranks(ar.gpus))
    live_traffic, live_note = None, "not measured (N > 1, --no_pmc or --no_prof)"
    if ar.gpus == 1 and os.environ.get("WORLD_SIZE", "1") == "1" and not ar.no_pmc and not ar.no_prof and not ar.pmc_json:
        live_traffic, live_note = live_pmc(ar, modes[0])          # child processes; this one has not touched the GPU yet
        if live_traffic is None:
            sys.stderr.write(f"[bench] roofline.traffic stays null: {live_note}\n")

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ar.gpus:
        raise SystemExit(f"--gpus {ar.gpus} but WORLD_SIZE={world}")
    ndev = max(torch.cuda.device_count(), 1)
    local = local % ndev                       # one rank per GPU; the modulo only matters for single-GPU rehearsals
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("PAA_DIST_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" for rehearsals on one GPU
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    import contextlib
    import __graft_entry__
    if rank == 0:
        with contextlib.redirect_stdout(sys.stderr):      # stdout carries the one JSON line and nothing else
            __graft_entry__.build()
    if world > 1:
        torch.distributed.barrier()
    from paa_amd import _lib, arch as A, synth
    from paa_amd.core import loss_helpers
    from paa_amd.model import PaaModel
    from paa_amd.training_utils import build, parser
    from paa_amd.training_utils.pgd import PgdStepper, ST_WER_ERR, ST_WER_REF

    a = {"base": A.BASE, "large-lv60": A.LARGE_LV60, "tiny": A.tiny()}[ar.arch]
    L = int(round(ar.seconds * 16000))
    B = ar.batch
    lib = _lib.lib()

    def make_args(dtype):
        return parser.create_arg_parser().parse_args(["--norm_type", ar.norm_type, "--snr_db", str(ar.snr_db), "--lr", "1e-4",
                                                      "--attack_mode", "untargeted", "--optimizer_type", "pgd", "--seed", "5",
                                                      "--device", str(dev), "--dtype", dtype])

    # two resident synthetic batches per rank, disjoint clips across ranks
    NB = 2
    cleans, labels, texts = [], [], []
    args0 = make_args(modes[0])
    for s in range(NB):
        first = (s * world + rank) * B
        cleans.append(torch.from_numpy(synth.clean_audio(B, L, seed=5, first_clip=first)).to(dev))
        tx = [make_transcript(first + b, ar.label_tokens, 5) for b in range(B)]
        texts.append(tx)
        labels.append(loss_helpers.make_labels(tx, None, args0, B).to(device=dev, dtype=torch.int32))
    # p0 ~ N(0,1) projected once (build.py:301-304); the global batch statistic it needs is taken from this rank's first batch
    p0 = build.init_perturbation(args0, L, None, None, cleans[0]).detach().clone()
    if world > 1:
        torch.distributed.broadcast(p0, src=0)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def time_mode(dtype):
        """W warm-up + exactly K timed steps of one arithmetic mode -> dict(value, ms_per_step, roofline, ...)."""
        args = make_args(dtype)
        model = PaaModel(a, A.rule_weights(a), B, L, dtype, dev)
        stepper = PgdStepper(model, args, L, None, build.init_phon_threshold_tensor(args))
        p = p0.clone()
        logits_buf = [torch.empty(B, model.frames, a.vocab_size, device=dev) for _ in range(2)]
        ids_dev = [torch.empty(B, model.frames, dtype=torch.int16, device=dev) for _ in range(2)]
        ids_host = [torch.empty(B, model.frames, dtype=torch.int16).pin_memory() for _ in range(2)]
        st_host = [torch.empty(8).pin_memory() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        book = {"loss": [], "wer": [], "wer_global": []}
        no_prof = ar.no_prof
        graphs = None
        if not ar.eager:
            graphs = [stepper.capture(p, cleans[j], labels[j], logits_out=logits_buf[j % 2]) for j in range(NB)]
            p.copy_(p0)                              # capture ran the step for real: restore the starting point
        # the last n_prof timed steps run eagerly (events around every GEMM cannot be recorded inside a replay)
        n_prof = 0 if (no_prof or rank != 0) else (ar.steps if ar.prof_steps <= 0 else min(ar.prof_steps, ar.steps))
        prof_from = ar.warmup + ar.steps - n_prof
        replayed = [0]

        def launch(i):
            k = i % 2
            if graphs is not None and not (n_prof and i >= prof_from):
                graphs[i % NB][0].replay()
                replayed[0] += 1
                r = graphs[i % NB][1]
            else:
                r = stepper.step(p, cleans[i % NB], labels[i % NB], want_logits=True, logits_out=logits_buf[k])
            # greedy ids (torch.argmax of loss_helpers.py:26) by the library's own kernel; ids + the stats tail go home async
            _lib.check(lib.paa_argmax_ids(_lib.ptr(r["logits"]), B * model.frames, a.vocab_size, _lib.ptr(ids_dev[k]), _lib.stream_ptr()))
            ids_host[k].copy_(ids_dev[k], non_blocking=True)
            st_host[k].copy_(stepper.stats, non_blocking=True)
            done[k].record()

        def collect(i):
            """Host side of train.py:146-153 for step i: loss value + WER of the greedy decode vs ground truth."""
            k = i % 2
            done[k].synchronize()
            pred = [t.lower() for t in loss_helpers.greedy_decode_ids(ids_host[k].tolist())]
            e, w = loss_helpers.wer_counts(pred, loss_helpers.clean_transcripts(texts[i % NB]))
            stepper.set_wer_counts(e, w)                     # summed over ranks by the next step's all-reduce
            book["loss"].append(float(st_host[k][0]))
            book["wer"].append(e / max(w, 1))
            if world > 1 and float(st_host[k][ST_WER_REF]) > 0:
                book["wer_global"].append(float(st_host[k][ST_WER_ERR]) / float(st_host[k][ST_WER_REF]))

        def run(n, start, on_step=None):
            for i in range(start, start + n):
                if on_step is not None:
                    on_step(i)
                launch(i)
                if i > start:
                    collect(i - 1)
            collect(start + n - 1)

        if ar.warmup > 0:
            run(ar.warmup, 0)
        prof_on = (not no_prof) and rank == 0
        # Per-launch HIP events around every GEMM are recorded on the LAST `--prof_steps` timed steps only.  The first timing
        # event recorded on a stream switches its HIP queue to profiled dispatch for the rest of the process, which slows
        # every later launch (~4 %); recording at the end keeps the steps before it unperturbed while the sample is
        # still taken live inside the timed region.  --prof_steps 0: every timed step.
        if prof_on:
            _lib.check(lib.paa_prof_enable(4096 * n_prof))
            _lib.check(lib.paa_prof_pause(1))
        sync()
        t0 = time.perf_counter()
        run(ar.steps, ar.warmup, (lambda i: lib.paa_prof_pause(0) if i == prof_from else None) if prof_on else None)
        sync()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax[0])

        passes = 3 if dtype == "fp32" else 1
        roofline = None
        if prof_on:
            out = (C.c_double * 256)()
            _lib.check(lib.paa_prof_read(out))
            lib.paa_prof_enable(0)
            rows = [(v, out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]) for v in range(64) if out[4 * v] > 0]
            rows.sort(key=lambda r: -r[2])
            if rows:
                v, n, ms, fl, by = rows[0]
                achieved = fl / (ms * 1e-3) / 1e12           # ALGORITHMIC flops (2 M N K) per second
                traffic = None
                if ar.pmc_json:                              # HBM bytes per launch from separate rocprofv3 --pmc passes of this commit
                    try:
                        with open(ar.pmc_json) as f:
                            pj = json.load(f)
                        for k in pj["kernels"]:
                            if k.get("variant") == VARIANTS[v] and pj.get("dtype") == dtype:
                                traffic = k["hbm_bytes_per_launch"]
                    except Exception:
                        traffic = None
                traffic_source = "--pmc_json" if traffic is not None else None
                if traffic is None and live_traffic is not None and dtype == modes[0]:
                    if VARIANTS[v] in live_traffic:
                        traffic, traffic_source = live_traffic[VARIANTS[v]], live_note
                    else:
                        sys.stderr.write(f"[bench] roofline.traffic stays null: the counter passes saw no launch labelled {VARIANTS[v]} "
                                         f"(labels: {sorted(live_traffic)})\n")
                roofline = {"bound": "mfma", "kernel": VARIANTS[v], "achieved": round(achieved, 2), "peak": MFMA_PEAK, "unit": "TFLOP/s",
                            "frac": round(achieved / MFMA_PEAK, 4), "traffic": traffic, "traffic_source": traffic_source,
                            "traffic_over_algorithmic": (round(traffic / (by / n), 3) if traffic else None), "launches": int(n),
                            "avg_launch_us": round(ms * 1e3 / n, 2), "algorithmic_bytes_per_launch": int(by / n),
                            "algorithmic_GBps": round(by / (ms * 1e-3) / 1e9, 1),
                            "mfma_passes_per_product": passes, "mfma_issue_frac": round(passes * achieved / MFMA_PEAK, 4),
                            "sampled_steps": n_prof, "share_of_step": round(ms * 1e-3 / (dt * max(n_prof, 1) / ar.steps), 4),
                            "all_gemm_variants": [{"kernel": VARIANTS[r[0]], "launches": int(r[1]), "ms": round(r[2], 3),
                                                   "tflops": round(r[3] / (r[2] * 1e-3) / 1e12, 2),
                                                   "algorithmic_MB_per_launch": round(r[4] / r[1] / 1e6, 1)} for r in rows]}
        fl_step = 2.0 * a.fwd_flops_per_clip(L) * B
        res = {"value": round(world * ar.steps / dt, 4), "ms_per_step": round(1e3 * dt / ar.steps, 3), "timed_region_s": round(dt, 3),
               "dtype": DTYPE_NAME[dtype], "model_tflops_achieved_per_gpu": round(fl_step / (dt / ar.steps) / 1e12, 2),
               "step_frac_of_mfma_peak": round(passes * fl_step / (dt / ar.steps) / 1e12 / MFMA_PEAK, 4),
               "hip_graph": bool(graphs), "hip_graph_replayed_steps": replayed[0] - (ar.warmup if graphs else 0), "last_loss": book["loss"][-1], "last_wer": book["wer"][-1], "roofline": roofline}
        if book["wer_global"]:
            res["last_wer_all_ranks"] = book["wer_global"][-1]
        # the gradient of ONE more step from the common starting point, for the cross-mode comparison below
        pc = p0.clone()
        stepper.step(pc, cleans[0], labels[0], want_logits=False)
        torch.cuda.synchronize()
        res["_grad"] = stepper.grad.detach().clone()
        res["_fl_step"] = fl_step
        return res

    import gc
    results = {}
    for m in modes:
        results[m] = time_mode(m)
        gc.collect()                                     # the mode's model arena (hipMalloc in the library) goes with its handle
        torch.cuda.empty_cache()

    if rank == 0:
        head = results[modes[0]]
        fl_step = head.pop("_fl_step")
        res = {
            "metric": "pgd_steps_per_sec", "value": head["value"],
            "unit": f"steps/s (one step = PGD step on {B} x {ar.seconds:g} s clips; counted over all ranks)",
            "n_gpus": world, "steps": ar.steps, "warmup": ar.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": f"untargeted PGD, norm_type {ar.norm_type} snr_db {ar.snr_db:g}, Wav2Vec2-{ar.arch} "
                                   f"(rule-generated weights), {B}x{ar.seconds:g}s 16 kHz clips per GPU, {ar.label_tokens}-token labels",
                       "global_batch": B * world, "per_gpu_batch": B, "samples_per_clip": L, "parallelism": f"dp{world}",
                       "collective": None if world == 1 else f"{backend} all_reduce(SUM) of grad(L)+8 floats per step",
                       "model_tflop_per_step_per_gpu": round(fl_step / 1e12, 3),
                       "model_tflops_achieved_per_gpu": head["model_tflops_achieved_per_gpu"],
                       "timed_region_s": head["timed_region_s"],
                       "hip_graph": head["hip_graph"], "hip_graph_replayed_steps": head["hip_graph_replayed_steps"], "last_loss": head["last_loss"], "last_wer": head["last_wer"]},
            "roofline": head["roofline"],
        }
        if "last_wer_all_ranks" in head:
            res["config"]["last_wer_all_ranks"] = head["last_wer_all_ranks"]
        for m in modes[1:]:
            o = results[m]
            o.pop("_fl_step")
            g1, g0 = o.pop("_grad").double().flatten(), head["_grad"].double().flatten()
            nz = (g0 != 0) & (g1 != 0)
            flips = float(((torch.sign(g0) != torch.sign(g1)) & nz).sum() / max(int(nz.sum()), 1))
            cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()).clamp_min(1e-300))
            o["vs_headline_mode"] = {"grad_sign_flip_rate": round(flips, 6), "grad_cosine": round(cos, 6),
                                     "meaning": "every flipped sign moves that sample of p the wrong way by 2*lr in this step "
                                                "(train.py:160-161); measured on this batch from the common starting point"}
            res[m if m != "fp32" else "fp32_parity"] = o
        head.pop("_grad", None)
        if not ar.no_fft_bench and ar.seconds == 10.0:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import proj_bench
            res["roofline_fft"] = proj_bench.roofline_block(dev)
        if not ar.no_cpu_baseline and world == 1:
            nthreads = torch.get_num_threads()
            cb = min(ar.cpu_batch, B)
            ts = cpu_baseline(a, args0, L, cb, ar.label_tokens, 5, ar.cpu_steps, weight_grads=False)
            t_cpu = sum(ts) / len(ts)
            res["cpu_baseline"] = {"value": round((cb / B) / t_cpu, 5), "unit": "steps/s (32-clip step equivalents)",
                                   "cores": nthreads, "kind": f"port, extrapolated from {cb} of {B} clips",
                                   "sample": f"oracle PGD step (torch CPU fp32, same arch/labels/length, input gradient only) on {cb} of the "
                                             f"{B} clips: 1 warm-up + {len(ts)} timed steps, mean {t_cpu:.2f} s "
                                             f"(each: {', '.join(f'{t:.2f}' for t in ts)}), scaled by {cb}/{B}"}
            if not ar.no_baseline_faithful:
                tf = cpu_baseline(a, args0, L, cb, ar.label_tokens, 5, 1, weight_grads=True)
                res["cpu_baseline"]["faithful"] = {
                    "value": round((cb / B) / tf[0], 5), "unit": "steps/s (32-clip step equivalents)", "cores": nthreads,
                    "sample": f"same sample with the model weights' requires_grad left on, as the reference leaves it "
                              f"(train.py:118,158 also back-propagate into every weight): 1 warm-up + 1 timed step, {tf[0]:.2f} s"}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
