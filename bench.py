#!/usr/bin/env python3
"""PGD steps/s on synthetic fixed-length 16 kHz clips (BASELINE.json metric, configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): untargeted PGD, --norm_type snr --snr_db 40, Wav2Vec2-base architecture with
rule-generated weights (no checkpoint exists offline), 32 x 10 s clips per GPU, lr 1e-4.  One "step" is one
full pass of the hot path over one batch: compose+clamp -> Wav2Vec2 forward -> CTC -> backward to the
waveform -> [all-reduce over ranks] -> sign step -> projection, plus the per-step loss / greedy-decode
read-back the reference does (train.py:146-153), pipelined one step behind so it stays off the critical path.
Weak scaling: the per-GPU batch is fixed, `value` counts 32-clip steps over ALL ranks per second.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VARIANTS = {v: (f"k_gemm_bf<{'256' if v & 32 else '128'},{'64' if v & 8 else '128'},{'split' if v & 4 else 'bf16'}>" if v & 16 else
                f"k_gemm<{'64' if v & 8 else '128'},{'split' if v & 4 else 'bf16'},A{'k' if v & 2 else 'm'},B{'k' if v & 1 else 'n'}>")
            for v in range(64)}
VARIANTS.update({51: "k_gemm_bf<256,128,bf16>", 55: "k_gemm_bf<256,128,split>", 59: "k_gemm_bf<192,128,bf16>",
                 40: "k_gemm_win<bf16>", 44: "k_gemm_win<split>"})


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--arch", default="base", choices=["base", "large-lv60", "tiny"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--norm_type", default="snr")
    ap.add_argument("--snr_db", type=float, default=40.0)
    ap.add_argument("--label_tokens", type=int, default=150)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_batch", type=int, default=4, help="clips in the bounded CPU-baseline sample")
    ap.add_argument("--no_prof", action="store_true", help="skip the per-launch GEMM event timing")
    ap.add_argument("--prof_steps", type=int, default=1,
                    help="number of timed steps (the last ones) whose GEMM launches carry HIP events; 0 = every timed step")
    ap.add_argument("--graph", action="store_true", help="replay the step from captured hipGraphs (single rank; implies --no_prof)")
    return ap.parse_args()


def make_transcript(i: int, n_chars: int, seed: int) -> str:
    from paa_amd import synth
    words = ["the", "quick", "brown", "fox", "jumps", "over", "a", "lazy", "dog", "and", "runs", "away", "home"]
    u = synth.uniform(synth.key_of(f"txt{i}", seed), n_chars)
    s, k = "", 0
    while len(s) < n_chars:
        s += words[int(u[k % n_chars] * len(words))] + " "
        k += 1
    return s[:n_chars].rstrip().ljust(n_chars, "a")


def cpu_baseline(a, args_ns, L, batch, label_tokens, seed):
    """The oracle's PGD step (torch CPU, float32, all host cores) on a bounded sample of the same workload."""
    from oracle import pgd as opgd, wav2vec2 as OW
    from paa_amd import arch as A, synth
    sd = OW.to_torch(A.rule_weights(a))
    clean = torch.from_numpy(synth.clean_audio(batch, L, seed=seed))
    p = torch.from_numpy(synth.perturbation(L, seed=seed) * np.float32(1e-3))
    texts = [make_transcript(i, label_tokens, seed) for i in range(batch)]
    labels = opgd.make_labels(texts, args_ns, batch)
    t0 = time.perf_counter()
    opgd.pgd_step(sd, a, args_ns, clean, labels, p)
    dt = time.perf_counter() - t0
    return dt


def main():
    ar = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ar.gpus:
        if world == 1 and ar.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    ndev = max(torch.cuda.device_count(), 1)
    local = local % ndev                       # one rank per GPU; the modulo only matters for single-GPU rehearsals
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
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
    from paa_amd.training_utils.pgd import PgdStepper

    a = {"base": A.BASE, "large-lv60": A.LARGE_LV60, "tiny": A.tiny()}[ar.arch]
    L = int(round(ar.seconds * 16000))
    B = ar.batch
    args = parser.create_arg_parser().parse_args(["--norm_type", ar.norm_type, "--snr_db", str(ar.snr_db), "--lr", "1e-4",
                                                  "--attack_mode", "untargeted", "--optimizer_type", "pgd", "--seed", "5",
                                                  "--device", str(dev), "--dtype", ar.dtype])
    model = PaaModel(a, A.rule_weights(a), B, L, ar.dtype, dev)
    stepper = PgdStepper(model, args, L, None, build.init_phon_threshold_tensor(args))
    # two resident synthetic batches per rank, disjoint clips across ranks
    NB = 2
    cleans, labels, texts = [], [], []
    for s in range(NB):
        first = (s * world + rank) * B
        cleans.append(torch.from_numpy(synth.clean_audio(B, L, seed=5, first_clip=first)).to(dev))
        tx = [make_transcript(first + b, ar.label_tokens, 5) for b in range(B)]
        texts.append(tx)
        labels.append(loss_helpers.make_labels(tx, None, args, B).to(device=dev, dtype=torch.int32))
    # p0 ~ N(0,1) projected once (build.py:301-304)
    p = build.init_perturbation(args, L, None, None, cleans[0]).detach().clone()
    logits_buf = [torch.empty(B, model.frames, a.vocab_size, device=dev) for _ in range(2)]
    ids_host = [torch.empty(B, model.frames, dtype=torch.int16).pin_memory() for _ in range(2)]
    loss_host = [torch.empty(1).pin_memory() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    bookkeeping = {"loss": [], "wer": []}

    graphs = None
    if ar.graph and world == 1:
        ar.no_prof = True
        p_snapshot = p.clone()
        graphs = [stepper.capture(p, cleans[j], labels[j], logits_out=logits_buf[j]) for j in range(NB)]
        p.copy_(p_snapshot)                      # capture ran the step for real: restore the starting point

    def launch(i):
        k = i % 2
        if graphs is not None:
            graphs[i % NB][0].replay()
            r = graphs[i % NB][1]
        else:
            r = stepper.step(p, cleans[i % NB], labels[i % NB], want_logits=True, logits_out=logits_buf[k])
        ids_host[k].copy_(torch.argmax(r["logits"], dim=-1).to(torch.int16), non_blocking=True)
        loss_host[k].copy_(r["loss"].reshape(1), non_blocking=True)
        done[k].record()

    def collect(i):
        """Host side of train.py:146-153 for step i: loss value + WER of the greedy decode vs ground truth."""
        k = i % 2
        done[k].synchronize()
        pred = [t.lower() for t in loss_helpers.greedy_decode_ids(ids_host[k].tolist())]
        e, w = loss_helpers.wer_counts(pred, loss_helpers.clean_transcripts(texts[i % NB]))
        bookkeeping["loss"].append(float(loss_host[k][0]))
        bookkeeping["wer"].append(e / max(w, 1))

    def run(n, start, on_step=None):
        for i in range(start, start + n):
            if on_step is not None:
                on_step(i)
            launch(i)
            if i > start:
                collect(i - 1)
        collect(start + n - 1)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    run(ar.warmup, 0) if ar.warmup > 0 else None
    prof_on = (not ar.no_prof) and rank == 0
    # Per-launch HIP events around every GEMM are recorded on the LAST `--prof_steps` timed steps only.  The first timing
    # event recorded on a stream switches its HIP queue to profiled dispatch for the rest of the process, which slows
    # every later launch (~4 % on this step: 50.1 -> 48.2 steps/s); recording at the end keeps the steps before it
    # unperturbed while the sample is still taken live inside the timed region.  --prof_steps 0: every timed step.
    n_prof = ar.steps if ar.prof_steps <= 0 else min(ar.prof_steps, ar.steps)
    prof_from = ar.warmup + ar.steps - n_prof
    if prof_on:
        _lib.check(_lib.lib().paa_prof_enable(4096 * n_prof))
        _lib.check(_lib.lib().paa_prof_pause(1))
    sync()
    t0 = time.perf_counter()
    run(ar.steps, ar.warmup, (lambda i: _lib.lib().paa_prof_pause(0) if i == prof_from else None) if prof_on else None)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax[0])

    roofline = None
    if prof_on:
        out = (C.c_double * 192)()
        _lib.check(_lib.lib().paa_prof_read(out))
        _lib.lib().paa_prof_enable(0)
        rows = [(v, out[3 * v], out[3 * v + 1], out[3 * v + 2]) for v in range(64) if out[3 * v] > 0]
        rows.sort(key=lambda r: -r[2])
        if rows:
            v, n, ms, fl = rows[0]
            # split mode issues 3 MFMA per algorithmic product; `achieved` counts ALGORITHMIC flops only
            achieved = fl / (ms * 1e-3) / 1e12
            peak = 2500.0
            # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
            # this same command (profiles/r1_hbm_traffic_pmc.json, FETCH_SIZE doubled per the gfx950 correction); null if absent
            traffic = None
            pmc_name = {51: "k_gemm_bf<256, 128, 0, 2,", 59: "k_gemm_bf<192, 128, 0, 2,", 55: "k_gemm_bf<256, 128, 1, 4,"}.get(v)
            try:
                with open(os.path.join(ROOT, "profiles", "r1_hbm_traffic_pmc.json")) as f:
                    for k in json.load(f)["kernels"]:
                        if pmc_name and pmc_name in k["kernel"] and ar.dtype == "bf16" and B == 32 and ar.arch == "base":
                            traffic = k["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            roofline = {"bound": "mfma", "kernel": VARIANTS[v], "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic, "launches": int(n),
                        "avg_launch_us": round(ms * 1e3 / n, 2),
                        "sampled_steps": n_prof, "share_of_step": round(ms * 1e-3 / (dt * n_prof / ar.steps), 4),
                        "all_gemm_variants": [{"kernel": VARIANTS[r[0]], "launches": int(r[1]), "ms": round(r[2], 3),
                                               "tflops": round(r[3] / (r[2] * 1e-3) / 1e12, 2)} for r in rows]}

    if rank == 0:
        fl_step = 2.0 * a.fwd_flops_per_clip(L) * B
        res = {
            "metric": "pgd_steps_per_sec", "value": round(world * ar.steps / dt, 4),
            "unit": f"steps/s (one step = PGD step on {B} x {ar.seconds:g} s clips; counted over all ranks)",
            "n_gpus": world, "steps": ar.steps, "warmup": ar.warmup, "ms_per_step": round(1e3 * dt / ar.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if ar.dtype == "bf16" else "bf16x3 (split-bf16, fp32-parity)", "data": "synthetic",
            "config": {"workload": f"untargeted PGD, norm_type {ar.norm_type} snr_db {ar.snr_db:g}, Wav2Vec2-{ar.arch} "
                                   f"(rule-generated weights), {B}x{ar.seconds:g}s 16 kHz clips per GPU, {ar.label_tokens}-token labels",
                       "global_batch": B * world, "per_gpu_batch": B, "samples_per_clip": L, "parallelism": f"dp{world}",
                       "model_tflop_per_step_per_gpu": round(fl_step / 1e12, 3),
                       "model_tflops_achieved_per_gpu": round(fl_step / (dt / ar.steps) / 1e12, 2),
                       "hip_graph": bool(graphs), "last_loss": bookkeeping["loss"][-1], "last_wer": bookkeeping["wer"][-1]},
            "roofline": roofline,
        }
        if not ar.no_cpu_baseline and world == 1:
            nthreads = torch.get_num_threads()
            cb = min(ar.cpu_batch, B)
            t_cpu = cpu_baseline(a, args, L, cb, ar.label_tokens, 5)
            res["cpu_baseline"] = {"value": round((cb / B) / t_cpu, 5), "unit": "steps/s (32-clip step equivalents)",
                                   "cores": nthreads, "kind": "port",
                                   "sample": f"1 oracle PGD step (torch CPU fp32, same arch/labels/length) on {cb} of the {B} clips: "
                                             f"{t_cpu:.1f} s, scaled by {cb}/{B}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
