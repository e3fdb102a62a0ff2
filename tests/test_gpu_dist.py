"""-m gpu: the data-parallel PgdStepper with 2 ranks (gloo backend, both ranks on the one GPU of the test box) ends
with the same perturbation as a single rank stepping on the full batch, and the replicas are bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


TEXTS = ["ab cd", "hello", "a b c", "xyz w"]


def _worker(rank, world, port, norm, q, sizes=(2, 2), graph=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import pgd as opgd
    from oracle.gen_cases import cli_to_args
    from paa_amd import arch as A, synth
    from paa_amd.model import PaaModel
    from paa_amd.training_utils.pgd import PgdStepper
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    a = A.tiny()
    B, L = sizes[rank], 8000                 # ranks may hold different numbers of clips (short last global batch)
    first = sum(sizes[:rank])
    args = cli_to_args(norm, ["--snr_db", "40"] if norm == "snr" else [])
    args.device = "cuda"
    texts = TEXTS[first:first + B]
    clean = torch.from_numpy(synth.clean_audio(B, L, first_clip=first)).cuda()
    p = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2)).cuda()
    m = PaaModel(a, A.rule_weights(a), B, L, "fp32")
    st = PgdStepper(m, args, L)
    assert st.world == world and st.collective
    labels = opgd.make_labels(texts, args, B)
    if graph:                                # the two-graph form: replay = graph 1, all-reduce, graph 2
        p0 = p.clone()
        g, r = st.capture(p, clean, labels)
        p.copy_(p0)
        for _ in range(2):
            g.replay()
    else:
        for _ in range(2):
            r = st.step(p, clean, labels)
    torch.cuda.synchronize()
    out = [torch.zeros_like(p) for _ in range(world)]
    dist.all_gather(out, p)
    if rank == 0:
        q.put((p.cpu().numpy(), float(r["loss"]), all(torch.equal(o, out[0]) for o in out)))
    dist.destroy_process_group()


@pytest.mark.parametrize("norm,sizes,graph", [("snr", (2, 2), False), ("max_phon", (2, 2), False), ("snr", (3, 1), False),
                                              ("tv", (1, 3), False), ("snr", (2, 2), True), ("snr", (3, 1), True)])
def test_two_ranks_equal_one(norm, sizes, graph):
    """sizes: clips per rank — unequal shards must work with the ONE packed all-reduce (the clip count rides in slot 5);
    graph: the step replayed from PgdStepper.capture's two-graph form (_SplitGraph) instead of launched eagerly."""
    from oracle import pgd as opgd
    from oracle.gen_cases import cli_to_args
    from paa_amd import arch as A, synth
    from paa_amd.model import PaaModel
    from paa_amd.training_utils.pgd import PgdStepper
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, norm, q, sizes, graph)) for r in range(2)]
    for pr in procs:
        pr.start()
    p_dp, loss_dp, identical = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert identical
    a = A.tiny()
    B, L = 4, 8000
    args = cli_to_args(norm, ["--snr_db", "40"] if norm == "snr" else [])
    args.device = "cuda"
    clean = torch.from_numpy(synth.clean_audio(B, L)).cuda()
    p = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2)).cuda()
    m = PaaModel(a, A.rule_weights(a), B, L, "fp32")
    st = PgdStepper(m, args, L)
    for _ in range(2):
        r = st.step(p, clean, opgd.make_labels(TEXTS, args, B))
    torch.cuda.synchronize()
    assert loss_dp == pytest.approx(float(r["loss"]), rel=1e-5)
    diff = np.abs(p_dp - p.cpu().numpy())
    scale = np.abs(p.cpu().numpy()).max()
    print(f"{norm}: DP vs single max diff {diff.max() / scale:.2e}; fraction differing {(diff > 1e-6 * scale).mean():.2e}")
    assert (diff > 1e-5 * scale).mean() < 5e-3          # only where a gradient sign is numerically undecided


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: bench.py starts its two ranks itself (fresh children, before any GPU
    call) and prints rank 0's JSON line; here both ranks share the one GPU of the test box over gloo, which rehearses
    the launch path, the packed all-reduce (grad + loss + clean statistics + WER counters) and the weak-scaling
    accounting.  RCCL itself needs one device per rank and is not exercised by this test."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PAA_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
                        "--seconds", "1", "--arch", "tiny", "--label_tokens", "10", "--no_cpu_baseline", "--no_fft_bench"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and np.isfinite(d["config"]["last_loss"]) and "bf16" in d
    assert "last_wer_all_ranks" in d["config"]
    # a failing rank must fail the whole run, loudly
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                         "--seconds", "0.001", "--arch", "tiny", "--no_cpu_baseline", "--no_fft_bench"],
                        env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode != 0 and "rank" in r2.stderr


def _runner_worker(rank, world, port, logs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      PAA_DIST_BACKEND="gloo")
    from paa_amd import run_attack
    from paa_amd.training_utils import parser
    args = parser.create_arg_parser().parse_args(["--arch", "tiny", "--audio_seconds", "0.5", "--batch_size", "2", "--steps_per_epoch", "2",
                                                  "--num_epochs", "2", "--logs_dir", logs, "--dtype", "fp32", "--silent",
                                                  "--optimizer_type", "pgd", "--norm_type", "snr", "--snr_db", "40"])
    rc = run_attack.main(args)
    q.put((rank, rc, args.save_dir))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_runner_two_ranks_matches_one_rank_on_the_global_batch(tmp_path):
    """The drop-in runner launched with WORLD_SIZE=2 (as torch.distributed.run does; gloo here, both ranks on the one GPU):
    global batches of batch_size x 2 clips sharded over the ranks (build.shard_batches), the packed all-reduce inside the step,
    the epoch-end reduction of the WER counts and the sharded evaluation's all-reduce.  Rank 0 writes the files; they must equal
    those of ONE rank run with batch_size x 2 — the same global batches — up to the summation order of the shards."""
    import json
    from paa_amd import run_attack
    from paa_amd.training_utils import parser
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    logs2 = str(tmp_path / "dp2")
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, logs2, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert [r[1] for r in res] == [0, 0]
    d2 = json.load(open(os.path.join(res[0][2], "results.json")))
    p2 = torch.load(os.path.join(res[0][2], "perturbation.pt"), weights_only=True)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    args = parser.create_arg_parser().parse_args(["--arch", "tiny", "--audio_seconds", "0.5", "--batch_size", "4", "--steps_per_epoch", "2",
                                                  "--num_epochs", "2", "--logs_dir", str(tmp_path / "dp1"), "--dtype", "fp32", "--silent",
                                                  "--optimizer_type", "pgd", "--norm_type", "snr", "--snr_db", "40"])
    assert run_attack.main(args) == 0
    d1 = json.load(open(os.path.join(args.save_dir, "results.json")))
    p1 = torch.load(os.path.join(args.save_dir, "perturbation.pt"), weights_only=True)
    print("DP2 results:", json.dumps(d2))
    print("one-rank results:", json.dumps(d1))
    print(open(os.path.join(res[0][2], "train.log")).read()[-1500:])
    assert d2["finished_training"] == 1.0 and d2["best_epoch"] == d1["best_epoch"]
    for key in ("final_test_perturbed", "final_test_clean", "best_train_score"):
        for m_ in ("ctc", "wer"):
            assert d2[key][m_] == pytest.approx(d1[key][m_], rel=2e-3, abs=1e-6), (key, m_, d2[key], d1[key])
    diff = (p2 - p1).abs().numpy()
    scale = float(p1.abs().max())
    print(f"runner DP2 vs one rank: p max diff {diff.max() / scale:.2e}, fraction differing {(diff > 1e-5 * scale).mean():.2e}")
    assert (diff > 1e-5 * scale).mean() < 2e-2            # sign flips where a shard-summed gradient entry is numerically ~0
