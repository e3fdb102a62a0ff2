"""Child process of tests/test_gpu_rccl.py (started fresh, before any GPU call of its own): ONE rank on cuda:0 with the "nccl"
(= RCCL) backend.  Runs the PGD step with the packed all-reduce really executing (PgdStepper(force_collective=True)) — eagerly and
through capture()'s two-graph form — next to the collective-free stepper from the same starting point, and prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from oracle import pgd as opgd
    from oracle.gen_cases import cli_to_args
    from paa_amd import arch as A, synth
    from paa_amd.model import PaaModel
    from paa_amd.training_utils.pgd import PgdStepper, ST_CLIPS, ST_LOSS

    a = A.tiny()
    B, L, steps = 3, 8000, 3
    clean = torch.from_numpy(synth.clean_audio(B, L)).to(dev)
    p0 = torch.from_numpy(synth.perturbation(L) * np.float32(1e-2)).to(dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    for norm in ("max_phon", "snr"):
        args = cli_to_args(norm, ["--snr_db", "40"] if norm == "snr" else [])
        args.device = "cuda"
        labels = opgd.make_labels(["ab cd", "hello", "a b c"], args, B)
        m = PaaModel(a, A.rule_weights(a), B, L, "fp32", dev)
        plain = PgdStepper(m, args, L)
        assert not plain.collective
        p_ref = p0.clone()
        for _ in range(steps):
            r_ref = plain.step(p_ref, clean, labels)
        loss_ref = float(r_ref["loss"])
        coll = PgdStepper(m, args, L, force_collective=True)
        assert coll.collective and coll.world == 1
        p_e = p0.clone()
        for _ in range(steps):
            r = coll.step(p_e, clean, labels)                     # fwd/bwd -> all_reduce(packed) over RCCL -> sign step + projection
        torch.cuda.synchronize()
        loss_e, clips = float(coll.stats[ST_LOSS]), float(coll.stats[ST_CLIPS])
        p_g = p0.clone()
        graph, _ = coll.capture(p_g, clean, labels)              # graph 1 | all_reduce | graph 2
        p_g.copy_(p0)
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        scale = float(p_ref.abs().max())
        out[norm] = {"eager_equal": bool(torch.equal(p_e, p_ref)), "graph_equal": bool(torch.equal(p_g, p_ref)),
                     "graph_equals_eager_collective": bool(torch.equal(p_g, p_e)),
                     "eager_maxdiff": float((p_e - p_ref).abs().max()) / scale, "graph_maxdiff": float((p_g - p_ref).abs().max()) / scale,
                     "loss_ref": loss_ref, "loss_collective": loss_e, "clips_slot": clips, "split_graph": type(graph).__name__}
        del graph, coll, plain, m
    dist.destroy_process_group()
    print("RCCL_CHILD " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
