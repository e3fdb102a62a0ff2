"""-m gpu: first execution of the RCCL branch on the hardware a test box has — ONE GPU, ONE rank (north_star / SURVEY 8e: the
universal-perturbation gradient is all-reduced with RCCL).  A fresh child process (started before this process's first GPU call
matters: it is a new interpreter, not a re-exec) initialises torch.distributed with the "nccl" backend bound to cuda:0 and runs the
PGD step with the packed all-reduce really executing, eagerly and through the two-graph capture; with one rank the sum is the
identity, so the result must equal the collective-free stepper's — bit for bit where the projection does not depend on the clean
statistics (max_phon), and to f32 rounding where it takes them from the all-reduced vector instead of reducing the clean batch
itself (snr: one float32 rounding of sum clean^2 earlier).  What it proves before the driver's 8-GPU box runs: process-group
initialisation with device_id, stream ordering between our kernels and the RCCL kernel, and hipGraph capture around it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_all_reduce_eager_and_captured():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_CHILD ")][-1]
    d = json.loads(line[len("RCCL_CHILD "):])
    print(d)
    assert d["backend"] == "nccl" and d["world"] == 1
    mp_, snr = d["max_phon"], d["snr"]
    assert mp_["split_graph"] == "_SplitGraph" and snr["split_graph"] == "_SplitGraph"
    assert mp_["eager_equal"] and mp_["graph_equal"], mp_
    assert snr["graph_equals_eager_collective"], snr
    assert snr["eager_maxdiff"] < 1e-6 and snr["graph_maxdiff"] < 1e-6, snr
    assert snr["clips_slot"] == 3.0                               # the clip count survived the all-reduce in slot 5
    for c in (mp_, snr):
        assert c["loss_collective"] == pytest.approx(c["loss_ref"], rel=1e-6)
