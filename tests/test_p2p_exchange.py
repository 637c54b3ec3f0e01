"""The peer-to-peer exchange of the 2 + 2 mvec sums (include/nka_hip.h: nka_hip_p2p_*): an opt-in
prototype of a fused intra-node all-reduce -- no communication kernel between the final sums and the scalar step.  Proven
here with 2, 3 and 4 processes sharing the box's one GPU (hipIpc between processes on one device); see the worker."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_peer_to_peer_exchange_equals_the_rank_ordered_staged_hook_bit_for_bit(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_p2p_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    assert p.stdout.count("p2p OK") == world


def test_bench_rehearsal_with_the_peer_to_peer_exchange():
    """bench.py --gpus 2 --allreduce p2p with both ranks on the box's GPU (control plane gloo): the line names the hook that
    ran, the replicas agree, every rank reports it."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", NKA_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--vlen", "3000001", "--mvec", "6",
           "--steps", "6", "--backend", "gloo", "--allreduce", "p2p", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["steady_state"] and "all-reduce=p2p" in d["config"]["parallelism"]
    assert all(r["hook"] == "p2p" for r in d["ranks"]) and len({r["state_digest"] for r in d["ranks"]}) == 1
    assert all(c["identical"] for c in d["replica_check"])
