"""The peer-to-peer exchange of the 2 + 2 mvec sums (include/nka_hip.h: nka_hip_p2p_*): an opt-in
prototype of a fused intra-node all-reduce -- no communication kernel between the final sums and the scalar step.  Proven
here with 2, 3 and 4 processes sharing the box's one GPU (hipIpc between processes on one device); see the worker."""
import os
import socket
import subprocess

from launch_util import run_ranks  # noqa: E402
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
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        # one more attempt, LOUDLY (see tests/test_soak_regressions_gpu.py): several processes that rendezvous over a local port
        # and share one GPU can fail for reasons outside the library; a defect of the exchange fails twice
        print("FIRST ATTEMPT FAILED (rc %d); its last words:\n%s" % (p.returncode, p.stdout[-1500:] + p.stderr[-3000:]), flush=True)
        p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=600)
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
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["steady_state"] and "all-reduce=p2p" in d["config"]["parallelism"]
    assert all(r["hook"] == "p2p" for r in d["ranks"]) and len({r["state_digest"] for r in d["ranks"]}) == 1
    assert all(c["identical"] for c in d["replica_check"])


@pytest.mark.parametrize("world", [2, 3])
def test_fortran_caller_sharded_through_the_peer_to_peer_exchange(world, tmp_path):
    """nka_amd/fortran/array/nka_p2p_driver.F90: a sharded caller written the reference's way (module nka_type, accel_update
    on a host array, every process the same calls on its slice: F08:58-64) whose ranks reduce through `call accel%p2p_export`
    / `p2p_attach` -- one process per rank sharing the box's GPU, the hipIpc handles gathered through a mapped file.  Against
    the unsharded oracle: num_vec after every call exact, the assembled result within the truth rule."""
    import numpy as np
    import parity_util as P
    from oracle import oracle_py as O
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    exe = os.path.join(ROOT, "nka_amd", "fortran", "build", "nka_p2p_driver")
    n, m, calls = 30011, 6, 16
    shm = tmp_path / "exchange.shm"
    shm.write_bytes(bytes(4096 + 8 * 64 * world))
    outs = [tmp_path / f"out{r}.bin" for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NKA_HIP_P2P_TIMEOUT_MS="20000")
    procs = [subprocess.Popen([exe, str(n), str(m), str(calls), str(outs[r]), str(r), str(world), str(shm)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    raws = [np.fromfile(o, dtype=np.float64) for o in outs]
    bounds = [tuple(int(v) for v in np.frombuffer(r[:2].tobytes(), dtype=np.int64)) for r in raws]
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:]))
    flavor = O.C_FLAVOR                      # the front ends' default: compact storage
    ora, spread = O.OracleNKA(n, m, flavor), P.Spread(O, n, m)
    pos = [2] * world
    for t in range(1, calls + 1):
        x = raws[0][pos[0]:pos[0] + n].copy()
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        got, nvs = np.empty(n), []
        for r, (lo, hi) in enumerate(bounds):
            assert np.array_equal(raws[r][pos[r]:pos[r] + n], x)                 # every rank drew the same global input
            nvs.append(int(raws[r][pos[r] + n]))
            got[lo:hi] = raws[r][pos[r] + n + 1:pos[r] + n + 1 + (hi - lo)]
            pos[r] += n + 1 + (hi - lo)
        err = float(np.linalg.norm(got - f) / np.linalg.norm(x))
        P.check(err, ora.state(), f"Fortran caller sharded x{world} through the peer-to-peer exchange", where=t, spread=spread.value,
                truth=spread.truth(got, x))
        if t == 7:                                                               # (the driver relaxes, then writes num_vec)
            ora.relax(); spread.relax()
        assert nvs == [ora.num_vec()] * world, (t, nvs, ora.num_vec())
    assert all("final num_vec" in lg for lg in logs)
