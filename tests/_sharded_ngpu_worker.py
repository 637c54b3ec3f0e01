"""Worker of tests/test_sharded_ngpu.py: BASELINE configs[3] as a parity test.

One rank per process, launched by torch.distributed.run BEFORE anything touches a GPU.  Every rank runs the PRODUCT path
(libnka_hip.so through nka_amd.nka) on its contiguous slice (nka_amd.dist.slice_bounds) of the global vector
F = tile(x, R): the tiled-oracle construction of tests/test_hip_fullsize.py -- every inner product of the big problem is
R = 4^k times the small one's, so the oracle runs the n0-element problem in milliseconds and the truth of the big problem
is the tiled truth of the small one.  After EVERY call, on every rank:
  * num_vec and the list order equal the oracle's on the small problem (decisions exact),
  * the replicated scalar state has the same digest on all ranks (nka_hip_state_digest),
  * the GLOBAL error ||F_out - tile(f_oracle)|| / ||F_in|| (local sums of squares, summed over the ranks on the gloo
    control plane) passes the truth rule of tests/parity_util.py at base 1e-10.
Modes (NKA_NGPU_MODE):
  rccl   one GPU per rank; the library's own RCCL communicator on the kernel stream, installed and proven by
         nka_amd.dist.attach_allreduce with NO fallback (ladder = rccl only); comm_info() must report (N, rank).
  share  every rank on cuda:0, the all-reduce staged through the host over gloo (RCCL refuses two ranks on a device): the
         rehearsal that runs on the one-GPU boxes.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nka_amd  # noqa: E402
from nka_amd import dist as nd  # noqa: E402
from nka_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
import parity_util as P  # noqa: E402


def small_inputs(n0, calls, seed):
    """As tests/test_hip_fullsize.py: independent vectors with one from a 3-dimensional span now and then, so that capacity
    AND dependence drops happen at full size."""
    basis = np.stack([synth.fill_numpy(seed + 50, j, 0, n0, n0) for j in range(3)])
    out = []
    for t in range(calls):
        if t % 7 == 5:
            out.append(synth.fill_numpy(seed + 60, t, 0, 3, 3) @ basis)
        else:
            out.append(synth.fill_numpy(seed, t, 0, n0, n0))
    return out


def main():
    mode = os.environ.get("NKA_NGPU_MODE", "share")
    n0 = int(os.environ.get("NKA_NGPU_N0", "97656"))
    R = int(os.environ.get("NKA_NGPU_R", "1024"))
    m = int(os.environ.get("NKA_NGPU_MVEC", "20"))
    flavors = [int(v) for v in os.environ.get("NKA_NGPU_FLAVORS", "2,0").split(",")]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", "0")) if mode == "rccl" else 0
    assert local < torch.cuda.device_count(), (rank, local, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n = n0 * R
    lo, hi = nd.slice_bounds(n, world, rank)
    idx = torch.arange(lo, hi, device=dev, dtype=torch.int64) % n0          # F[i] = x[i mod n0]
    calls = m + 8
    X = small_inputs(n0, calls, seed=321)
    tag = f"sharded x{world} ({mode}) n={n} m={m}"
    for flavor in flavors:
        acc = nka_amd.nka().init(hi - lo, m, flavor=flavor, device=local)
        if mode == "rccl":
            hook = nd.attach_allreduce(acc, rank, world, prefer="rccl", ladder=("rccl",))
            assert hook == "rccl" and acc.comm_info() == (world, rank), (hook, acc.comm_info(), world, rank)
        else:
            hook = nd.attach_allreduce(acc, rank, world, prefer="staged", ladder=("staged",))
            assert acc.comm_info() == (0, -1)
        ora = O.OracleNKA(n0, m, flavor)
        spread = P.Spread(O, n0, m)
        worst = 0.0
        for t, x in enumerate(X):
            f = x.copy()
            ora.accel_update(f)
            spread.update(x)
            xd = torch.from_numpy(x).to(dev)
            big = xd[idx].contiguous()
            acc.accel_update(big)
            # decisions: exact, on every rank
            assert acc.num_vec() == ora.num_vec(), (rank, flavor, t, acc.num_vec(), ora.num_vec())
            assert acc.state().list_order() == ora.state().list_order(), (rank, flavor, t)
            digs = nd.replica_digests(acc)
            assert all(d == digs[0] for d in digs), (rank, flavor, t, [f"{d:016x}" for d in digs])
            # values: the global error from local sums of squares
            ref = torch.from_numpy(f).to(dev)[idx]
            ex = torch.from_numpy(spread.exact).to(dev)[idx]
            sums = torch.stack([((big - ref) ** 2).sum(), ((big - ex) ** 2).sum(), (xd[idx] ** 2).sum()]).cpu()
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            nx = max(float(sums[2]) ** 0.5, 1e-300)
            err, err_ex = float(sums[0]) ** 0.5 / nx, float(sums[1]) ** 0.5 / nx
            P.check(err, ora.state(), f"{tag} flavor {flavor}", base=1e-10, where=t, spread=spread.value,
                    truth=(err_ex, spread.err_ref, n, m))
            worst = max(worst, err)
            del big, ref, ex
        P.finish()
        assert acc.defined() and acc.num_vec() == m
        if rank == 0:
            print(f"{tag} flavor {flavor}: hook={hook} comm={acc.comm_info()} worst rel err vs tiled oracle {worst:.2e}", flush=True)
        acc.delete()
    print(f"rank {rank}/{world} slice [{lo},{hi}) OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
