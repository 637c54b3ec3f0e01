"""Worker of tests/test_sharded_ngpu.py::test_sharded_reference_order_sums_*: N ranks, each on its contiguous slice, with the
sums formed in the reference's order RANK AFTER RANK (nka_hip_set_sum_order + nka_hip_set_shard; nka_hip.hip: ordered_chain).
The N-rank run must return the bits of the SINGLE-rank compiled reference:
  * every golden scenario of the compiled reference (tests/golden/scenario_*.npz, all three flavours): np.array_equal on
    this rank's slice of every output, the num_vec trace exact;
  * n = 100 003, m = 20, independent and dependent inputs, relax and a tolerance change in mid-stream, against
    oracle/_ref/libnka_ref_f08.so itself (the src-F08 module compiled from /root/reference where it lies; if it did not
    travel: the oracle's F08 flavour, pinned to it bit for bit) and the oracle's other two flavours.
Ranks share cuda:0 (NKA_NGPU_MODE=share: the sums staged through the host over gloo -- any hook that sums carries the
chain) or take one GPU each with the library's RCCL communicator (NKA_NGPU_MODE=rccl)."""
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
from oracle import oracle_py as O  # noqa: E402
import scenarios as S  # noqa: E402

KEYS = {0: "f_out_f08", 1: "f_out_f08vec", 2: "f_out_c"}


def main():
    mode = os.environ.get("NKA_NGPU_MODE", "share")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", "0")) if mode == "rccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ladder = ("rccl",) if mode == "rccl" else ("staged",)

    def make(n, m, flavor):
        lo, hi = nd.slice_bounds(n, world, rank)
        many = os.environ.get("NKA_TEST_CHAIN_MANY") == "1"          # the whole-device form of the sums wherever a slice has a full block
        acc = nka_amd.nka(diagnostic=many).init(hi - lo, m, flavor=flavor, device=local)
        if many:
            acc.set_tuning("chain_many", 1)
        nd.attach_allreduce(acc, rank, world, prefer=ladder[0], ladder=ladder)      # (tells the handle its slice: set_shard)
        acc.set_sum_order(nka_amd.SUMS_REFERENCE_ORDER)
        return acc, lo, hi

    def update(acc, lo, hi, x):
        t = torch.from_numpy(np.ascontiguousarray(x[lo:hi])).to(dev)
        acc.accel_update(t)
        return t.cpu().numpy()

    # ---- the golden scenarios of the compiled reference -------------------------------------------------------
    checked = 0
    for name in S.scenario_names():
        g = S.load(name)
        n, m = int(g["n"]), int(g["mvec"])
        for flavor, key in KEYS.items():
            if key not in g.files:
                continue
            acc, lo, hi = make(n, m, flavor)
            outs, trace = S.replay(acc, g, update=lambda a, f: update(a, lo, hi, f))
            assert np.array_equal(trace, g["num_vec"]), (rank, name, flavor)
            for u in range(len(outs)):
                assert np.array_equal(outs[u], g[key][u][lo:hi]), (rank, name, flavor, u)
            digs = nd.replica_digests(acc)
            assert all(d == digs[0] for d in digs), (rank, name, flavor)
            assert acc.defined()
            acc.delete()
            checked += 1

    # ---- n = 100 003, m = 20 against the compiled src-F08 module itself ----------------------------------------
    n, m, calls = 100003, 20, 27
    rng = np.random.default_rng(77)
    basis = rng.standard_normal((3, n))
    X = [rng.standard_normal(3) @ basis if t % 8 == 5 else rng.standard_normal(n) for t in range(calls)]
    X[12] = X[11].copy()                                   # a repeated input: s == 0 -> relax (F08:275), exactly
    which = []
    for flavor in (0, 2, 1):
        use_ref = flavor == 0 and O.have_ref()
        ref = O.RefF08(n, m) if use_ref else O.OracleNKA(n, m, flavor)
        which.append("compiled src-F08 reference" if use_ref else f"oracle flavour {flavor}")
        acc, lo, hi = make(n, m, flavor)
        for t, x in enumerate(X):
            f = x.copy()
            ref.accel_update(f)
            out = update(acc, lo, hi, x)
            assert np.array_equal(out, f[lo:hi]), (rank, flavor, t, float(np.abs(out - f[lo:hi]).max()))
            assert acc.num_vec() == ref.num_vec(), (rank, flavor, t)
            if t == 15:
                acc.relax(); ref.relax()
            if t == 20:
                acc.set_vec_tol(0.2); ref.set_vec_tol(0.2)
        digs = nd.replica_digests(acc)
        assert all(d == digs[0] for d in digs), (rank, flavor)
        assert acc.defined()
        acc.delete()
    if rank == 0:
        print(f"sharded reference-order sums x{world} ({mode}): {checked} scenario replays and n={n} m={m} x {calls} calls against "
              f"{', '.join(which)}: every bit equal", flush=True)
    print(f"rank {rank}/{world} OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
