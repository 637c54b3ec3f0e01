"""Worker of tests/test_sharded_cpu.py: one rank of a world-size-N gloo run.

Checks the sharded contract of the accel_update path on the CPU: every rank
drives the ORACLE (checker code; no product arithmetic runs on the CPU) on its
contiguous slice with a dot-product hook that all-reduces the local partial sum,
exactly the reference's parallel contract (src-F08/nka_type.F90:58-64), and
compares with the unsharded oracle on the full vector."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nka_amd import dist as nd  # noqa: E402
from nka_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, m, calls = 1001, 5, 14
    lo, hi = nd.slice_bounds(n, world, rank)

    # every rank must see the same bytes of the (fake) communicator id
    uid = nd.broadcast_unique_id(lambda: bytes(range(128)), rank)
    assert uid == bytes(range(128))

    # a failure on rank 0 reaches EVERY rank as the same exception (no rank is left waiting)
    def boom():
        raise OSError("no RCCL here")
    try:
        nd.broadcast_unique_id(boom, rank)
        raise AssertionError("expected a RuntimeError on every rank")
    except RuntimeError as exc:
        assert "no RCCL here" in str(exc)

    # collective agreement: one dissenting rank turns the decision on all ranks
    assert nd.all_agree(True) is True
    assert nd.all_agree(rank != world - 1) is False
    assert nd.all_agree(False) is False

    # An ASYMMETRIC early failure -- one rank cannot bind RCCL / select its device and so would never
    # reach the blocking rendezvous ncclCommInitRank -- must move EVERY rank to the torch hook before
    # anyone enters that rendezvous (attach_allreduce: pre-flight, agree, only then use_rccl).
    class FakeAcc:
        _device = 0

        def __init__(self):
            self.log = []

        def use_rccl(self, *a):
            raise AssertionError("the blocking RCCL rendezvous was entered although a rank failed its pre-flight")

        def drop_rccl(self):
            self.log.append("drop")

    saved = nd.rccl_preflight, nd.check_allreduce, nd.attach_torch_allreduce
    try:
        nd.rccl_preflight = lambda acc: rank != world - 1            # the last rank fails early
        nd.check_allreduce = lambda acc, r, w: True
        nd.attach_torch_allreduce = lambda acc, group=None: acc.log.append("torch")
        fake = FakeAcc()
        assert nd.attach_allreduce(fake, rank, world, prefer="rccl") == "torch"
        assert "torch" in fake.log
        # the whole ladder: the torch hook cannot be set up on ONE rank either (e.g. its nccl group fails there) ->
        # every rank ends on the host-staged hook; a self-test that fails on one rank moves everybody as well
        saved_staged = nd.attach_staged_allreduce

        def torch_hook_broken_on_rank0(acc, group=None):
            if rank == 0:
                raise OSError("no nccl group here")
            acc.log.append("torch")
        nd.attach_torch_allreduce = torch_hook_broken_on_rank0
        nd.attach_staged_allreduce = lambda acc, group=None: acc.log.append("staged")
        fake = FakeAcc()
        assert nd.attach_allreduce(fake, rank, world, prefer="rccl") == "staged"
        assert fake.log[-1] == "staged"
        made = []
        nd.attach_torch_allreduce = lambda acc, group=None: acc.log.append(("torch", group))
        nd.check_allreduce = lambda acc, r, w: not (acc.log and acc.log[-1] == ("torch", "DG") and r == world - 1)
        fake = FakeAcc()
        assert nd.attach_allreduce(fake, rank, world, prefer="torch", data_group=lambda: made.append(1) or "DG") == "staged"
        assert made == [1] and ("torch", "DG") in fake.log        # the data group is created on demand, once
        # a ladder that ends before it works raises on EVERY rank
        try:
            nd.attach_allreduce(FakeAcc(), rank, world, prefer="torch", data_group="DG", ladder=("rccl", "torch"))
            raise AssertionError("expected a RuntimeError")
        except RuntimeError as exc:
            assert "no hook left" in str(exc)
        # the opt-in first step "p2p" (mailboxes through hipIpc): ONE rank cannot export its mailbox -> every rank takes part
        # in the gather of the handles all the same (nobody is left waiting), learns of the failure, and moves on together
        class P2PAcc(FakeAcc):
            def p2p_export(self, nranks):
                if rank == world - 1:
                    raise OSError("hipIpcGetMemHandle refused")
                return bytes(64)

            def p2p_attach(self, handles, nranks, r):
                raise AssertionError("attach must not be reached when an export failed")

            def p2p_detach(self):
                self.log.append("p2p-detach")

            def set_dot_prod(self, fn):
                pass
        nd.attach_staged_allreduce = lambda acc, group=None: acc.log.append("staged")
        nd.check_allreduce = lambda acc, r, w: True
        fake = P2PAcc()
        assert nd.attach_allreduce(fake, rank, world, prefer="p2p", ladder=("p2p", "staged")) == "staged"
        assert fake.log == ["p2p-detach", "staged"], fake.log
        nd.attach_staged_allreduce = saved_staged
    finally:
        nd.rccl_preflight, nd.check_allreduce, nd.attach_torch_allreduce = saved

    def global_dot(x, y):
        t = torch.tensor([float(np.dot(x, y))], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    shard = O.OracleNKA(hi - lo, m)
    shard.set_dot_prod(global_dot)
    full = O.OracleNKA(n, m)
    basis = np.stack([synth.fill_numpy(7, 100 + j, 0, n, n) for j in range(3)])
    for t in range(calls):
        if t % 4 == 3:      # dependent input now and then: forces dependence drops
            coef = synth.fill_numpy(9, t, 0, 3, 3)
            x = coef @ basis
        else:
            x = synth.fill_numpy(12345, t, 0, n, n)
        f_full = x.copy()
        f_loc = x[lo:hi].copy()
        full.accel_update(f_full)
        shard.accel_update(f_loc)
        assert shard.num_vec() == full.num_vec(), (rank, t)
        assert shard.state().list_order() == full.state().list_order(), (rank, t)
        assert shard.state().free_order() == full.state().free_order(), (rank, t)
        # replicated scalars must be bitwise identical on all ranks
        c = torch.from_numpy(shard.state().c.copy())
        cmax, cmin = c.clone(), c.clone()
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        assert torch.equal(cmax, cmin), (rank, t)
        err = np.linalg.norm(f_loc - f_full[lo:hi]) / np.linalg.norm(x)
        assert err <= 1e-11, (rank, t, err)
        if t == 8:
            shard.relax(); full.relax()
    print(f"rank {rank}/{world} slice [{lo},{hi}) OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
